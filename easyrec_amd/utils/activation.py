"""Activation lookup (reference easy_rec/python/utils/activation.py:68-124 `get_activation`)."""
import math

import torch

from easyrec_amd import kernels
from easyrec_amd.core import context


def dice(x, name='dice', epsilon=1e-9, training=True, momentum=0.99):
  """Dice (reference utils/activation.py:14-44): alpha*(1-p)*x + p*x, p = sigmoid(BN_noaffine(x))."""
  vs = context.varstore()
  n = x.shape[-1]
  alpha = vs.get_variable('alpha_' + name, (n,), 'zeros')
  mm = vs.get_variable(name + '/batch_normalization/moving_mean', (n,), 'zeros', trainable=False)
  mv = vs.get_variable(name + '/batch_normalization/moving_variance', (n,), 'ones', trainable=False)
  shape = x.shape
  x2 = x.reshape(-1, n)
  if not training:
    p = torch.sigmoid((x2 - mm) * torch.rsqrt(mv + epsilon))
    return (alpha * (1.0 - p) * x2 + p * x2).reshape(shape)
  frozen = context.current().building
  y = kernels.DiceFn.apply(x2.contiguous(), alpha, None if frozen else mm, None if frozen else mv, epsilon,
                           momentum, alpha.grad)
  return y.reshape(shape)


def gelu(x, name='gelu'):
  cdf = 0.5 * (1.0 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))
  return x * cdf


_RELU_NAMES = ('relu', 'tf.nn.relu', 'nn.relu', 'tf.keras.activations.relu')


def is_relu(activation_string):
  return isinstance(activation_string, str) and activation_string.lower() in _RELU_NAMES


def get_activation(activation_string, **kwargs):
  """Returns a callable(x, name=...) or None for linear."""
  if not isinstance(activation_string, str):
    return activation_string
  if not activation_string:
    return None
  act = activation_string.lower()
  if act == 'linear':
    return None
  if act in _RELU_NAMES:
    return lambda x, name=None: torch.relu(x)
  if act == 'gelu':
    return gelu
  if act in ('leaky_relu', 'tf.nn.leaky_relu', 'prelu'):
    return lambda x, name=None: torch.nn.functional.leaky_relu(x, 0.2)
  if act == 'dice':
    return lambda x, name='dice': dice(x, name=name, **kwargs)
  if act in ('elu', 'tf.nn.elu'):
    return lambda x, name=None: torch.nn.functional.elu(x)
  if act in ('selu', 'tf.nn.selu'):
    return lambda x, name=None: torch.selu(x)
  if act in ('tanh', 'tf.tanh', 'tf.nn.tanh'):
    return lambda x, name=None: torch.tanh(x)
  if act in ('swish', 'tf.nn.swish'):
    return lambda x, name=None: x * torch.sigmoid(x)
  if act in ('sigmoid', 'tf.nn.sigmoid', 'tf.sigmoid'):
    return lambda x, name=None: torch.sigmoid(x)
  raise ValueError('unsupported activation: %s' % activation_string)

"""Feature-interaction layers of the backbone API (reference easy_rec/python/layers/keras/interaction.py).

FM    (:24-44)   0.5 * ((sum_f e)^2 - sum_f e^2): [B, D] with `use_variant`, else summed over D -> [B, 1].
Cross (:131-308) DCN-v2: x_{l+1} = x0 * (W x_l + b + diag_scale * x_l) + x_l; W full rank d x d or low rank
                 U (d x r) V (r x d); kernel init truncated_normal (keras: stddev 0.05), bias zeros.
                 W x_l is an MFMA GEMM (er_gemm, fp32 or bf16); bias, diag term, Hadamard product and residual
                 are ONE fused epilogue kernel (er_cross_v2_epilogue) instead of BiasAdd/Mul/Mul/Add.
CIN   (:311-409) xDeepFM's compressed interaction network over a [B, H0, D] stack of field embeddings: per layer
                 x_{k+1}[b,n,d] = relu(sum_{h,m} W[n,h,m] x_k[b,h,d] x_0[b,m,d] + bias[n]); output = the feature maps
                 summed over d, concatenated over the layers.  Kernels `cin_kernel_<k>` [H_k+1, H_k, H0] he_normal,
                 `cin_bias_<k>` zeros.  The outer product is written once and contracted on the matrix cores
                 (kernels.CINFn: er_cin_outer_fwd + er_gemm_f32 + er_cin_act_pool_fwd per layer).
"""
import torch

from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.layers import dnn
from easyrec_amd.utils.activation import get_activation


class FM(object):

  def __init__(self, params, name='fm', reuse=None, **kwargs):
    self.name = name
    self.use_variant = params.get_or_default('use_variant', False)

  def __call__(self, inputs, **kwargs):
    blk = inputs.uniform_block() if hasattr(inputs, 'uniform_block') else None
    if blk is not None:
      base, col0, F, D = blk
      x = base if (col0 == 0 and base.shape[1] == F * D) else base[:, col0:col0 + F * D]
    elif isinstance(inputs, (list, tuple)):
      dims = set(int(t.shape[-1]) for t in inputs)
      if len(dims) != 1:
        raise ValueError('all embedding dim must be equal in FM layer:' + ','.join(str(d) for d in dims))
      F, D = len(inputs), inputs[0].shape[1]
      x = torch.cat(list(inputs), dim=1)
    else:
      assert inputs.dim() == 3, 'input of FM layer must be a 3D tensor or a list of 2D tensors'
      _, F, D = inputs.shape
      x = inputs.reshape(inputs.shape[0], F * D)
    cross = kernels.FMFn.apply(x, F, D)  # [B, D] = 0.5 * (square_of_sum - sum_of_square)
    if self.use_variant:
      return cross
    return kernels.RowSumFn.apply(cross)


class Cross(object):

  def __init__(self, params, name='cross', reuse=None, **kwargs):
    self.name = name
    self._projection_dim = params.get_or_default('projection_dim', None)
    self._diag_scale = float(params.get_or_default('diag_scale', 0.0))
    self._use_bias = params.get_or_default('use_bias', True)
    self._preactivation = get_activation(params.get_or_default('preactivation', None) or '')
    self._kernel_initializer = params.get_or_default('kernel_initializer', 'truncated_normal')
    self._bias_initializer = params.get_or_default('bias_initializer', 'zeros')
    if self._diag_scale < 0:
      raise ValueError('`diag_scale` should be non-negative. Got `diag_scale` = {}'.format(self._diag_scale))

  def __call__(self, inputs, **kwargs):
    if isinstance(inputs, (list, tuple)):
      x0, x = inputs
    else:
      x0, x = inputs, inputs
    if x0.shape[-1] != x.shape[-1]:
      raise ValueError('`x0` and `x` dimension mismatch! Got `x0` dimension {}, and x dimension {}. This case '
                       'is not supported yet.'.format(x0.shape[-1], x.shape[-1]))
    vs = context.varstore()
    d = x.shape[-1]
    bias = vs.get_variable(self.name + '/dense/bias', (d,), self._bias_initializer) if self._use_bias else None
    if self._projection_dim is None:
      w = vs.get_variable(self.name + '/dense/kernel', (d, d), self._kernel_initializer)
      fused = self._fused(x0, x, w, bias)
      if fused is not None:
        return fused
      u = dnn._linear(x, w, None)
    else:
      r = int(self._projection_dim)
      wu = vs.get_variable(self.name + '/dense_u/kernel', (d, r), self._kernel_initializer)
      wv = vs.get_variable(self.name + '/dense_v/kernel', (r, d), self._kernel_initializer)
      u = dnn._linear(dnn._linear(x, wu, None), wv, None)
    if self._preactivation is not None:
      u = self._preactivation(u + bias if bias is not None else u)
      bias = None
    x0g = kernels.slot_gate(x0)
    return kernels.CrossV2EpilogueFn.apply(x0g, x0g if x is x0 else kernels.slot_gate(x), u, bias, self._diag_scale, None if bias is None else bias.grad,
                                           kernels.grad_sink_of(x0))


  def _fused(self, x0, x, w, bias):
    """The whole layer as one launch forward / one per layer backward (kernels.CrossLayerFn), or None: the general form
    (low rank, a preactivation, evaluation, operands the fused contraction does not take)."""
    ctx = context.current()
    be = kernels.hip()
    if not (getattr(be, 'fused_cross', False) and self._preactivation is None and torch.is_grad_enabled() and
            ctx.is_training and not ctx.building and x0.dim() == 2 and x.dim() == 2 and x0.is_contiguous() and
            x.is_contiguous() and kernels.grad_slots_of_step() is not None and w.grad is not None and
            (bias is None or bias.grad is not None)):
      return None
    bf16 = getattr(ctx, 'dense_dtype', 'f32') == 'bf16'
    if bf16 and be._cross_b16(w, True, x0, x) is False:
      return None
    same = x is x0
    x0g = kernels.slot_gate(x0)
    prev = None if same else kernels.cross_source_of(x)
    out = kernels.CrossLayerFn.apply(x0g, x0g if same else x, w, bias, self._diag_scale, w.grad,
                                     None if bias is None else bias.grad, kernels.grad_sink_of(x0), bf16, prev)
    out._er_cross_src = kernels.take_last_cross_source()
    return out


class CIN(object):

  def __init__(self, params, name='cin', reuse=None, **kwargs):
    self.name = name
    self._hidden_feature_sizes = [int(h) for h in params.get_or_default('hidden_feature_sizes', [])]
    assert len(self._hidden_feature_sizes) > 0, \
        'parameter hidden_feature_sizes must be a list of int with length greater than 0'
    assert not params.get_or_default('kernel_regularizer', None) and not params.get_or_default('bias_regularizer', None), \
        'CIN kernel / bias regularizers are outside the hot-path scope'

  def __call__(self, inputs, **kwargs):
    if isinstance(inputs, (list, tuple)):  # a feature list: stacked along a new field axis
      inputs = torch.stack(list(inputs), dim=1)
    if inputs.dim() != 3:
      raise ValueError('Unexpected inputs dimensions %d, expect to be 3 dimensions' % inputs.dim())
    vs = context.varstore()
    sizes = [int(inputs.shape[1])] + self._hidden_feature_sizes
    ws, bs = [], []
    for i in range(len(self._hidden_feature_sizes)):
      ws.append(vs.get_variable('%s/cin_kernel_%d' % (self.name, i), (sizes[i + 1], sizes[i], sizes[0]), 'he_normal'))
      bs.append(vs.get_variable('%s/cin_bias_%d' % (self.name, i), (sizes[i + 1],), 'zeros'))
    n = len(ws)
    return kernels.CINFn.apply(inputs, n, *ws, *bs, *[w.grad for w in ws], *[b.grad for b in bs])


class DotInteraction(object):
  """DLRM's dot interaction as a backbone block (reference layers/keras/interaction.py:47-127): all pairwise dot
  products of a list of [B, D] features, listed in the order of the LOWER triangle of the F x F matrix row by row
  (`tf.boolean_mask`), with the diagonal when `self_interaction`.  er_dot_interaction produces the same products in the
  order of the DLRM model class (upper triangle, model/dlrm.py:51-57): pair (i, j) there is pair (j, i) here, so the
  output is a column permutation of the kernel's."""

  def __init__(self, params, name=None, reuse=None, **kwargs):
    self.name = name
    self._self_interaction = bool(params.get_or_default('self_interaction', False))
    assert not params.get_or_default('skip_gather', False), 'DotInteraction.skip_gather is outside the hot-path scope'
    self._perm = {}

  def _permutation(self, F, device):
    key = (F, str(device))
    if key not in self._perm:
      off = 0 if self._self_interaction else 1
      mine = {p: k for k, p in enumerate((i, j) for i in range(F) for j in range(i + off, F))}
      order = [(i, j) for i in range(F) for j in range(i + 1 if self._self_interaction else i)]
      self._perm[key] = torch.tensor([mine[(j, i)] for i, j in order], dtype=torch.int64, device=device)
    return self._perm[key]

  def __call__(self, inputs, **kwargs):
    if isinstance(inputs, (list, tuple)):
      dims = set(int(t.shape[-1]) for t in inputs)
      if len(dims) != 1:
        raise ValueError('Input tensors` dimensions must be equal, got: %s' % sorted(dims))
      F, D = len(inputs), int(inputs[0].shape[-1])
      x = kernels.concat_cols(list(inputs))
    else:
      assert inputs.dim() == 3, 'input of dot func must be a 3D tensor or a list of 2D tensors'
      _, F, D = inputs.shape
      x = inputs.reshape(inputs.shape[0], F * D)
    out = kernels.DotInteractionFn.apply(x, F, D, self._self_interaction)
    return out.index_select(1, self._permutation(F, out.device))

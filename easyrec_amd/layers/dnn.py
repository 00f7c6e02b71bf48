"""DNN: dense -> BatchNorm -> activation -> dropout per layer.

Mirror of reference easy_rec/python/layers/dnn.py:13-87.  Per layer the reference issues
MatMul, BiasAdd, FusedBatchNorm/moments, Relu; here: one GEMM + ONE fused HIP kernel pair
(`er_bn_act_fwd` / `er_bn_act_bwd`: bias + batch statistics + affine + ReLU).
TF defaults relied on: tf.layers.dense use_bias=True, glorot_uniform kernel,
zeros bias; tf.layers.batch_normalization momentum 0.99, epsilon 1e-3, biased batch variance.
"""
import logging

import torch

from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.utils.activation import get_activation, is_relu

BN_MOMENTUM = 0.99
BN_EPSILON = 1e-3


def _linear(x2, w, b, src=None, sink=None):
  """x2 [rows, in] . w [in, units] (+ b) through the hand-written MFMA GEMMs (kernels.LinearFn); weight and
  bias gradients accumulate directly into the variables' slices of the flat gradient buffer."""
  ctx = context.current()
  bf16 = getattr(ctx, 'dense_dtype', 'f32') == 'bf16'
  if not torch.is_grad_enabled() or not ctx.is_training:
    kernels.finish_pending_bn(x2)
    return kernels.hip().gemm(kernels.GEMM_NN, x2 if x2.stride(-1) == 1 else x2.contiguous(), w.detach(),
                              bias=None if b is None else b.detach(), bf16=bf16)
  wg = w.grad if (w.requires_grad and w.grad is not None) else None
  bg = b.grad if (b is not None and b.requires_grad and b.grad is not None) else None
  return kernels.LinearFn.apply(kernels.slot_gate(x2), w, b, wg, bg, bf16, src, sink)


def dense(x, units, name, l2_reg=None, use_bias=True, kernel_initializer='glorot_uniform', head=False):
  """tf.layers.dense(activation=None): returns x @ kernel (+ bias).  Variables <name>/kernel, /bias.
  head: the caller hands the result UNCHANGED to the prediction dict as a rank model's logits (units == 1): while training
  the projection is then computed by the loss builder's fused head launch (kernels.HeadFn) - nothing may read the returned
  tensor before the loss graph is built (the estimator materialises whatever is still pending after it)."""
  vs = context.varstore()
  in_dim = x.shape[-1]
  w = vs.get_variable(name + '/kernel', (in_dim, units), kernel_initializer, l2=l2_reg or 0.0)
  b = vs.get_variable(name + '/bias', (units,), 'zeros') if use_bias else None
  shape = x.shape
  ctx = context.current()
  if head and units == 1 and x.dim() == 2 and torch.is_grad_enabled() and ctx.is_training and not ctx.building and \
      getattr(kernels.hip(), 'fused_head', False) and hasattr(ctx, 'heads'):
    wg = w.grad if (w.requires_grad and w.grad is not None) else None
    bg = b.grad if (b is not None and b.requires_grad and b.grad is not None) else None
    bf16 = getattr(ctx, 'dense_dtype', 'f32') == 'bf16'
    return kernels.HeadFn.apply(x, w, b, wg, bg, bf16, kernels.bn_source_of(x), ctx.heads)
  y = _linear(x if x.dim() == 2 else x.reshape(-1, in_dim), w, b, kernels.bn_source_of(x), kernels.grad_sink_of(x))
  return y.reshape(shape[:-1] + (units,))


def dense_bn_act(x, units, name, l2_reg, use_bias, use_bn, act_relu, training, bn_name=None,
                 kernel_initializer='glorot_uniform', defer_apply=False):
  """GEMM followed by the fused bias + BatchNorm(train) + ReLU kernel.  defer_apply: the caller hands the result straight to
  the ONE op that runs the BatchNorm finalize + apply inside its own launch (kernels.WideFmConcatFn); anything else must
  call kernels.finish_pending_bn on it first."""
  ctx = context.current()
  vs = ctx.varstore
  in_dim = x.shape[-1]
  w = vs.get_variable(name + '/kernel', (in_dim, units), kernel_initializer, l2=l2_reg or 0.0)
  b = vs.get_variable(name + '/bias', (units,), 'zeros') if use_bias else None
  gamma = beta = mm = mv = None
  if use_bn:
    bn = bn_name or (name + '/bn')
    gamma = vs.get_variable(bn + '/gamma', (units,), 'ones')
    beta = vs.get_variable(bn + '/beta', (units,), 'zeros')
    mm = vs.get_variable(bn + '/moving_mean', (units,), 'zeros', trainable=False)
    mv = vs.get_variable(bn + '/moving_variance', (units,), 'ones', trainable=False)
  shape = x.shape
  act = kernels.ACT_RELU if act_relu else kernels.ACT_NONE
  freeze = ctx.building and training  # build pass: do not touch the moving statistics
  if use_bn and training and torch.is_grad_enabled():
    # GEMM (bias + column statistics in its epilogue) + ONE fused BatchNorm/ReLU launch
    bufs = None
    if w.grad is not None and gamma.grad is not None and beta.grad is not None:
      bufs = (w.grad, gamma.grad, beta.grad)
    bf16 = getattr(ctx, 'dense_dtype', 'f32') == 'bf16'
    y = kernels.LinearBNActFn.apply(x if x.dim() == 2 else x.reshape(-1, in_dim), w, b, gamma, beta,
                                    None if freeze else mm, None if freeze else mv, BN_EPSILON, BN_MOMENTUM, act, bf16,
                                    bufs, kernels.bn_source_of(x) or kernels.bn_cols_of(x), kernels.grad_sink_of(x),
                                    defer_apply if (defer_apply and x.dim() == 2) else False)
    src = kernels.take_last_bn_source()
    y = y.reshape(shape[:-1] + (units,))
    return kernels.tag_bn_source(y, src) if src is not None else y
  # no BatchNorm, or BatchNorm on the moving statistics (evaluation; in training the experts of the reference's MMoE /
  # DBMTL, whose MMOE layer is built without is_training): plain GEMM, then ONE bias + normalise + activation launch
  be = kernels.hip()
  if x.dim() == 2 and not use_bn and act == kernels.ACT_NONE and b is not None and torch.is_grad_enabled() and ctx.is_training and \
      getattr(be, 'tall_gemv', False) and x.shape[0] >= be.BN_IN_STAGING_MIN_ROWS:
    # a TALL plain projection (DIN's attention scores over B x L rows): the bias in the contraction's epilogue and its gradient
    # as a column sum, instead of a bias launch forward and three (column sums, merge, copy) backward
    return _linear(x, w, b, kernels.bn_source_of(x), kernels.grad_sink_of(x)).reshape(shape[:-1] + (units,))
  if x.dim() == 2:
    z = _linear(x, w, None, kernels.bn_source_of(x), kernels.grad_sink_of(x))  # (a pending BatchNorm apply behind x: LinearFn)
  else:
    kernels.finish_pending_bn(x)
    z = _linear(x.reshape(-1, in_dim), w, None)
  y = kernels.BNActFn.apply(z, b, gamma, beta, None if freeze else mm, None if freeze else mv, use_bn,
                            BN_EPSILON, BN_MOMENTUM, act, training, _grad_bufs(b, gamma, beta))
  return y.reshape(shape[:-1] + (units,))


def din_first_layer(q, hist, units, name, l2_reg, act_relu, training, defer_apply=False):
  """dense + BatchNorm(train) + activation over DIN's attention input [q, h, q - h, q * h] ([B, L, 4E]) WITHOUT building it
  (kernels.DINFirstLayerFn; reference model/multi_tower_din.py:62-80, layers/dnn.py:57-79; the same variables: <name>/kernel
  [4E, units], /bias, /bn).  -> [B * L, units]."""
  ctx = context.current()
  vs = ctx.varstore
  B, L, E = hist.shape
  w = vs.get_variable(name + '/kernel', (4 * E, units), 'glorot_uniform', l2=l2_reg or 0.0)
  b = vs.get_variable(name + '/bias', (units,), 'zeros')
  bn = name + '/bn'
  gamma = vs.get_variable(bn + '/gamma', (units,), 'ones')
  beta = vs.get_variable(bn + '/beta', (units,), 'zeros')
  mm = vs.get_variable(bn + '/moving_mean', (units,), 'zeros', trainable=False)
  mv = vs.get_variable(bn + '/moving_variance', (units,), 'ones', trainable=False)
  freeze = ctx.building and training
  bufs = (w.grad, gamma.grad, beta.grad) if (w.grad is not None and gamma.grad is not None and beta.grad is not None) else None
  act = kernels.ACT_RELU if act_relu else kernels.ACT_NONE
  y = kernels.DINFirstLayerFn.apply(q, hist, w, b, gamma, beta, None if freeze else mm, None if freeze else mv, BN_EPSILON,
                                    BN_MOMENTUM, act, bufs, bool(defer_apply))
  src = kernels.take_last_bn_source()
  return kernels.tag_bn_source(y, src) if src is not None else y


def din_first_layer_ok(din_layer, q, hist):
  """May the attention DNN's first layer take (q, history) instead of the built [q, h, q - h, q * h] input?  A training step
  in fp32 whose first layer is dense -> BatchNorm -> ReLU (or no activation), no dropout, on shapes er_din_gemm_* take."""
  ctx = context.current()
  be = kernels.hip()
  n = len(din_layer.hidden_units)
  if not (getattr(be, 'din_fused', False) and din_layer._is_training and torch.is_grad_enabled() and ctx.is_training and
          not ctx.building and getattr(ctx, 'dense_dtype', 'f32') == 'f32' and n > 1 and din_layer.hidden_units[0] > 0 and
          din_layer._config.use_bn and (is_relu(din_layer._act_string) or din_layer.activation is None) and
          not (len(din_layer.dropout_ratio) > 0 and din_layer.dropout_ratio[0] > 0)):
    return False
  vs = ctx.varstore
  E = hist.shape[-1]
  w = vs.get_variable('%s/dnn_0/kernel' % din_layer._name, (4 * E, din_layer.hidden_units[0]), 'glorot_uniform',
                      l2=din_layer._l2_reg or 0.0)
  return w.grad is not None and be.din_gemm_ok(q, hist, w)


def _grad_bufs(b, gamma, beta):
  """(bias.grad, gamma.grad, beta.grad) when the variables are packed into the flat gradient buffer."""
  bufs = tuple(None if t is None else t.grad for t in (b, gamma, beta))
  present = [t for t in (b, gamma, beta) if t is not None]
  if not present or any(t.grad is None for t in present):
    return None
  return bufs


def batch_norm(x, name, training):
  """tf.layers.batch_normalization(x, training=..., name=name) on the last axis (TF defaults:
  momentum 0.99, epsilon 1e-3, gamma ones, beta zeros): reference model/multi_tower_din.py:105-109."""
  ctx = context.current()
  vs = ctx.varstore
  units = x.shape[-1]
  gamma = vs.get_variable(name + '/gamma', (units,), 'ones')
  beta = vs.get_variable(name + '/beta', (units,), 'zeros')
  mm = vs.get_variable(name + '/moving_mean', (units,), 'zeros', trainable=False)
  mv = vs.get_variable(name + '/moving_variance', (units,), 'ones', trainable=False)
  shape = x.shape
  x2 = x.reshape(-1, units)
  freeze = ctx.building and training
  y = kernels.BNActFn.apply(x2, None, gamma, beta, None if freeze else mm, None if freeze else mv, True,
                            BN_EPSILON, BN_MOMENTUM, kernels.ACT_NONE, training, _grad_bufs(None, gamma, beta))
  return y.reshape(shape)


def dense_parallel(items):
  """[(x, units, name, l2_reg)] plain dense layers (bias, no activation) over 2-D inputs - the towers' output projections
  of a multi-task model (reference model/mmoe.py:56-68: one tf.layers.dense per tower) - as ONE grouped launch forward and
  ONE for the input gradients (kernels.GroupedLinearFn); anything the grouped form does not cover: dense() one by one."""
  ctx = context.current()
  ok = torch.is_grad_enabled() and ctx.is_training and getattr(ctx, 'dense_dtype', 'f32') == 'f32' and \
      getattr(kernels.hip(), 'grouped_stacks', False) and len(items) > 1 and all(x.dim() == 2 for x, _, _, _ in items)
  if not ok:
    return [dense(x, units, name, l2_reg=l2) for x, units, name, l2 in items]
  vs = ctx.varstore
  xs = [x for x, _, _, _ in items]
  ws = [vs.get_variable(name + '/kernel', (x.shape[-1], units), 'glorot_uniform', l2=l2 or 0.0) for x, units, name, l2 in items]
  bs = [vs.get_variable(name + '/bias', (units,), 'zeros') for _, units, name, _ in items]
  E = len(items)
  out = kernels.GroupedLinearFn.apply(E, (False,) * E, tuple(kernels.grad_sink_of(x) for x in xs),
                                      tuple(kernels.bn_source_of(x) for x in xs), *xs, *ws, *bs)
  return list(out[:E])


def run_parallel(stacks, inputs, extra_dense=()):
  """E DNN stacks over (possibly the same) 2-D inputs, LAYER BY LAYER: the same-depth dense layers of all stacks run as
  one grouped launch (kernels.GroupedLinearFn), each followed by its own bias / BatchNorm / activation kernel - the
  variables, the arithmetic and the results are those of calling the stacks one after the other (the reference's order:
  layers/mmoe.py:62-83, model/multi_task_model.py:33-100), only the launch count differs (MMoE 4 tasks: 40 forward GEMM
  launches -> 8).  extra_dense: [(x, units, name, l2_reg)] plain dense layers that join the first depth's launch (MMoE's
  gates).  Returns ([stack outputs], [extra outputs]).  Anything the lock-step form does not cover (dropout, non-ReLU
  activations, per-layer outputs, unequal depths, evaluation) falls back to the sequential calls."""
  ctx = context.current()
  ok = torch.is_grad_enabled() and getattr(ctx, 'dense_dtype', 'f32') == 'f32' and \
      getattr(kernels.hip(), 'grouped_stacks', False) and len(stacks) > 1
  depth = len(stacks[0].hidden_units) if stacks else 0
  for d, x in zip(stacks, inputs):
    ok = ok and x.dim() == 2 and len(d.hidden_units) == depth and not (depth == 1 and d.hidden_units[0] == 0) and \
        not (len(d.dropout_ratio) > 0 and d._is_training and any(r > 0 for r in d.dropout_ratio)) and \
        (is_relu(d._act_string) or d.activation is None)
  if not ok:
    return [d(x) for d, x in zip(stacks, inputs)], [dense(x, units, name, l2_reg=l2) for x, units, name, l2 in extra_dense]
  vs = ctx.varstore
  cur = list(inputs)
  extras = []
  for i in range(depth):
    xs, ws, bs, metas = [], [], [], []
    for d, x in zip(stacks, cur):
      unit, layer = d.hidden_units[i], '%s/dnn_%d' % (d._name, i)
      use_bn = d._config.use_bn and ((i + 1 < depth) or not d._last_layer_no_batch_norm)
      use_act = (i + 1 < depth) or not d._last_layer_no_activation
      w = vs.get_variable(layer + '/kernel', (x.shape[-1], unit), 'glorot_uniform', l2=d._l2_reg or 0.0)
      b = vs.get_variable(layer + '/bias', (unit,), 'zeros')
      gamma = beta = mm = mv = None
      if use_bn:
        gamma = vs.get_variable(layer + '/bn/gamma', (unit,), 'ones')
        beta = vs.get_variable(layer + '/bn/beta', (unit,), 'zeros')
        mm = vs.get_variable(layer + '/bn/moving_mean', (unit,), 'zeros', trainable=False)
        mv = vs.get_variable(layer + '/bn/moving_variance', (unit,), 'ones', trainable=False)
      xs.append(x)
      ws.append(w)
      # the bias: in the GEMM's epilogue where batch statistics follow (they must see it; its gradient is zero under
      # BatchNorm), else left to the bias + normalise + activate kernel, whose backward also yields the bias gradient -
      # no separate column sums either way
      bs.append(b if (use_bn and d._is_training) else None)
      metas.append((use_bn, use_act and is_relu(d._act_string), d._is_training, gamma, beta, mm, mv, b))
    if i == 0:
      for x, units, name, l2 in extra_dense:
        xs.append(x)
        ws.append(vs.get_variable(name + '/kernel', (x.shape[-1], units), 'glorot_uniform', l2=l2 or 0.0))
        bs.append(vs.get_variable(name + '/bias', (units,), 'zeros'))
        metas.append(None)
    E = len(xs)
    train_bn = [m is not None and bool(m[0]) and m[2] for m in metas]  # batch statistics: from the GEMM's epilogue
    sinks = tuple(kernels.grad_sink_of(x) for x in xs)
    srcs = tuple(kernels.bn_source_of(x) for x in xs)
    be = kernels.hip()
    lay = [e for e, m in enumerate(metas) if m is not None]
    multi_bn = getattr(be, 'grouped_bn', False) and len(lay) > 1 and all(xs[e].shape[0] <= be.BN_MULTI_MAX_ROWS for e in lay)
    # layers on the MOVING statistics (the experts of the reference's MMoE: batch_normalization(training=False) in training):
    # bias + normalise + activate inside the contraction's epilogue - the launch writes z and y (er_gemm_problem.fz_*)
    fzs = None
    if multi_bn and getattr(be, 'frozen_bn_epilogue', False) and not ctx.building:
      fzs = [None] * E
      for e in lay:
        use_bn, relu, training, gamma, beta, mm, mv, bias = metas[e]
        if use_bn and not training and not train_bn[e] and bs[e] is None:
          fzs[e] = dict(bias=bias.detach(), gamma=gamma.detach(), beta=beta.detach(), moving_mean=mm, moving_var=mv,
                        eps=BN_EPSILON, act=kernels.ACT_RELU if relu else kernels.ACT_NONE)
      if not any(f is not None for f in fzs):
        fzs = None
    if fzs is not None:
      out = kernels.GroupedLinearFn.apply(E, tuple(train_bn), sinks, srcs, *xs, *ws, *bs, fzs)
      pres = [(out[2 * E + 2 * e], out[2 * E + 2 * e + 1]) if fzs[e] is not None else None for e in range(E)]
    else:
      out = kernels.GroupedLinearFn.apply(E, tuple(train_bn), sinks, srcs, *xs, *ws, *bs)
      pres = None
    zs, stats = out[:E], out[E:2 * E]
    if multi_bn:
      # the bias / BatchNorm / activation kernels of the depth as one launch (kernels.GroupedBNActFn)
      cfgs, a_b, a_g, a_be = [], [], [], []
      for e in lay:
        use_bn, relu, training, gamma, beta, mm, mv, bias = metas[e]
        freeze = ctx.building and training  # build pass: do not touch the moving statistics
        mode = kernels.BN_NONE if not use_bn else (kernels.BN_BATCH if training else kernels.BN_FROZEN)
        eb = None if train_bn[e] else bias  # (batch statistics: the bias went into the GEMM's epilogue)
        gb = tuple(None if t is None else t.grad for t in (eb, gamma, beta))
        if any(t is not None and t.grad is None for t in (eb, gamma, beta)):
          gb = None
        cfgs.append((mode, kernels.ACT_RELU if relu else kernels.ACT_NONE, None if freeze else mm, None if freeze else mv,
                     BN_EPSILON, BN_MOMENTUM, gb))
        a_b.append(eb)
        a_g.append(gamma)
        a_be.append(beta)
      extra = [pres[e] for e in lay] if pres is not None else []
      ys = kernels.GroupedBNActFn.apply(len(lay), tuple(cfgs), *[zs[e] for e in lay],
                                        *[stats[e] if train_bn[e] else None for e in lay], *a_b, *a_g, *a_be, *extra)
      owns = kernels.take_last_bn_source() or [None] * len(lay)
      for e in range(E):
        if metas[e] is None:
          extras.append(zs[e])
      cur = [kernels.tag_bn_source(y, o) if o is not None else y for y, o in zip(ys, owns)]
      continue
    nxt = []
    for e, m in enumerate(metas):
      if m is None:
        extras.append(zs[e])
        continue
      use_bn, relu, training, gamma, beta, mm, mv, bias = m
      act = kernels.ACT_RELU if relu else kernels.ACT_NONE
      freeze = ctx.building and training  # build pass: do not touch the moving statistics
      if train_bn[e]:
        gb = (gamma.grad, beta.grad) if (gamma.grad is not None and beta.grad is not None) else None
        y = kernels.BNFromStatsFn.apply(zs[e], stats[e], gamma, beta, None if freeze else mm, None if freeze else mv,
                                        BN_EPSILON, BN_MOMENTUM, act, gb)
        src = kernels.take_last_bn_source()
        if src is not None:
          y = kernels.tag_bn_source(y, src)
      elif use_bn or relu:
        # BatchNorm on the moving statistics (the experts of the reference's MMoE) and / or ReLU: one launch
        y = kernels.BNActFn.apply(zs[e], bias, gamma, beta, None if freeze else mm, None if freeze else mv, use_bn,
                                  BN_EPSILON, BN_MOMENTUM, act, training, _grad_bufs(bias, gamma, beta))
      else:
        y = kernels.BNActFn.apply(zs[e], bias, None, None, None, None, 0, BN_EPSILON, BN_MOMENTUM, kernels.ACT_NONE,
                                  training, _grad_bufs(bias, None, None))
      nxt.append(y)
    cur = nxt
  return cur, extras


class DNN(object):

  def __init__(self, dnn_config, l2_reg, name='dnn', is_training=False, last_layer_no_activation=False,
               last_layer_no_batch_norm=False):
    self._config = dnn_config
    self._l2_reg = l2_reg
    self._name = name
    self._is_training = is_training
    logging.info('dnn activation function = %s' % self._config.activation)
    self._act_string = self._config.activation
    self.activation = get_activation(self._config.activation, training=is_training) \
        if self._config.activation.lower() == 'dice' else get_activation(self._config.activation)
    self._last_layer_no_activation = last_layer_no_activation
    self._last_layer_no_batch_norm = last_layer_no_batch_norm

  @property
  def hidden_units(self):
    return self._config.hidden_units

  @property
  def dropout_ratio(self):
    return self._config.dropout_ratio

  def __call__(self, deep_fea, hidden_layer_feature_output=False, din=None, defer_last_apply=False):
    """din = (query [B, E], history [B, L, E]): the input is DIN's [q, h, q - h, q * h] ([B, L, 4E]), never built - the first
    layer generates it inside its contractions (din_first_layer; the caller checked din_first_layer_ok); deep_fea is
    ignored."""
    hidden_units_len = len(self.hidden_units)
    if hidden_units_len == 1 and self.hidden_units[0] == 0:
      return deep_fea
    hidden_feature_dict = {}
    lead_shape = None
    if din is not None:
      assert not hidden_layer_feature_output
      lead_shape = din[1].shape[:-1]
    if din is None and deep_fea.dim() > 2 and not hidden_layer_feature_output:
      # [B, L, d] inputs (DIN's attention MLP): the whole stack runs on the flattened [B * L, d] view - BatchNorm
      # normalises over every axis but the last either way
      lead_shape = deep_fea.shape[:-1]
      deep_fea = deep_fea.reshape(-1, deep_fea.shape[-1])
    for i, unit in enumerate(self.hidden_units):
      layer = '%s/dnn_%d' % (self._name, i)
      use_bn = self._config.use_bn and ((i + 1 < hidden_units_len) or not self._last_layer_no_batch_norm)
      use_act = (i + 1 < hidden_units_len) or not self._last_layer_no_activation
      fuse_relu = use_act and is_relu(self._act_string)
      last = i + 1 == hidden_units_len
      no_drop = not (len(self.dropout_ratio) > 0 and self._is_training and self.dropout_ratio[i] > 0)
      # a TALL layer (DIN's attention MLP over B x L rows) hands its BatchNorm apply to the NEXT layer's contraction, which
      # runs it while staging (kernels.LinearBNActFn / LinearFn -> HipBackend.gemm_bn_a): when nothing between the two reads
      # the activations
      rows = (din[1].shape[0] * din[1].shape[1]) if (i == 0 and din is not None) else \
          (deep_fea.shape[0] if deep_fea.dim() == 2 else 0)
      to_next = bool(not last and use_bn and self._is_training and torch.is_grad_enabled() and self._config.use_bn and
                     (fuse_relu or not use_act or self.activation is None) and no_drop and not hidden_layer_feature_output and
                     getattr(kernels.hip(), 'bn_in_staging', False) and rows >= kernels.hip().BN_IN_STAGING_MIN_ROWS and
                     getattr(context.current(), 'dense_dtype', 'f32') == 'f32' and context.current().is_training)
      if i == 0 and din is not None:
        deep_fea = din_first_layer(din[0], din[1], unit, layer, self._l2_reg, fuse_relu, self._is_training, defer_apply=to_next)
      else:
        # (defer_last_apply: see dense_bn_act - only when nothing below touches the layer's output again)
        defer = bool(defer_last_apply and last and use_bn and (fuse_relu or not use_act) and not hidden_layer_feature_output and
                     lead_shape is None and no_drop)
        if to_next:
          defer = 'staging'
        deep_fea = dense_bn_act(deep_fea, unit, layer, self._l2_reg, True, use_bn, fuse_relu, self._is_training,
                                defer_apply=defer)
      if use_act and not fuse_relu and self.activation is not None:
        deep_fea = self.activation(deep_fea, name='%s/dnn_%d/act' % (self._name, i))
      if len(self.dropout_ratio) > 0 and self._is_training:
        assert self.dropout_ratio[i] < 1, 'invalid dropout_ratio: %.3f' % self.dropout_ratio[i]
        if self.dropout_ratio[i] > 0:
          deep_fea = torch.nn.functional.dropout(deep_fea, p=self.dropout_ratio[i], training=True)
      if hidden_layer_feature_output:
        hidden_feature_dict['hidden_layer' + str(i)] = deep_fea
        if i + 1 == hidden_units_len:
          hidden_feature_dict['hidden_layer_end'] = deep_fea
          return hidden_feature_dict
    if lead_shape is not None:
      deep_fea = deep_fea.reshape(tuple(lead_shape) + (deep_fea.shape[-1],))
    return deep_fea

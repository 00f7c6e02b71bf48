"""ctypes binding of libeasyrec_hip.so (include/easyrec_hip.h) at tensor level.

The reference issues this work as TensorFlow ops from Python (file:line cited per entry point in
include/easyrec_hip.h); here the same Python call sites call hand-written gfx950 kernels.

`hip()` returns the singleton `HipBackend`.  It fails loudly (RuntimeError) when the shared
library has not been built or no MI355X is visible - there is no CPU fallback in this package.
"""
import ctypes
import threading
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch


class KvJob(ctypes.Structure):
  """er_kv_job (include/easyrec_hip.h)"""
  _fields_ = [('ids', ctypes.c_void_p), ('n', ctypes.c_int64), ('map_keys', ctypes.c_void_p), ('map_rows', ctypes.c_void_p),
              ('map_slots', ctypes.c_int64), ('next_row', ctypes.c_void_p), ('var', ctypes.c_void_p), ('seed', ctypes.c_uint64),
              ('rows_out', ctypes.c_void_p), ('overflow', ctypes.c_void_p), ('capacity', ctypes.c_int32), ('dim', ctypes.c_int32),
              ('init_mean', ctypes.c_float), ('init_stddev', ctypes.c_float), ('n_limit', ctypes.c_void_p),
              ('freq', ctypes.c_void_p), ('version', ctypes.c_void_p), ('n_keys', ctypes.c_void_p), ('step', ctypes.c_void_p),
              ('filter_freq', ctypes.c_int32), ('table_ld', ctypes.c_int32)]


class KvRouteJob(ctypes.Structure):
  """er_kv_route_job (include/easyrec_hip.h)"""
  _fields_ = [('ids', ctypes.c_void_p), ('n', ctypes.c_int64), ('n_limit', ctypes.c_void_p), ('rows_out', ctypes.c_void_p),
              ('slot', ctypes.c_void_p), ('send_off', ctypes.c_int64)]


class CastDesc(ctypes.Structure):
  """er_cast_desc (include/easyrec_hip.h)"""
  _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('rows', ctypes.c_int64), ('cols', ctypes.c_int32),
              ('ld_src', ctypes.c_int32), ('ld_dst', ctypes.c_int32), ('transpose', ctypes.c_int32)]

_HERE = os.path.dirname(os.path.abspath(__file__))
# EASYREC_AMD_LIB: another build of the same ABI (same-box A/B of a kernel change, tools/gpu_ab.sh)
LIB_PATH = os.environ.get('EASYREC_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libeasyrec_hip.so')

COMBINER_SUM, COMBINER_MEAN, COMBINER_SQRTN = 0, 1, 2
COMBINERS = {'sum': COMBINER_SUM, 'mean': COMBINER_MEAN, 'sqrtn': COMBINER_SQRTN}
OPT_SGD, OPT_ADAM, OPT_LAZY_ADAM, OPT_ADAGRAD = 0, 1, 2, 3
ACT_NONE, ACT_RELU = 0, 1
BN_NONE, BN_BATCH, BN_FROZEN = 0, 1, 2  # er_bn_act_fwd / _bwd use_bn (include/easyrec_hip.h)
GEMM_NN, GEMM_NT, GEMM_TN = 0, 1, 2


class GemmEpilogue(ctypes.Structure):  # = er_gemm_epilogue
  _fields_ = [('kind', ctypes.c_int32), ('diag', ctypes.c_float), ('col_stats', ctypes.c_void_p),
              ('bn_z', ctypes.c_void_p), ('bn_zbias', ctypes.c_void_p), ('bn_y', ctypes.c_void_p), ('bn_mean', ctypes.c_void_p),
              ('bn_invstd', ctypes.c_void_p), ('bn_partial', ctypes.c_void_p),
              ('bn_ld', ctypes.c_int32), ('bn_use_bn', ctypes.c_int32), ('bn_act', ctypes.c_int32), ('bn_col0', ctypes.c_int32),
              ('bn_n_src', ctypes.c_int32),
              ('ld_x0', ctypes.c_int32), ('ld_xl', ctypes.c_int32), ('ld_u', ctypes.c_int32), ('ld_dout', ctypes.c_int32),
              ('ld_du_in', ctypes.c_int32), ('ld_prev_u', ctypes.c_int32), ('ld_dx0', ctypes.c_int32),
              ('ld_du_out', ctypes.c_int32), ('ld_du_out_bf16', ctypes.c_int32), ('accumulate_dx0', ctypes.c_int32),
              ('x0', ctypes.c_void_p), ('xl', ctypes.c_void_p), ('u', ctypes.c_void_p), ('dout', ctypes.c_void_p),
              ('du_in', ctypes.c_void_p), ('prev_u', ctypes.c_void_p), ('prev_bias', ctypes.c_void_p), ('dx0', ctypes.c_void_p),
              ('du_out', ctypes.c_void_p), ('du_out_bf16', ctypes.c_void_p), ('partial', ctypes.c_void_p)]


EPI_PLAIN, EPI_STATS, EPI_BN_BWD, EPI_CROSS_FWD, EPI_CROSS_BWD = 0, 1, 2, 3, 4


class LookupDesc(ctypes.Structure):
  """Mirror of `er_lookup_desc` (include/easyrec_hip.h)."""
  _fields_ = [
      ('table', ctypes.c_void_p),
      ('ids', ctypes.c_void_p),
      ('offsets', ctypes.c_void_p),
      ('weights', ctypes.c_void_p),
      ('out', ctypes.c_void_p),
      ('rows', ctypes.c_int64),
      ('key_base', ctypes.c_int64),
      ('dim', ctypes.c_int32),
      ('out_stride', ctypes.c_int32),
      ('out_col', ctypes.c_int32),
      ('combiner', ctypes.c_int32),
      ('n_rows', ctypes.c_int32),
      ('max_nnz', ctypes.c_int32),
      ('table_ld', ctypes.c_int32),
  ]


# er_opt_hyper: 16 floats
HYPER_FLOATS = 16
HYPER_LR, HYPER_LR_T, HYPER_BETA1, HYPER_BETA2, HYPER_OMB1, HYPER_OMB2, HYPER_EPS, HYPER_GSCALE = range(8)
HYPER_CLIP = 8  # er_opt_hyper.clip_scale: 0 = no clipping


_wgrad_tls = threading.local()
_bn_tls = threading.local()


def take_last_bn_source():
  """The BnSource of the LinearBNActFn that just ran on this thread (None when the fused path is off)."""
  src, _bn_tls.last = getattr(_bn_tls, 'last', None), None
  return src


class WgradSink(object):
  """Weight gradients queued during one backward pass (HipBackend.defer_wgrads / flush_wgrads)."""

  def __init__(self):
    self.active = False
    self.queue = []
    self.queue_bf16 = []  # (a bf16 step's weight gradients: er_gemm_grouped_bf16)
    # bias gradients left as per-row-tile column sums by the fused cross backward [(partial [T, ld], dst [n], n)]:
    # finished by ONE launch when the backward pass is over (HipBackend.take_wgrads -> colsum_partials_multi)
    self.colsum_jobs = []

  # a bf16 step's weight gradients dW = x^T . dz contract over the BATCH: their operands are k-strided in HBM, which the bf16
  # kernel with operands in HBM (er_gemm_bf16_nt) does not take, and er_gemm_grouped_bf16 rounds fp32 operands while staging
  # them (122 us per DCN-v2 step for what the fp32 launch inside the step's fused tail does in the shadow of the embedding
  # update).  Default: the fp32 contraction of the fp32 activations / gradients the step holds anyway - the weight
  # gradient is then the one quantity of a bf16 step computed at HIGHER precision; '0' = er_gemm_grouped_bf16.
  bf16_wgrad_f32 = os.environ.get('EASYREC_AMD_BF16_WGRAD_F32', '1') != '0'

  def put(self, x, dy, out, bf16):
    if not self.active:
      return False
    (self.queue_bf16 if (bf16 and not self.bf16_wgrad_f32) else self.queue).append((x, dy, out, None, True))
    return True


class BnSource(object):
  """What the dgrad GEMM of a consumer needs in order to emit, in its epilogue, the BatchNorm-backward column sums
  of the layer that produced its input (HipBackend.gemm_bn_bwd): that layer's z, y and batch statistics."""
  __slots__ = ('z', 'zbias', 'y', 'mean', 'invstd', 'act', 'partial', 'dx_ptr', 'gamma', 'grad_bufs', 'beta', 'fused', 'pending',
               'frozen', 'dz_done')

  def __init__(self, z, zbias, y, mean, invstd, act, gamma=None, grad_bufs=None, beta=None, fused=True):
    assert y is not None
    self.z, self.zbias, self.y, self.mean, self.invstd, self.act = z, zbias, y, mean, invstd, act
    self.partial, self.dx_ptr = None, 0
    self.gamma, self.grad_bufs = gamma, grad_bufs  # grad_bufs = (gamma.grad, beta.grad) slices of the flat buffer
    # fused: the consumer's dgrad GEMM may emit this layer's BatchNorm-backward column sums (HipBackend.fused_bn_bwd)
    self.beta, self.fused = beta, fused
    # pending: the layer's BatchNorm finalize + apply has NOT been launched yet (LinearBNActFn(defer_apply=True)): y, mean,
    # invstd are unwritten until the ONE consumer the model promised runs it (WideFmConcatFn: er_bn_apply_wide_fm) or
    # finish_pending_bn does; holds the arguments of HipBackend.bn_apply_from_stats
    self.pending = None
    # frozen: the layer normalises with the MOVING statistics (its backward is elementwise); dz_done: the contraction that
    # produced this layer's dy has already turned it into dz (er_gemm_problem.bn_dz_out, GroupedLinearFn.backward)
    self.frozen, self.dz_done = False, False


class BnColsView(object):
  """`src`'s layer output as the column block [col0, col0 + width) of a wider tensor (DeepFM's [sum(wide) | FM | deep]):
  the consumer's dgrad GEMM emits that layer's BatchNorm-backward column sums from those columns of its output
  (HipBackend.gemm_bn_bwd(col0=...))."""
  __slots__ = ('src', 'col0', 'ref')
  fused = True  # (what LinearBNActFn.forward asks of a BnSource)

  def __init__(self, src, col0, ref):
    self.src, self.col0, self.ref = src, int(col0), ref


def tag_bn_cols(joined, part, col0):
  """`joined[:, col0:col0 + part.shape[1]]` is a copy of `part`, the output of a fused dense + BatchNorm layer."""
  src = bn_source_of(part)
  if src is not None and src.fused:
    joined._er_bn_cols = BnColsView(src, col0, joined)
  return joined


def bn_cols_of(x):
  """The BnColsView of x if x IS the untouched tensor tag_bn_cols marked (dense + BatchNorm layers take it as `src`)."""
  if not getattr(hip(), 'bn_cols_epilogue', False):
    return None
  v = getattr(x, '_er_bn_cols', None)
  if v is None or x.dim() != 2 or x.data_ptr() != v.ref.data_ptr() or x.shape != v.ref.shape or x.stride() != v.ref.stride():
    return None
  return v


def tag_bn_source(y, src):
  """Mark `y` (what the layer function returns) as the output described by src."""
  y._er_bn_src = src
  return y


def grad_sink_of(x, col0=0, width=None):
  """The GradSink of x if x IS a group output of the embedding engine (layers/input_layer.py) and a deposit into
  columns [col0, col0 + width) is allowed now, else None."""
  sink = getattr(x, '_er_sink', None)
  if sink is None or x.dim() != 2:
    return None
  return sink


def grad_slots_of_step():
  """The step's gradient-slot table (kernels.grad_slot), or None: taken in a Function's FORWARD and kept on its ctx - the
  backward may run on an autograd worker thread, where the thread-local model context is not set."""
  from easyrec_amd.core import context
  stack = context._stack()
  return getattr(stack[-1], 'grad_slots', None) if (stack and _GRAD_SLOTS) else None


_GRAD_SLOTS = os.environ.get('EASYREC_AMD_GRAD_SLOTS', '1') != '0'  # A/B switch
_CAT_DGRAD = os.environ.get('EASYREC_AMD_CAT_DGRAD', '1') != '0'    # A/B switch: one input-gradient GEMM for the readers of a shared input


class _SlotGateFn(torch.autograd.Function):
  """Identity.  Every SLOT-AWARE consumer of an activation reads it through ONE gate per step (slot_gate), so the gate's
  backward - which autograd runs only after all of THEM - is where their shared gradient buffer enters x's gradient."""

  @staticmethod
  def forward(ctx, x):
    # (when every consumer deposited its gradient elsewhere - an embedding group's GradSink - nothing arrives here: without
    # this autograd would materialise a zero tensor and the group would add it to its buffer: a fill + an axpy2d for nothing)
    ctx.set_materialize_grads(False)
    return x.view_as(x)

  @staticmethod
  def backward(ctx, g):
    return g


def slot_gate(x):
  """The tensor a slot-aware Function (LinearFn, CrossV2EpilogueFn, DINConcatFn, DINPoolFn) must be APPLIED to instead of
  x.  grad_slot lets those consumers accumulate their input gradients into one buffer that the first of them returns to
  autograd.  If that buffer went to x directly, an ordinary consumer of x (a torch op, LinearBNActFn, ...) whose gradient
  arrives between two slot consumers would make autograd sum out of place - buffer + other - and the later slot consumers
  would add into a buffer nobody reads any more.  Behind the gate the buffer is the ONLY gradient autograd sees for the gate's
  output, complete when the gate's backward hands it on to x, whatever else reads x and in whatever order."""
  slots = grad_slots_of_step()
  if slots is None or not torch.is_grad_enabled() or not x.requires_grad or x.dim() < 2:
    return x
  gates = slots.setdefault('gates', {})
  key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()))
  ent = gates.get(key)
  if ent is not None and (ent[0] is x or ent[1] is x):
    return ent[1]
  if ent is not None and x._base is not None and ent[0]._base is x._base and ent[0]._version == x._version and \
      ent[0].dtype == x.dtype:
    # another VIEW OBJECT of the same activation with the same geometry (dense() on a 3-D input reshapes it on every call):
    # its consumers must share the first view's gate - grad_slot keys the shared buffer by storage start and shape, and two
    # gates over one buffer would let the first hand it to autograd while the second's consumers are still adding into it.
    # The gradient reaches the common base through the first view's node: the same values.
    return ent[1]
  xg = _SlotGateFn.apply(x)
  for attr in ('_er_bn_src', '_er_sink'):
    if hasattr(x, attr):
      setattr(xg, attr, getattr(x, attr))
  gates[key] = (x, xg)
  return xg


def grad_slot(slots, x):
  """One gradient buffer per activation tensor and step, shared by the backward kernels of ALL its slot-aware consumers: the
  first to ask gets (a fresh tensor, accumulate = False, first = True) and returns that tensor to autograd; the others get
  (the same tensor, True, False), ADD into it inside their own kernels and return None - autograd then has one gradient for
  x and launches no add kernel.  The consumers must have been applied to slot_gate(x): see there.  Keyed by storage start
  and shape; the table (grad_slots_of_step) is emptied by EasyRecModel.begin_step."""
  key = (x.data_ptr(), tuple(x.shape))
  t = slots.get(key)
  if t is not None:
    return t, True, False
  t = torch.empty(x.shape, dtype=torch.float32, device=x.device)
  slots[key] = t
  return t, False, True


def bn_source_of(x):
  """The BnSource of x if x IS the untouched 2-D output of a fused dense + BatchNorm + activation layer."""
  src = getattr(x, '_er_bn_src', None)
  if src is None:
    return None
  ref = src.y
  if x.dim() != 2 or x.data_ptr() != ref.data_ptr() or x.shape != ref.shape or x.stride() != ref.stride():
    return None
  return src


def _ptr(t):
  return None if t is None else t.data_ptr()


class GemmProblem(ctypes.Structure):  # = er_gemm_problem
  _fields_ = [('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('A', ctypes.c_void_p),
              ('lda', ctypes.c_int32), ('B', ctypes.c_void_p), ('ldb', ctypes.c_int32), ('C', ctypes.c_void_p),
              ('ldc', ctypes.c_int32), ('bias', ctypes.c_void_p), ('accumulate', ctypes.c_int32),
              ('col_stats', ctypes.c_void_p),
              ('bn_z', ctypes.c_void_p), ('bn_zbias', ctypes.c_void_p), ('bn_y', ctypes.c_void_p),
              ('bn_mean', ctypes.c_void_p), ('bn_invstd', ctypes.c_void_p), ('bn_ld', ctypes.c_int32),
              ('bn_use_bn', ctypes.c_int32), ('bn_act', ctypes.c_int32), ('bn_partial', ctypes.c_void_p),
              ('fz_bias', ctypes.c_void_p), ('fz_gamma', ctypes.c_void_p), ('fz_beta', ctypes.c_void_p),
              ('fz_mean', ctypes.c_void_p), ('fz_var', ctypes.c_void_p), ('fz_eps', ctypes.c_float), ('fz_act', ctypes.c_int32),
              ('fz_y', ctypes.c_void_p), ('fz_save', ctypes.c_void_p), ('bn_gamma', ctypes.c_void_p), ('bn_dz_out', ctypes.c_int32)]


class BnLayer(ctypes.Structure):  # = er_bn_layer
  _fields_ = [('x', ctypes.c_void_p), ('bias', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p),
              ('moving_mean', ctypes.c_void_p), ('moving_var', ctypes.c_void_p), ('col_stats', ctypes.c_void_p),
              ('chunks', ctypes.c_int32), ('B', ctypes.c_int32), ('N', ctypes.c_int32), ('use_bn', ctypes.c_int32),
              ('act', ctypes.c_int32), ('eps', ctypes.c_float), ('momentum', ctypes.c_float), ('y', ctypes.c_void_p),
              ('save_mean', ctypes.c_void_p), ('save_invstd', ctypes.c_void_p), ('y_in', ctypes.c_void_p),
              ('dy', ctypes.c_void_p), ('dy_ld', ctypes.c_int32), ('partial', ctypes.c_void_p), ('dx', ctypes.c_void_p),
              ('dbias', ctypes.c_void_p), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p),
              ('accumulate', ctypes.c_int32)]


class CeHead(ctypes.Structure):  # = er_ce_head
  _fields_ = [('logits', ctypes.c_void_p), ('labels', ctypes.c_void_p), ('weights', ctypes.c_void_p), ('B', ctypes.c_int32),
              ('loss_scale', ctypes.c_float), ('loss_out', ctypes.c_void_p), ('dlogits', ctypes.c_void_p),
              ('probs_out', ctypes.c_void_p)]


CROSS_HASH_KEY = 0xDECAFCAFFE  # tf.sparse.cross_hashed's default hash_key (crossed_column(hash_key=None))


class TailJob(ctypes.Structure):  # = er_tail_job
  _fields_ = [('partial', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('n_parts', ctypes.c_int32), ('n_cols', ctypes.c_int32),
              ('ld', ctypes.c_int32)]


class ColsumJob(ctypes.Structure):  # = er_colsum_job
  _fields_ = [('x', ctypes.c_void_p), ('rows', ctypes.c_int32), ('cols', ctypes.c_int32), ('x_stride', ctypes.c_int32),
              ('out', ctypes.c_void_p)]


class LossTailJob(ctypes.Structure):  # = er_loss_tail_job
  _fields_ = [('emb_partials', ctypes.c_void_p), ('n_partials', ctypes.c_int32), ('emb_scale', ctypes.c_float),
              ('dense_partials', ctypes.c_void_p), ('n_dense', ctypes.c_int32),
              ('losses', ctypes.c_void_p), ('report', ctypes.c_void_p), ('loss_parts', ctypes.c_void_p),
              ('loss_scales', ctypes.c_void_p), ('loss_divs', ctypes.c_void_p), ('loss_values', ctypes.c_void_p),
              ('n_losses', ctypes.c_int32), ('jobs', ctypes.c_void_p), ('n_jobs', ctypes.c_int32),
              ('reg_out', ctypes.c_void_p), ('total_out', ctypes.c_void_p)]


class DenseOptJob(ctypes.Structure):  # = er_dense_opt_job
  _fields_ = [('w', ctypes.c_void_p), ('m', ctypes.c_void_p), ('v', ctypes.c_void_p), ('grad', ctypes.c_void_p),
              ('l2coef', ctypes.c_void_p), ('n', ctypes.c_int64), ('opt_kind', ctypes.c_int32), ('hyper', ctypes.c_void_p),
              ('l2_partials', ctypes.c_void_p)]


class GradTerm(ctypes.Structure):  # = er_grad_term
  _fields_ = [('kind', ctypes.c_int32), ('col0', ctypes.c_int32), ('width', ctypes.c_int32), ('dim', ctypes.c_int32),
              ('g', ctypes.c_void_p), ('g_ld', ctypes.c_int32), ('pad_', ctypes.c_int32), ('saved', ctypes.c_void_p)]


class GradGroup(ctypes.Structure):  # = er_grad_group
  _fields_ = [('dout', ctypes.c_void_p), ('out', ctypes.c_void_p), ('ld', ctypes.c_int32), ('batch', ctypes.c_int32),
              ('width', ctypes.c_int32), ('has_base', ctypes.c_int32), ('n_terms', ctypes.c_int32),
              ('lam', ctypes.c_float), ('terms', GradTerm * 4)]


GRAD_TERM_ROWSUM, GRAD_TERM_FM = 0, 1


class DenseApplyDesc(ctypes.Structure):  # = er_dense_apply_desc
  _fields_ = [('var', ctypes.c_void_p), ('m', ctypes.c_void_p), ('v', ctypes.c_void_p), ('dense', ctypes.c_void_p),
              ('ld', ctypes.c_int32), ('dim', ctypes.c_int32), ('rows', ctypes.c_int64), ('table_ld', ctypes.c_int64)]


def same_lookup_keys(group, leader):
  """Do two table groups see the same keys every step (er_emb_group_share_sort's condition)?"""
  if group is leader or leader.get('sort_leader') is not None:
    return False
  a, b = group['specs'], leader['specs']
  if len(a) != len(b) or group.get('n_active', -1) != leader.get('n_active', -1):
    return False
  for k in ('world', 'shard_stride', 'local_base'):
    if group.get(k) != leader.get(k):
      return False

  def ptr(t):
    return None if t is None else t.data_ptr()

  for x, y in zip(a, b):
    if (ptr(x.ids), ptr(x.offsets), x.rows, x.key_base, x.n_rows, x.max_nnz) != \
        (ptr(y.ids), ptr(y.offsets), y.rows, y.key_base, y.n_rows, y.max_nnz):
      return False
  return True


@dataclass
class LookupSpec:
  """One embedding lookup (tensor-level view of er_lookup_desc).

  table:   [rows, dim] fp32 view of the lookup's table inside its table group
  ids:     int64 [n_rows] (dense mode, id < 0 = missing) or [max_nnz] (ragged mode)
  offsets: None or int32 [n_rows + 1]
  weights: None or fp32, aligned with ids
  out:     2-D fp32 base tensor [n_rows, out_stride]; forward output or upstream gradient
  """
  table: torch.Tensor
  ids: torch.Tensor
  offsets: Optional[torch.Tensor]
  weights: Optional[torch.Tensor]
  out: torch.Tensor
  out_col: int
  rows: int
  key_base: int
  dim: int
  combiner: int
  n_rows: int
  max_nnz: int
  name: str = ''

  def with_out(self, out):
    return LookupSpec(self.table, self.ids, self.offsets, self.weights, out, self.out_col,
                      self.rows, self.key_base, self.dim, self.combiner, self.n_rows,
                      self.max_nnz, self.name)


def _p(t):
  if t is None:
    return ctypes.c_void_p(0)
  return ctypes.c_void_p(t.data_ptr())


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name='tensor'):
  assert t.dtype == torch.float32 and t.is_contiguous(), '%s must be contiguous fp32' % name
  return t


class Bf16Shadows(object):
  """dense_dtype 'bf16' (BASELINE config 3): the bf16 operands the contractions read from HBM.

  Weights: two bf16 shadows per fp32 master in `varstore.flat` - `plain` [K, pad8(N)] (the k-contiguous operand of the
  input-gradient product dx = dy . W^T) and `t` [N, pad8(K)] (of the forward product x . W) - written by ONE er_cast_bf16
  launch after every dense optimizer step (HipBackend.dense_opt_step) and whenever the store's weights were replaced
  (VarStore.version).  Activations: cast by act() when a contraction needs them (their producers write fp32)."""

  def __init__(self, be, varstore):
    import weakref
    self.be = be
    self.vs = weakref.ref(varstore)
    flat = varstore.flat
    self.lo, self.hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    self.device = flat.device
    self.weights = {}  # data_ptr -> (master view [K, N], plain, t)
    self.seen_version = getattr(varstore, 'version', 0)
    # per step (begin_step): bf16 copies of fp32 activations / gradients that their PRODUCER wrote beside the fp32 tensor
    # (BatchNorm apply, the cross epilogue, concat, the BatchNorm backward: register) or that an earlier contraction of
    # the forward pass cast (act(cache=True)): (data_ptr, shape, stride) -> (the fp32 tensor, kept alive so that its
    # address cannot be handed out again within the step; its bf16 copy)
    self.copies = {}

  def begin_step(self):
    self.copies = {}

  @staticmethod
  def _key(x):
    return (x.data_ptr(), tuple(x.shape), tuple(x.stride()))

  def register(self, x, xb):
    self.copies[self._key(x)] = (x, xb)

  def new_copy(self, x):
    """An empty bf16 buffer [rows, pad8(cols)] for a producer to fill beside the fp32 matrix x (the padding columns are
    read as the k-tail of a contraction: zeroed here when there are any), registered as x's copy."""
    rows, cols = x.shape
    pc = self.pad8(cols)
    xb = torch.empty(rows, pc, dtype=torch.bfloat16, device=x.device) if pc == cols else \
        torch.zeros(rows, pc, dtype=torch.bfloat16, device=x.device)
    self.register(x, xb)
    return xb

  @staticmethod
  def pad8(n):
    return (int(n) + 7) // 8 * 8

  def owns(self, t):
    return self.lo <= t.data_ptr() < self.hi

  def alive(self):
    return self.vs() is not None

  def _descs(self, entries):
    return [d for w, plain, t in entries for d in ((w, plain, False), (w, t, True))]

  def weight(self, w):
    vs = self.vs()
    if vs is not None and getattr(vs, 'version', 0) != self.seen_version:
      self.refresh()
    e = self.weights.get(w.data_ptr())
    if e is None or e[0].shape != w.shape:
      K, N = w.shape
      plain = torch.zeros(K, self.pad8(N), dtype=torch.bfloat16, device=self.device)
      t = torch.zeros(N, self.pad8(K), dtype=torch.bfloat16, device=self.device)
      e = self.weights[w.data_ptr()] = (w.detach(), plain, t)
      self.be.cast_bf16(self._descs([e]))
    return e

  def refresh(self):
    vs = self.vs()
    if vs is not None:
      self.seen_version = getattr(vs, 'version', 0)
    if self.weights:
      self.be.cast_bf16(self._descs(self.weights.values()))

  def act(self, x, transpose=False, cache=False):
    """A bf16 copy of the fp32 matrix x ([rows, pad8(cols)], or its transpose [cols, pad8(rows)]): the one its producer
    registered, else a cast launch.  cache: x will not change any more within the step (a forward activation) - later
    contractions reuse this cast."""
    if not transpose:
      e = self.copies.get(self._key(x))
      if e is not None:
        return e[1]
    rows, cols = x.shape
    dst = torch.empty((cols, self.pad8(rows)) if transpose else (rows, self.pad8(cols)), dtype=torch.bfloat16,
                      device=x.device)
    self.be.cast_bf16([(x, dst, transpose)])
    if cache and not transpose:
      self.register(x, dst)
    return dst


class HipBackend(object):
  """Tensor-level wrappers.  Every method launches asynchronously on torch's current stream."""

  name = 'hip'

  def __init__(self):
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          'easyrec_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; '
          'g.build()"` or `make -C easyrec_amd/csrc`; there is no CPU fallback.' % LIB_PATH)
    self.lib = ctypes.CDLL(LIB_PATH)
    self.lib.er_last_error.restype = ctypes.c_char_p
    self.lib.er_emb_group_num_entries.restype = ctypes.c_int64
    if self.lib.er_abi_version() != 1:
      raise RuntimeError('easyrec_amd: ABI version mismatch in %s' % LIB_PATH)

  # -- measurement hook (bench.py): with `op_log` a list, every contraction appends (kernel name as rocprof prints it,
  # flops): one eager step gives the algorithmic work behind each GEMM kernel of the step's profile
  op_log = None
  _LAYOUT_TPL = {0: '<true, false>', 1: '<true, true>', 2: '<false, false>'}

  def _log_gemm(self, kernel, layout, M, N, K):
    if self.op_log is not None:
      self.op_log.append(('er::' + kernel + (self._LAYOUT_TPL[int(layout)] if layout is not None else ''),
                          2.0 * M * N * K))

  # -- plumbing
  def _ck(self, rc, what):
    if rc != 0:
      msg = self.lib.er_last_error()
      raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))

  @staticmethod
  def require_device():
    if not torch.cuda.is_available():
      raise RuntimeError('easyrec_amd: no HIP device visible; the training path needs an MI355X '
                         '(there is no CPU fallback).')

  def reserve_scratch(self, floats):
    self._ck(self.lib.er_reserve_scratch(ctypes.c_int64(int(floats))), 'er_reserve_scratch')

  def config_set(self, key, value):
    self._ck(self.lib.er_config_set(key.encode(), ctypes.c_int64(int(value))), 'er_config_set')

  def device_info(self):
    cu, wave = ctypes.c_int(0), ctypes.c_int(0)
    buf = ctypes.create_string_buffer(64)
    self._ck(self.lib.er_device_info(ctypes.byref(cu), ctypes.byref(wave), buf, 64), 'er_device_info')
    return {'cu_count': cu.value, 'wave_size': wave.value, 'arch': buf.value.decode()}

  # -- K14 sharded embedding checkpoint files (host code)
  def save_dense_embed(self, ckpt_path, var_name, task_index, task_num, vals_np):
    vals_np = np.ascontiguousarray(vals_np, dtype=np.float32)
    rows, dim = vals_np.shape
    self._ck(self.lib.er_save_dense_embed(ckpt_path.encode(), var_name.encode(), ctypes.c_int32(task_index),
                                          ctypes.c_int32(task_num), vals_np.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_int64(rows), ctypes.c_int32(dim)), 'er_save_dense_embed')

  def load_dense_embed(self, ckpt_path, var_name, task_index, task_num, embed_dim, embed_part_size):
    out = np.empty((embed_part_size, embed_dim), dtype=np.float32)
    n = ctypes.c_int64(0)
    self._ck(self.lib.er_load_dense_embed(ckpt_path.encode(), var_name.encode(), ctypes.c_int32(task_index),
                                          ctypes.c_int32(task_num), ctypes.c_int32(embed_dim),
                                          ctypes.c_int64(embed_part_size), out.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.byref(n)), 'er_load_dense_embed')
    return out

  def load_kv_embed(self, ckpt_path, var_name, task_index, task_num, embed_dim):
    n = ctypes.c_int64(0)
    args = (ckpt_path.encode(), var_name.encode(), ctypes.c_int32(task_index), ctypes.c_int32(task_num),
            ctypes.c_int32(embed_dim))
    self._ck(self.lib.er_load_kv_embed(*args, ctypes.c_int64(0), None, None, ctypes.byref(n)), 'er_load_kv_embed')
    keys = np.empty(n.value, dtype=np.int64)
    vals = np.empty((n.value, embed_dim), dtype=np.float32)
    if n.value:
      self._ck(self.lib.er_load_kv_embed(*args, ctypes.c_int64(n.value), keys.ctypes.data_as(ctypes.c_void_p),
                                         vals.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)), 'er_load_kv_embed')
    return keys, vals

  # -- K1 hashing
  def decode_csv_host(self, text, sep, kinds, max_rows, threads=0, out=None):
    """CSVInput's decode step on the host.  text: uint8 array; kinds: 0 string / 1 int / 2 float per field.  threads: 0 =
    one per hardware thread (at most 8), n > 1 = that many (er_decode_csv_host_mt); 1 = the single-pass form
    (er_decode_csv_host).  out: a dict the five output arrays are kept in between calls of the same shape (for a reader
    that is done with a batch's arrays before it decodes the next: fresh 1.3 MB arrays are page-faulted in on every call).
    Returns (n_rows, consumed bytes, ints [F, max_rows], floats, empty mask, str_begin, str_len)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    kinds = np.ascontiguousarray(kinds, dtype=np.int32)
    F = len(kinds)
    # (the threaded decoder writes its column-major outputs at a pitch that is not a power of two: a row's F fields x 5
    # arrays at a 4096-element pitch share their cache sets; the arrays handed back are [F, max_rows] views)
    pitch = max_rows if int(threads) == 1 else max_rows + 24
    key = (F, int(max_rows), int(pitch))
    if out is not None and out.get('key') == key:
      ints, flts, empty, begin, length = out['arrays']
    else:
      ints = np.empty((F, pitch), dtype=np.int64)[:, :max_rows]
      flts = np.empty((F, pitch), dtype=np.float64)[:, :max_rows]
      empty = np.empty((F, pitch), dtype=np.uint8)[:, :max_rows]
      begin = np.empty((F, pitch), dtype=np.int64)[:, :max_rows]
      length = np.empty((F, pitch), dtype=np.int32)[:, :max_rows]
      if out is not None:
        out['key'], out['arrays'] = key, (ints, flts, empty, begin, length)
    n_rows, consumed = ctypes.c_int64(0), ctypes.c_int64(0)
    sep_b = sep.encode('utf-8') if isinstance(sep, str) else bytes(sep)
    assert len(sep_b) == 1, 'separator must be one byte: %r' % sep

    if text.size == 0:
      return 0, 0, ints, flts, empty, begin, length

    def ptr(a):
      return a.ctypes.data_as(ctypes.c_void_p)
    if int(threads) == 1:
      self._ck(self.lib.er_decode_csv_host(ptr(text), ctypes.c_int64(text.size), ctypes.c_uint8(sep_b[0]), ctypes.c_int32(F),
                                           ptr(kinds), ctypes.c_int64(int(max_rows)), ptr(ints), ptr(flts), ptr(empty),
                                           ptr(begin), ptr(length), ctypes.byref(n_rows), ctypes.byref(consumed)),
               'er_decode_csv_host')
    else:
      self._ck(self.lib.er_decode_csv_host_mt(ptr(text), ctypes.c_int64(text.size), ctypes.c_uint8(sep_b[0]), ctypes.c_int32(F),
                                              ptr(kinds), ctypes.c_int64(int(max_rows)), ctypes.c_int64(int(pitch)), ptr(ints),
                                              ptr(flts), ptr(empty), ptr(begin), ptr(length), ctypes.byref(n_rows),
                                              ctypes.byref(consumed), ctypes.c_int32(int(threads))), 'er_decode_csv_host_mt')
    return n_rows.value, consumed.value, ints, flts, empty, begin, length

  def pack_cells_host(self, text, begin, length):
    """Cells (begin, length) of `text` -> (packed uint8 bytes, int64 offsets[n + 1])."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    begin = np.ascontiguousarray(begin, dtype=np.int64)
    length = np.ascontiguousarray(length, dtype=np.int32)
    n = len(begin)
    out = np.empty(max(int(length.sum(dtype=np.int64)), 1), dtype=np.uint8)
    offsets = np.empty(n + 1, dtype=np.int64)

    def ptr(a):
      return a.ctypes.data_as(ctypes.c_void_p)
    self._ck(self.lib.er_pack_cells_host(ptr(text) if text.size else ptr(out), ptr(begin), ptr(length), ctypes.c_int64(n),
                                         ptr(out), ptr(offsets)), 'er_pack_cells_host')
    return out[:int(offsets[-1])], offsets

  def split_cells_host(self, text, begin, length, seps, keep_empty=False, max_tokens=0):
    """Cells (begin, length) of `text` split on the bytes of `seps` -> (tok_begin int64 [T], tok_len int32 [T],
    row_offsets int64 [n + 1]) (er_split_cells_host)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    begin = np.ascontiguousarray(begin, dtype=np.int64)
    length = np.ascontiguousarray(length, dtype=np.int32)
    n = len(begin)
    cap = int(length.sum(dtype=np.int64)) + n + 1
    tb, tl = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int32)
    offs = np.empty(n + 1, dtype=np.int64)
    sb = np.frombuffer(seps if isinstance(seps, bytes) else seps.encode('utf-8'), dtype=np.uint8)

    def ptr(a):
      return a.ctypes.data_as(ctypes.c_void_p)
    self._ck(self.lib.er_split_cells_host(ptr(text) if text.size else ptr(tb), ptr(begin), ptr(length), ctypes.c_int64(n), ptr(sb),
                                          ctypes.c_int32(len(sb)), ctypes.c_int32(int(bool(keep_empty))),
                                          ctypes.c_int32(int(max_tokens)), ptr(tb), ptr(tl), ctypes.c_int64(cap), ptr(offs)),
             'er_split_cells_host')
    T = int(offs[-1])
    return tb[:T], tl[:T], offs

  def pack_int_decimal_host(self, values):
    """int64 array -> (packed uint8 bytes, int64 offsets[n + 1]) of the values' decimal strings (str(int))."""
    values = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
    n = values.size
    out = np.empty(max(20 * n, 1), dtype=np.uint8)
    offsets = np.empty(n + 1, dtype=np.int64)

    def ptr(a):
      return a.ctypes.data_as(ctypes.c_void_p)
    self._ck(self.lib.er_pack_int_decimal_host(ptr(values), ctypes.c_int64(n), ptr(out), ptr(offsets)), 'er_pack_int_decimal_host')
    return out[:int(offsets[-1])], offsets

  def sparse_cross_hashed_host(self, bytes_np, offsets_np, n_rows, n_cols, num_buckets, hash_key=None):
    """ComboFeature through crossed_column: column-major strings -> int64 [n_rows] bucket ids ('' is a value too)."""
    bytes_np = np.ascontiguousarray(bytes_np, dtype=np.uint8)
    offsets_np = np.ascontiguousarray(offsets_np, dtype=np.int64)
    assert len(offsets_np) == n_rows * n_cols + 1
    out = np.empty(n_rows, dtype=np.int64)
    if bytes_np.size == 0:
      bytes_np = np.zeros(1, dtype=np.uint8)
    key = CROSS_HASH_KEY if hash_key is None else int(hash_key)
    self._ck(
        self.lib.er_sparse_cross_hashed_host(
            bytes_np.ctypes.data_as(ctypes.c_void_p), offsets_np.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int64(int(n_rows)), ctypes.c_int32(int(n_cols)), ctypes.c_uint64(int(num_buckets)),
            ctypes.c_uint64(key), out.ctypes.data_as(ctypes.c_void_p)), 'er_sparse_cross_hashed_host')
    return out

  def hash_bucket_fast_host(self, bytes_np, offsets_np, n_per_col, num_buckets, drop_empty):
    """numpy in / numpy out; runs on the host (data-loader threads)."""
    bytes_np = np.ascontiguousarray(bytes_np, dtype=np.uint8)
    offsets_np = np.ascontiguousarray(offsets_np, dtype=np.int64)
    nb = np.ascontiguousarray(num_buckets, dtype=np.uint64)
    n = len(offsets_np) - 1
    out = np.empty(n, dtype=np.int64)
    if bytes_np.size == 0:
      bytes_np = np.zeros(1, dtype=np.uint8)
    self._ck(
        self.lib.er_hash_bucket_fast_host(
            bytes_np.ctypes.data_as(ctypes.c_void_p), offsets_np.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int64(n), ctypes.c_int64(int(n_per_col)), nb.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int(int(drop_empty)), out.ctypes.data_as(ctypes.c_void_p)), 'er_hash_bucket_fast_host')
    return out

  def hash_bucket_fast(self, bytes_t, offsets_t, n_per_col, num_buckets_t, drop_empty, out=None):
    n = offsets_t.numel() - 1
    if out is None:
      out = torch.empty(n, dtype=torch.int64, device=bytes_t.device)
    self._ck(
        self.lib.er_hash_bucket_fast(_p(bytes_t), _p(offsets_t), ctypes.c_int64(n),
                                     ctypes.c_int64(int(n_per_col)), _p(num_buckets_t),
                                     ctypes.c_int(int(drop_empty)), _p(out), _stream()), 'er_hash_bucket_fast')
    return out

  def hash_bucket_fast_int64(self, values_t, n_per_col, num_buckets_t, out=None):
    n = values_t.numel()
    if out is None:
      out = torch.empty(n, dtype=torch.int64, device=values_t.device)
    self._ck(
        self.lib.er_hash_bucket_fast_int64(_p(values_t), ctypes.c_int64(n), ctypes.c_int64(int(n_per_col)),
                                           _p(num_buckets_t), _p(out), _stream()), 'er_hash_bucket_fast_int64')
    return out

  # -- K2 / K3 / K4 embeddings
  @staticmethod
  def _descs(specs):
    arr = (LookupDesc * len(specs))()
    for i, s in enumerate(specs):
      assert s.table.dtype == torch.float32 and s.ids.dtype == torch.int64
      assert s.out.dim() == 2 and s.out.stride(1) == 1
      d = arr[i]
      d.table = s.table.data_ptr()
      d.ids = s.ids.data_ptr()
      d.offsets = s.offsets.data_ptr() if s.offsets is not None else None
      d.weights = s.weights.data_ptr() if s.weights is not None else None
      d.out = s.out.data_ptr()
      d.rows, d.key_base, d.dim = s.rows, s.key_base, s.dim
      d.out_stride, d.out_col, d.combiner = s.out.stride(0), s.out_col, s.combiner
      d.n_rows, d.max_nnz = s.n_rows, s.max_nnz
      # a column block of a wider buffer as the table (er_emb_fwd only: rows received side by side with another group's)
      assert s.table.dim() != 2 or s.table.stride(1) == 1
      d.table_ld = s.table.stride(0) if (s.table.dim() == 2 and s.table.stride(0) != s.dim) else 0
    return arr

  def emb_plan_create(self, specs):
    plan = ctypes.c_void_p(0)
    self._ck(self.lib.er_emb_plan_create(self._descs(specs), len(specs), ctypes.byref(plan)),
             'er_emb_plan_create')
    nblk = self.lib.er_emb_plan_num_blocks(plan)
    return {'handle': plan, 'specs': list(specs), 'num_blocks': nblk}

  def emb_plan_destroy(self, plan):
    self.lib.er_emb_plan_destroy(plan['handle'])

  def emb_fwd(self, plan, sumsq_partials=None):
    self._ck(self.lib.er_emb_fwd(plan['handle'], _p(sumsq_partials), _stream()), 'er_emb_fwd')

  def emb_group_create(self, specs, dim, total_rows, var, m, v, bitmap):
    grp = ctypes.c_void_p(0)
    self._ck(
        self.lib.er_emb_group_create(self._descs(specs), len(specs), ctypes.c_int32(dim),
                                     ctypes.c_int64(total_rows), _p(var), _p(m), _p(v), _p(bitmap),
                                     ctypes.byref(grp)), 'er_emb_group_create')
    n_ent = self.lib.er_emb_group_num_entries(grp)
    group = {'handle': grp, 'specs': list(specs), 'dim': dim, 'total_rows': total_rows, 'var': var,
             'm': m, 'v': v, 'bitmap': bitmap, 'num_entries': n_ent, 'last_step': None}
    self._set_row_pitch(group)
    return group

  def _set_row_pitch(self, group):
    """var / m / v (and last_step) may be column blocks of ONE [rows, ld] record buffer (input_layer._alloc_storage):
    the pitches are read off the tensors' strides."""
    var, ls = group['var'], group.get('last_step')
    ld = var.stride(0) if var.dim() == 2 else group['dim']
    for t in (group['m'], group['v']):
      assert t is None or (t.dim() == 2 and t.stride(0) == ld and t.stride(1) == 1), 'var / m / v must share one row pitch'
    assert var.dim() != 2 or var.stride(1) == 1
    ls_ld = 1 if ls is None else (ls.stride(0) if ls.numel() > 1 else 1)
    if ld != group['dim'] or ls_ld != 1:
      self._ck(self.lib.er_emb_group_set_row_pitch(group['handle'], ctypes.c_int64(ld), ctypes.c_int64(ls_ld)),
               'er_emb_group_set_row_pitch')

  def emb_group_destroy(self, group):
    self.lib.er_emb_group_destroy(group['handle'])

  def emb_bwd_update(self, group, opt_kind, hyper):
    self._ck(self.lib.er_emb_bwd_update(group['handle'], ctypes.c_int(opt_kind), _p(hyper), _stream()),
             'er_emb_bwd_update')

  def emb_bwd_reduce(self, group, out=None):
    """out: a (keys, grads, n_unique) triple from an earlier call to reuse (fixed addresses for hipGraphs)."""
    n = group['num_entries']
    dev = group['var'].device
    if out is None:
      keys = torch.empty(n, dtype=torch.int32, device=dev)
      grads = torch.empty(n, group['dim'], dtype=torch.float32, device=dev)
      n_unique = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
      keys, grads, n_unique = out
    self._ck(self.lib.er_emb_bwd_reduce(group['handle'], _p(keys), _p(grads), _p(n_unique), _stream()),
             'er_emb_bwd_reduce')
    return keys, grads, n_unique

  # -- K13 GEMM on the matrix cores
  def gemm_reserve(self, floats):
    self._ck(self.lib.er_gemm_reserve(ctypes.c_int64(int(floats))), 'er_gemm_reserve')

  def gemm_row_tiles(self, M):
    return int(self.lib.er_gemm_row_tiles(ctypes.c_int32(int(M))))

  def bn_apply_from_stats(self, x, bias, col_stats, chunks, gamma, beta, eps, momentum, moving_mean, moving_var, act,
                          bf16_state=None):
    """BatchNorm(train) + activation from ready-made column statistics (er_gemm's epilogue).  bf16_state (a Bf16Shadows):
    the launch also writes y's bf16 copy, registered there for the contraction that reads y next."""
    B, N = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device)
    invstd = torch.empty(N, dtype=torch.float32, device=x.device)
    yb = bf16_state.new_copy(y) if bf16_state is not None else None
    self._ck(
        self.lib.er_bn_apply_from_stats_b16(_p(_f32c(x)), _p(bias), _p(col_stats), ctypes.c_int32(int(chunks)), _p(gamma),
                                            _p(beta), B, N, ctypes.c_float(eps), ctypes.c_float(momentum),
                                            _p(moving_mean), _p(moving_var), int(act), _p(y), _p(mean), _p(invstd),
                                            _p(yb), ctypes.c_int32(0 if yb is None else yb.stride(0)),
                                            _stream()), 'er_bn_apply_from_stats')
    return y, mean, invstd

  def gemm(self, layout, a, b, out=None, bias=None, accumulate=False, bf16=False, col_stats=None):
    """out (+)= op(a) . op(b) (+ bias).  2-D fp32 tensors with unit inner stride.
    layout GEMM_NN: a[M,K] b[K,N]; GEMM_NT: a[M,K] b[N,K]; GEMM_TN: a[K,M] b[K,N]."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    assert a.dtype == torch.float32 and b.dtype == torch.float32
    if layout == GEMM_NN:
      (M, K), (K2, N) = a.shape, b.shape
    elif layout == GEMM_NT:
      (M, K), (N, K2) = a.shape, b.shape
    else:
      (K, M), (K2, N) = a.shape, b.shape
    assert K == K2, 'gemm: inner dimensions %d vs %d' % (K, K2)
    if out is None:
      assert not accumulate
      out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    if bf16 and self._gemm_bf16_fast(layout, a, b, out, bias, accumulate, M, N, K, col_stats=col_stats):
      return out
    if layout == GEMM_NN and not bf16 and not accumulate and col_stats is None and self.gemv_ok(a, M, N, K):
      return self._gemv(a, None, b, bias, out)
    self._log_gemm('gemm_bf16_kernel' if bf16 else 'gemm_f32_kernel', layout, M, N, K)
    fn = self.lib.er_gemm_bf16 if bf16 else self.lib.er_gemm_f32
    if col_stats is not None:
      assert col_stats.numel() >= self.gemm_row_tiles(M) * N * 3 and col_stats.dtype == torch.float32
    self._ck(fn(ctypes.c_int(layout), M, N, K, _p(a), ctypes.c_int32(a.stride(0)), _p(b), ctypes.c_int32(b.stride(0)),
                _p(out), ctypes.c_int32(out.stride(0)), _p(bias), int(bool(accumulate)), _p(col_stats), _stream()),
             'er_gemm')
    return out

  # -- bf16 operands in HBM (dense_dtype 'bf16'): er_gemm_bf16_nt
  bf16_nt = True  # (False: er_gemm_bf16, which converts fp32 operands while staging - round 1's form, kept for tests)

  # the producers of a bf16 step write their consumers' bf16 operands and the bf16 contractions carry the BatchNorm /
  # cross epilogues (er_gemm_bf16_nt_epi); A/B switch: '0' = round 5's arrangement (cast launches, fp32-only epilogues)
  bf16_epilogues = os.environ.get('EASYREC_AMD_BF16_EPILOGUES', '1') != '0'

  def bf16_enable(self, varstore):
    """Called by the estimator (dense_dtype 'bf16') once the dense variables are packed; outside graph capture."""
    self._ck(self.lib.er_gemm_bf16_nt_prepare(), 'er_gemm_bf16_nt_prepare')
    states = [s for s in getattr(self, '_bf16_states', []) if s.alive()]
    st = Bf16Shadows(self, varstore)
    states.append(st)
    self._bf16_states = states
    return st

  def _bf16_state_of(self, t):
    for st in getattr(self, '_bf16_states', ()):
      if st.owns(t) and st.alive():
        return st
    return None

  def cast_bf16(self, items):
    """items: [(src fp32 [rows, cols] (unit inner stride), dst bf16, transpose)]: dst[r, c] = src[r, c] or
    dst[c, r] = src[r, c], padding columns of dst zeroed; ceil(n / 16) launches."""
    n = len(items)
    arr = (CastDesc * n)()
    for i, (src, dst, tr) in enumerate(items):
      assert src.dim() == 2 and src.stride(1) == 1 and src.dtype == torch.float32
      assert dst.dim() == 2 and dst.stride(1) == 1 and dst.dtype == torch.bfloat16
      rows, cols = src.shape
      assert dst.shape[0] == (cols if tr else rows) and dst.stride(0) >= (rows if tr else cols)
      arr[i] = CastDesc(src.data_ptr(), dst.data_ptr(), rows, cols, src.stride(0), dst.stride(0), 1 if tr else 0)
    self._ck(self.lib.er_cast_bf16(arr, n, _stream()), 'er_cast_bf16')

  def gemm_bf16_nt(self, a, bt, M, N, K, out=None, out_bf16=None, bias=None, accumulate=False, epi=None):
    """out[M, N] (+)= a[M, K] . bt[N, K]^T (+ bias): bf16 operands with leading dimensions that are multiples of 8,
    fp32 accumulate; out fp32 and / or out_bf16.  epi: a GemmEpilogue record (er_gemm_bf16_nt_epi)."""
    assert a.dtype == torch.bfloat16 and bt.dtype == torch.bfloat16 and a.stride(1) == 1 and bt.stride(1) == 1
    assert a.shape[0] >= M and bt.shape[0] >= N
    Kp = (K + 7) // 8 * 8  # (the k-tail up to the padded width reads zeros: Bf16Shadows pads with zeros)
    assert a.stride(0) >= Kp and bt.stride(0) >= Kp and a.stride(0) % 8 == 0 and bt.stride(0) % 8 == 0
    kind = 0 if epi is None else int(epi.kind)
    self._log_gemm('gemm_bf16_nt_kernel<4, %d>' % kind, None, M, N, K)
    self._ck(self.lib.er_gemm_bf16_nt_epi(M, N, Kp, ctypes.c_void_p(a.data_ptr()), ctypes.c_int32(a.stride(0)),
                                          ctypes.c_void_p(bt.data_ptr()), ctypes.c_int32(bt.stride(0)), _p(out),
                                          ctypes.c_int32(0 if out is None else out.stride(0)),
                                          ctypes.c_void_p(0 if out_bf16 is None else out_bf16.data_ptr()),
                                          ctypes.c_int32(0 if out_bf16 is None else out_bf16.stride(0)), _p(bias),
                                          int(bool(accumulate)), None if epi is None else ctypes.byref(epi), _stream()),
             'er_gemm_bf16_nt')
    return out

  @staticmethod
  def _nt_rows_ok(*ts):
    """The epilogues of er_gemm_bf16_nt_epi read / write fp32 rows in 16-byte pieces."""
    return all(t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1) for t in ts)

  def _gemm_bf16_fast(self, layout, a, b, out, bias, accumulate, M, N, K, col_stats=None):
    """The bf16 contraction through er_gemm_bf16_nt when one operand is a weight of a bf16-enabled store (forward
    x . W, input gradient dy . W^T); False = not applicable.  col_stats: the following BatchNorm's per-row-tile column
    statistics from the epilogue (ER_EPI_STATS)."""
    if not self.bf16_nt or layout == GEMM_TN:
      return False
    st = self._bf16_state_of(b)
    if st is None:
      return False
    epi = None
    if col_stats is not None:
      if N % 4 != 0 or accumulate or not self._nt_rows_ok(out):
        return False
      epi = GemmEpilogue(kind=EPI_STATS, col_stats=col_stats.data_ptr())
    w, plain, t = st.weight(b)
    a16 = st.act(a, cache=(layout == GEMM_NN))
    self.gemm_bf16_nt(a16, t if layout == GEMM_NN else plain, M, N, K, out=out, bias=bias, accumulate=accumulate, epi=epi)
    return True

  # er_gemm_f32_bn_bwd / er_bn_act_bwd_from_partials are used
  fused_bn_bwd = True
  # the BatchNorm-backward column sums of a layer whose output is a column block of the consumer's input (DeepFM's deep
  # tower inside [sum(wide) | FM | deep]) from the consumer's dgrad epilogue (er_gemm_f32_bn_bwd_cols); A/B switch
  bn_cols_epilogue = os.environ.get('EASYREC_AMD_BN_COLS_EPILOGUE', '1') != '0'

  def gemm_bn_bwd(self, layout, a, b, src, partial, col0=None, bf16=False):
    """dgrad GEMM whose epilogue also emits the BatchNorm-backward column sums of the layer described by `src`
    (a BnSource: the producer of this GEMM's input) into partial [row tiles][N][2].  col0: that layer produced the
    columns [col0, col0 + its width) of this GEMM's input only.  bf16: through er_gemm_bf16_nt_epi (ER_EPI_BN_BWD) when
    b is a weight of a bf16-enabled store; None = not applicable there."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if layout == GEMM_NN:
      (M, K), (K2, N) = a.shape, b.shape
    elif layout == GEMM_NT:
      (M, K), (N, K2) = a.shape, b.shape
    else:
      (K, M), (K2, N) = a.shape, b.shape
    if bf16:
      st = self._bf16_state_of(b) if (self.bf16_nt and layout == GEMM_NT) else None
      n_src = src.y.shape[1]
      c0 = 0 if col0 is None else int(col0)
      if st is None or N % 4 or n_src % 4 or c0 % 4 or not self._nt_rows_ok(src.z, src.y) or src.y.stride() != src.z.stride():
        return None
      assert K == K2 and src.z.shape == (M, n_src) and c0 + n_src <= N and partial.numel() >= self.gemm_row_tiles(M) * n_src * 2
      out = torch.empty(M, N, dtype=torch.float32, device=a.device)
      epi = GemmEpilogue(kind=EPI_BN_BWD, bn_z=src.z.data_ptr(), bn_zbias=_ptr(src.zbias), bn_y=src.y.data_ptr(),
                         bn_mean=_ptr(src.mean), bn_invstd=_ptr(src.invstd), bn_partial=partial.data_ptr(),
                         bn_ld=src.y.stride(0), bn_use_bn=int(src.mean is not None), bn_act=int(src.act), bn_col0=c0,
                         bn_n_src=n_src)
      w, plain, t = st.weight(b)
      self.gemm_bf16_nt(st.act(a), plain, M, N, K, out=out, epi=epi)
      return out
    if col0 is not None:
      n_src = src.y.shape[1]
      assert K == K2 and src.z.shape == (M, n_src) and src.y.stride() == src.z.stride() and 0 <= col0 and col0 + n_src <= N
      assert partial.numel() >= self.gemm_row_tiles(M) * n_src * 2
      self._log_gemm('gemm_f32_bn_bwd_kernel', layout, M, N, K)
      out = torch.empty(M, N, dtype=torch.float32, device=a.device)
      self._ck(self.lib.er_gemm_f32_bn_bwd_cols(ctypes.c_int(layout), M, N, K, _p(a), ctypes.c_int32(a.stride(0)), _p(b),
                                                ctypes.c_int32(b.stride(0)), _p(out), ctypes.c_int32(out.stride(0)),
                                                _p(src.z), _p(src.zbias), _p(src.y), _p(src.mean), _p(src.invstd),
                                                ctypes.c_int32(src.y.stride(0)), int(src.mean is not None), int(src.act),
                                                _p(partial), ctypes.c_int32(int(col0)), ctypes.c_int32(n_src), _stream()),
               'er_gemm_f32_bn_bwd_cols')
      return out
    assert K == K2 and src.z.shape == (M, N) and src.y.shape == (M, N) and src.y.stride() == src.z.stride()
    assert partial.numel() >= self.gemm_row_tiles(M) * N * 2
    self._log_gemm('gemm_f32_bn_bwd_kernel', layout, M, N, K)
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    use_bn = src.mean is not None
    self._ck(self.lib.er_gemm_f32_bn_bwd(ctypes.c_int(layout), M, N, K, _p(a), ctypes.c_int32(a.stride(0)), _p(b),
                                         ctypes.c_int32(b.stride(0)), _p(out), ctypes.c_int32(out.stride(0)),
                                         _p(src.z), _p(src.zbias), _p(src.y), _p(src.mean), _p(src.invstd),
                                         ctypes.c_int32(src.y.stride(0)), int(use_bn), int(src.act), _p(partial),
                                         _stream()), 'er_gemm_f32_bn_bwd')
    return out

  # the same-depth layers of parallel stacks (MMoE's experts, task towers) as one grouped launch: layers/dnn.py run_parallel
  grouped_stacks = True

  # the bias / BatchNorm / activation launches of those layers as one launch forward, two backward (er_bn_fwd_multi /
  # er_bn_bwd_multi)
  grouped_bn = True
  BN_MULTI_MAX_ROWS = 8192  # (taller layers reduce their partial sums in a merge launch of their own: single launches)

  def bn_fwd_multi(self, layers):
    """layers: [dict(x, bias, gamma, beta, moving_mean, moving_var, col_stats, use_bn, act, eps, momentum)] -> [(y, mean,
    invstd)]: the forward of BNFromStatsFn (use_bn 1: batch statistics from the GEMM's epilogue) / BNActFn (0, BN_FROZEN)
    for all of them in ONE launch."""
    arr = (BnLayer * len(layers))()
    outs = []
    for q, l in zip(arr, layers):
      x = l['x']
      assert x.dim() == 2 and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] <= self.BN_MULTI_MAX_ROWS
      B, N = x.shape
      y = torch.empty_like(x)
      mean = torch.empty(N, dtype=torch.float32, device=x.device) if l['use_bn'] else None
      invstd = torch.empty(N, dtype=torch.float32, device=x.device) if l['use_bn'] else None
      q.x, q.bias, q.gamma, q.beta = x.data_ptr(), _ptr(l.get('bias')), _ptr(l.get('gamma')), _ptr(l.get('beta'))
      q.moving_mean, q.moving_var = _ptr(l.get('moving_mean')), _ptr(l.get('moving_var'))
      q.B, q.N, q.use_bn, q.act = B, N, int(l['use_bn']), int(l['act'])
      q.eps, q.momentum = float(l.get('eps', 0.0)), float(l.get('momentum', 0.0))
      if int(l['use_bn']) == BN_BATCH:
        q.col_stats, q.chunks = l['col_stats'].data_ptr(), self.gemm_row_tiles(B)
      q.y, q.save_mean, q.save_invstd = y.data_ptr(), _ptr(mean), _ptr(invstd)
      outs.append((y, mean, invstd))
    self._ck(self.lib.er_bn_fwd_multi(arr, len(layers), _stream()), 'er_bn_fwd_multi')
    return outs

  def bn_bwd_multi(self, layers):
    """layers: [dict(x, bias, gamma, beta, y, mean, invstd, dy, use_bn, act, partial, into)] -> [(dx, dbias, dgamma, dbeta)]
    (bn_act_bwd's arguments and results, layer by layer) in two launches: column sums (of the layers without `partial`),
    finalize + apply."""
    arr = (BnLayer * len(layers))()
    outs = []
    for q, l in zip(arr, layers):
      x, dy = l['x'], l['dy']
      B, N = x.shape
      assert dy.dim() == 2 and dy.stride(1) == 1 and dy.dtype == torch.float32 and B <= self.BN_MULTI_MAX_ROWS
      dev = x.device
      dx_done = bool(l.get('dx_done'))  # (dy already holds dx: the contraction that produced it ran the elementwise backward)
      dx = dy if dx_done else torch.empty_like(x)
      into = l.get('into')
      if into is not None:
        dbias, dgamma, dbeta = into
      else:
        dbias = torch.empty(N, dtype=torch.float32, device=dev) if l.get('bias') is not None else None
        dgamma = torch.empty(N, dtype=torch.float32, device=dev) if l.get('gamma') is not None else None
        dbeta = torch.empty(N, dtype=torch.float32, device=dev) if l.get('gamma') is not None else None
      partial = l.get('partial')
      if partial is not None and dy.stride(0) != N:
        partial = None
      q.x, q.bias, q.gamma, q.beta = x.data_ptr(), _ptr(l.get('bias')), _ptr(l.get('gamma')), _ptr(l.get('beta'))
      q.B, q.N, q.use_bn, q.act = B, N, int(l['use_bn']), int(l['act'])
      q.save_mean, q.save_invstd = _ptr(l.get('mean')), _ptr(l.get('invstd'))
      q.y_in = _ptr(l['y'])
      q.dy, q.dy_ld = dy.data_ptr(), dy.stride(0)
      if partial is not None:
        q.partial, q.chunks = partial.data_ptr(), self.gemm_row_tiles(B)
      assert not dx_done or (partial is not None and int(l['use_bn']) == BN_FROZEN)
      q.dx, q.dbias, q.dgamma, q.dbeta = (None if dx_done else dx.data_ptr()), _ptr(dbias), _ptr(dgamma), _ptr(dbeta)
      q.accumulate = int(into is not None)
      outs.append((dx, None, None, None) if into is not None else (dx, dbias, dgamma, dbeta))
    self._ck(self.lib.er_bn_bwd_multi(arr, len(layers), _stream()), 'er_bn_bwd_multi')
    return outs

  def gemm_grouped(self, layout, problems, bf16=False):
    """problems: [(a, b, out, bias, accumulate)] fp32, one layout: ONE launch (+ one for the split-K reduces).
    bf16: operands rounded to bf16 while staged (er_gemm_grouped_bf16)."""
    arr = self._gemm_problems(layout, problems, bf16)
    if bf16:
      self._ck(self.lib.er_gemm_grouped_bf16(ctypes.c_int(layout), arr, len(problems), _stream()), 'er_gemm_grouped_bf16')
    else:
      self._ck(self.lib.er_gemm_grouped_f32(ctypes.c_int(layout), arr, len(problems), _stream()), 'er_gemm_grouped_f32')

  def _gemm_problems(self, layout, problems, bf16=False, log_as=None):
    """The er_gemm_problem array of a grouped launch (log_as: the kernel the op log books the contractions under)."""
    arr = (GemmProblem * len(problems))()
    for q, pr in zip(arr, problems):
      a, b, out, bias, accumulate = pr[:5]
      stats = pr[5] if len(pr) > 5 else None
      bn = pr[6] if len(pr) > 6 else None  # (BnSource of the layer that produced this problem's output position, partial)
      assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
      assert a.dtype == torch.float32 and b.dtype == torch.float32 and out.dtype == torch.float32
      if layout == GEMM_NN:
        (M, K), (K2, N) = a.shape, b.shape
      elif layout == GEMM_NT:
        (M, K), (N, K2) = a.shape, b.shape
      else:
        (K, M), (K2, N) = a.shape, b.shape
      assert K == K2 and out.shape == (M, N)
      if self.op_log is not None:
        if log_as is not None:
          self._log_gemm(log_as, None, M, N, K)
        elif bf16:
          self._log_gemm('gemm_bf16_grouped_kernel', layout, M, N, K)
        else:
          self._log_gemm('gemm_f32_grouped_kernel', layout, M, N, K)
      q.M, q.N, q.K = M, N, K
      q.A, q.lda, q.B, q.ldb = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
      q.C, q.ldc = out.data_ptr(), out.stride(0)
      q.bias = bias.data_ptr() if bias is not None else None
      q.accumulate = int(bool(accumulate))
      if stats is not None:
        assert stats.dtype == torch.float32 and stats.numel() >= self.gemm_row_tiles(M) * N * 3 and not accumulate
        q.col_stats = stats.data_ptr()
      if bn is not None:
        src, partial = bn[:2]
        if len(bn) > 2 and bn[2]:  # (the layer's elementwise backward in this launch's epilogue: frozen statistics only)
          assert src.frozen and layout == GEMM_NT
          q.bn_gamma, q.bn_dz_out = _ptr(src.gamma), 1
        assert src.z.shape == (M, N) and partial.numel() >= self.gemm_row_tiles(M) * N * 2 and not accumulate
        q.bn_z, q.bn_zbias = _ptr(src.z), _ptr(src.zbias)
        q.bn_y = _ptr(src.y)
        q.bn_mean, q.bn_invstd = _ptr(src.mean), _ptr(src.invstd)
        q.bn_ld, q.bn_use_bn, q.bn_act = src.z.stride(0), int(src.mean is not None), int(src.act)
        q.bn_partial = partial.data_ptr()
      fz = pr[7] if len(pr) > 7 else None  # (the frozen-BatchNorm forward epilogue: er_gemm_problem.fz_*)
      if fz is not None:
        assert layout == GEMM_NN and not bf16 and stats is None and bn is None and not accumulate
        y, save = fz['y'], fz['save']
        assert y.shape == (M, N) and y.stride(0) == out.stride(0) and y.stride(1) == 1 and save.is_contiguous() and \
            save.numel() == 2 * N
        q.fz_bias, q.fz_gamma, q.fz_beta = _ptr(fz.get('bias')), _ptr(fz.get('gamma')), _ptr(fz.get('beta'))
        q.fz_mean, q.fz_var = fz['moving_mean'].data_ptr(), fz['moving_var'].data_ptr()
        q.fz_eps, q.fz_act = float(fz['eps']), int(fz['act'])
        q.fz_y, q.fz_save = y.data_ptr(), save.data_ptr()
    return arr

  # weight gradients of a backward pass: queued by LinearFn / LinearBNActFn, contracted together by flush_wgrads()
  def wgrad_sink(self):
    """The calling thread's queue object.  Functions fetch it in forward() (the caller's thread: the simulated ranks
    of the embedding-parallel tests are threads) because backward() runs on an autograd engine thread."""
    sink = getattr(_wgrad_tls, 'sink', None)
    if sink is None:
      sink = _wgrad_tls.sink = WgradSink()
    return sink

  def defer_wgrads(self):
    sink = self.wgrad_sink()
    assert not sink.queue and not sink.queue_bf16, 'flush_wgrads() was not called for the previous backward pass'
    sink.active = True

  def flush_wgrads(self):
    """Launch the queued weight gradients on the current stream.  Returns the queue: when that stream is not the one
    the operands were produced on, the caller keeps it alive until the streams have joined."""
    q, qb = self.take_wgrads()
    if q:
      self.gemm_grouped(GEMM_TN, q)
    if qb:
      self.gemm_grouped(GEMM_TN, qb, bf16=True)
    return q + qb

  def take_wgrads(self):
    """The queued weight gradients (fp32, bf16), NOT launched: the caller contracts them - the fused tail puts the fp32
    ones into the embedding backward's grid (emb_bwd_fused(wgrads=...))."""
    sink = self.wgrad_sink()
    q, sink.queue, sink.active = sink.queue, [], False
    qb, sink.queue_bf16 = sink.queue_bf16, []
    jobs, sink.colsum_jobs = sink.colsum_jobs, []
    if jobs:
      self.colsum_partials_multi(jobs)
    return q, qb

  def queue_colsum(self, sink, partial, dst, n_cols):
    """dst[j] += sum_p partial[p, j] once the backward pass is over (one launch for all queued jobs), or now."""
    if sink is not None and sink.active:
      sink.colsum_jobs.append((partial, dst, n_cols))
    else:
      self.colsum_partials_multi([(partial, dst, n_cols)])

  @staticmethod
  def colsum_is_narrow(x):
    """er_colsum_acc would take its one-workgroup-per-column form for x"""
    return x.dim() == 2 and x.shape[1] <= 8 and x.shape[0] * x.shape[1] <= (1 << 17) and x.stride(1) == 1

  def colsum_narrow_multi(self, jobs, accumulate=True):
    """jobs: [(x [rows, cols <= 8], out [cols])]: out[j] (+)= sum_i x[i, j], all jobs in ONE launch (column by column the
    sums of colsum())."""
    arr = (ColsumJob * len(jobs))()
    for q, (x, out) in zip(arr, jobs):
      assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and out.is_contiguous() and out.numel() == x.shape[1]
      q.x, q.rows, q.cols, q.x_stride, q.out = x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr()
    self._ck(self.lib.er_colsum_narrow_multi(arr, ctypes.c_int32(len(jobs)), int(bool(accumulate)), _stream()),
             'er_colsum_narrow_multi')

  def colsum_partials_multi(self, jobs, accumulate=True):
    """jobs: [(partial [P, ld], dst [n_cols], n_cols)]: dst[j] (+)= sum_p partial[p, j], all jobs in one launch."""
    arr = (TailJob * len(jobs))()
    for q, (partial, d, n_cols) in zip(arr, jobs):
      assert partial.dim() == 2 and partial.stride(1) == 1 and d.is_contiguous() and d.numel() == n_cols
      q.partial, q.dst, q.n_parts, q.n_cols, q.ld = partial.data_ptr(), d.data_ptr(), partial.shape[0], int(n_cols), partial.stride(0)
    self._ck(self.lib.er_colsum_partials_multi(arr, ctypes.c_int32(len(jobs)), int(bool(accumulate)), _stream()),
             'er_colsum_partials_multi')

  # the step's tail in one grid (er_emb_bwd_fused_wgrad); A/B switch, and the workgroups its contraction's k-splits aim at
  # (0: the stand-alone launch's 512 - bit-identical to the unfused tail; same-box 256 -> 54.5 us, 512 -> 60, 1024 -> 66)
  fused_tail = os.environ.get('EASYREC_AMD_FUSED_TAIL', '1') != '0'
  tail_wgrad_blocks = int(os.environ.get('EASYREC_AMD_TAIL_BLOCKS', '256'))

  @staticmethod
  def wgrads_fit_the_tail(q):
    """plain fp32 contractions only, at most one grouped launch's worth"""
    return 0 < len(q) <= 16 and all(len(pr) <= 5 for pr in q)

  # -- K12 embedding-parallel routing (include/easyrec_hip.h)
  def emb_group_set_routing(self, group, world, shard_stride, local_base):
    arr = (ctypes.c_int64 * len(local_base))(*[int(x) for x in local_base])
    self._ck(self.lib.er_emb_group_set_routing(group['handle'], ctypes.c_int32(world), ctypes.c_int64(shard_stride), arr),
             'er_emb_group_set_routing')
    group['world'], group['shard_stride'], group['local_base'] = world, shard_stride, list(local_base)

  def emb_group_set_active(self, group, n_rows):
    self._ck(self.lib.er_emb_group_set_active(group['handle'], ctypes.c_int64(int(n_rows))), 'er_emb_group_set_active')
    group['n_active'] = int(n_rows)

  def emb_group_share_sort(self, group, leader):
    """True when `group` now reuses `leader`'s per-step sort (identical ids and table geometry), else False."""
    if not same_lookup_keys(group, leader):
      return False
    self._ck(self.lib.er_emb_group_share_sort(group['handle'], leader['handle']), 'er_emb_group_share_sort')
    group['sort_leader'] = leader
    return True

  def emb_route(self, group, unique_keys, n_unique, entry_unique_index, owner_counts):
    if unique_keys is None:  # follower of a shared sort: the outputs are the leader's
      assert group.get('sort_leader') is not None and n_unique is None and entry_unique_index is None
      self._ck(self.lib.er_emb_route(group['handle'], None, None, None, None, _stream()), 'er_emb_route')
      return
    assert unique_keys.dtype == torch.int32 and n_unique.dtype == torch.int32
    assert entry_unique_index is None or entry_unique_index.dtype == torch.int64
    assert owner_counts is None or owner_counts.dtype == torch.int32
    self._ck(self.lib.er_emb_route(group['handle'], _p(unique_keys), _p(n_unique), _p(entry_unique_index),
                                   _p(owner_counts), _stream()), 'er_emb_route')

  def emb_bwd_reduce_routed(self, group, unique_grads):
    assert unique_grads.dtype == torch.float32 and unique_grads.dim() == 2 and unique_grads.stride(1) == 1
    self._ck(self.lib.er_emb_bwd_reduce_routed(group['handle'], _p(unique_grads), ctypes.c_int32(unique_grads.stride(0)),
                                               _stream()), 'er_emb_bwd_reduce_routed')

  # the embedding-parallel requester's local reductions + the step's weight gradients + the loss tail as two launches
  # (er_emb_reduce_local_tail); A/B switch: '0' = one launch per reduction kind and dim group, the grouped wgrad launch apart
  ep_merged_reduce = os.environ.get('EASYREC_AMD_EP_MERGED_REDUCE', '1') != '0'

  def emb_reduce_local_tail(self, routed, dense, wgrads=None):
    """routed: [(sharded requester group, unique_grads [slots, >= dim])], dense: [(replicated group, dense [rows, dim + 1])]
    (at most 4 together): their local gradient reductions in ONE tile launch + ONE fix launch; wgrads: take_wgrads()'s fp32
    list (wgrads_fit_the_tail) contracted in the same grid, with a deferred loss tail (loss_tail(defer=True)) as one more
    workgroup of it."""
    items = [(g, 1, t) for g, t in routed] + [(g, 2, t) for g, t in dense]
    n = len(items)
    assert 1 <= n <= 4
    for _, _, t in items:
      assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g, _, _ in items])
    modes = (ctypes.c_int32 * n)(*[m for _, m, _ in items])
    outs = (ctypes.c_void_p * n)(*[t.data_ptr() for _, _, t in items])
    ld = (ctypes.c_int32 * n)(*[t.stride(0) for _, _, t in items])
    pr, lt = None, None
    if wgrads:
      pr = self._gemm_problems(GEMM_TN, wgrads, log_as='emb_reduce_local_wgrad_kernel')
      lt = self._deferred_loss_tail
      self._deferred_loss_tail = None
    self._ck(self.lib.er_emb_reduce_local_tail(gh, modes, outs, ld, n, pr, len(wgrads) if wgrads else 0,
                                               ctypes.c_int32(int(self.tail_wgrad_blocks) if wgrads else 0),
                                               ctypes.byref(lt[0]) if lt is not None else None, _stream()),
             'er_emb_reduce_local_tail')

  def gather_rows(self, table, keys, n, key_sub, out):
    assert keys.dtype == torch.int32 and table.dim() == 2 and table.stride(1) == 1
    self._ck(self.lib.er_gather_rows_ld(_p(table), ctypes.c_int64(table.stride(0)), ctypes.c_int64(table.shape[0]),
                                        ctypes.c_int32(table.shape[1]), _p(keys), ctypes.c_int64(int(n)),
                                        ctypes.c_int64(int(key_sub)), _p(out), _stream()), 'er_gather_rows_ld')

  def scatter_unique(self, keys, grads, n_unique, capacity, dim, dense):
    assert dense.dim() == 2 and dense.stride(1) == 1
    self._ck(self.lib.er_scatter_unique(_p(keys), _p(grads), _p(n_unique), ctypes.c_int64(int(capacity)),
                                        ctypes.c_int32(dim), _p(dense), ctypes.c_int32(dense.stride(0)), _stream()),
             'er_scatter_unique')

  def emb_owner_merge(self, group, run_counts):
    """Owner side: the received keys are len(run_counts) ascending duplicate-free runs; merge instead of sort."""
    n = len(run_counts)
    rc = (ctypes.c_int32 * n)(*[int(x) for x in run_counts])
    self._ck(self.lib.er_emb_owner_merge(group['handle'], rc, n, _stream()), 'er_emb_owner_merge')

  def emb_group_set_peer_capacity(self, group, peer_cap, count_header=True):
    self._ck(self.lib.er_emb_group_set_peer_capacity(group['handle'], ctypes.c_int64(int(peer_cap)),
                                                     ctypes.c_int32(1 if count_header else 0)),
             'er_emb_group_set_peer_capacity')
    group['peer_cap'], group['peer_hdr'] = int(peer_cap), bool(count_header)

  def emb_route_overflow(self, group):
    out = ctypes.c_int32(0)
    self._ck(self.lib.er_emb_route_overflow(group['handle'], ctypes.byref(out)), 'er_emb_route_overflow')
    return bool(out.value)

  def emb_owner_ids(self, recv_keys, counts, n_runs, peer_cap, key_sub, ids, counts_out):
    """counts None: the received runs carry their count in front ([count, keys ...], peer_cap + 1 slots each)."""
    hdr = 1 if counts is None else 0
    assert recv_keys.dtype == torch.int32 and ids.dtype == torch.int64 and counts_out.dtype == torch.int32
    assert recv_keys.numel() >= n_runs * (peer_cap + hdr) and ids.numel() >= n_runs * peer_cap and counts_out.numel() >= n_runs
    assert counts is None or (counts.dtype == torch.int32 and counts.numel() >= n_runs)
    self._ck(self.lib.er_emb_owner_ids(_p(recv_keys), None if counts is None else _p(counts), ctypes.c_int(n_runs),
                                       ctypes.c_int64(int(peer_cap)), ctypes.c_int64(int(key_sub)), _p(ids), _p(counts_out),
                                       _stream()), 'er_emb_owner_ids')

  # owner ids + entry build + merge (+ the serve launch's lag-1 replay table) as one launch - A/B switch
  ep_owner_fused = os.environ.get('EASYREC_AMD_EP_OWNER_FUSED', '1') != '0'

  def emb_owner_ids_merge(self, group, recv_keys, n_runs, peer_cap, key_sub, ids, counts_out, build_tables):
    """er_emb_owner_ids_merge: emb_owner_ids (runs with their count in front) + emb_owner_merge_padded in one launch."""
    assert recv_keys.dtype == torch.int32 and ids.dtype == torch.int64 and counts_out.dtype == torch.int32
    assert recv_keys.numel() >= n_runs * (peer_cap + 1) and ids.numel() >= n_runs * peer_cap and counts_out.numel() >= n_runs
    self._ck(self.lib.er_emb_owner_ids_merge(group['handle'], _p(recv_keys), None, ctypes.c_int(n_runs),
                                             ctypes.c_int64(int(peer_cap)), ctypes.c_int64(int(key_sub)), _p(ids),
                                             _p(counts_out), ctypes.c_int(1 if build_tables else 0), _stream()),
             'er_emb_owner_ids_merge')

  def emb_owner_merge_padded(self, group, counts, n_runs, peer_cap):
    assert counts.dtype == torch.int32 and counts.numel() >= n_runs
    self._ck(self.lib.er_emb_owner_merge_padded(group['handle'], _p(counts), ctypes.c_int(n_runs),
                                                ctypes.c_int64(int(peer_cap)), _stream()), 'er_emb_owner_merge_padded')

  def emb_owner_serve(self, groups, rows_out, hyper):
    """Catch up (lazy dense decay) and reply the received rows of up to 4 owner groups in one launch."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    for g, o in zip(groups, rows_out):  # (a column block of a wider buffer is fine: ld = its row stride)
      assert o.stride(1) == 1 and o.dtype == torch.float32 and o.shape[1] == g['dim']
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in rows_out])
    ld = (ctypes.c_int32 * n)(*[o.stride(0) for o in rows_out])
    self._ck(self.lib.er_emb_owner_serve(gh, op, ld, n, None if hyper is None else _p(hyper), _stream()),
             'er_emb_owner_serve')

  def emb_bwd_reduce_dense(self, groups, dense):
    """Per-row gradient sums of the groups straight into their dense [rows, dim + 1] buffers (count in the last
    column); one launch for all groups."""
    n = len(groups)
    for d in dense:
      assert d.dim() == 2 and d.stride(1) == 1 and d.dtype == torch.float32
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dense])
    ld = (ctypes.c_int32 * n)(*[d.stride(0) for d in dense])
    self._ck(self.lib.er_emb_bwd_reduce_dense(gh, dp, ld, n, _stream()), 'er_emb_bwd_reduce_dense')

  def emb_dense_apply(self, tables, opt_kind, hyper):
    """tables: [(var, m, v, dense)] - one pass over each replicated table after the all-reduce of `dense`."""
    n = len(tables)
    descs = (DenseApplyDesc * n)()
    for i, (var, m, v, dense) in enumerate(tables):
      assert var.stride(1) == 1 and dense.stride(1) == 1 and dense.shape[0] == var.shape[0]
      assert all(t is None or t.stride() == var.stride() for t in (m, v))
      descs[i] = DenseApplyDesc(var.data_ptr(), None if m is None else m.data_ptr(), None if v is None else v.data_ptr(),
                                dense.data_ptr(), dense.stride(0), var.shape[1], var.shape[0], var.stride(0))
    self._ck(self.lib.er_emb_dense_apply(descs, n, ctypes.c_int(opt_kind), _p(hyper), _stream()), 'er_emb_dense_apply')

  def emb_mark_touched(self, group):
    self._ck(self.lib.er_emb_mark_touched(group['handle'], _stream()), 'er_emb_mark_touched')

  def emb_sweep_untouched(self, group, hyper):
    self._ck(self.lib.er_emb_sweep_untouched(group['handle'], _p(hyper), _stream()), 'er_emb_sweep_untouched')

  def stream_copy(self, src, dst):
    nbytes = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() >= nbytes
    self._ck(self.lib.er_stream_copy(_p(src), _p(dst), ctypes.c_int64(nbytes), _stream()), 'er_stream_copy')

  def adam_decay_sweep(self, var, m, v, bitmap, total_rows, dim, hyper):
    ld = var.stride(0) if var.dim() == 2 else dim
    assert var.dim() != 2 or (m.stride() == var.stride() and v.stride() == var.stride() and var.stride(1) == 1)
    self._ck(
        self.lib.er_adam_decay_sweep_ld(_p(var), _p(m), _p(v), _p(bitmap), ctypes.c_int64(total_rows),
                                        ctypes.c_int32(dim), ctypes.c_int64(ld), _p(hyper), _stream()),
        'er_adam_decay_sweep_ld')

  # -- K5 FM / wide
  def fm_fwd(self, x, F, D):
    """x: [B, >=F*D] row-major (stride(0) = row stride).  Returns (fm [B,D], S [B,D])."""
    B = x.shape[0]
    fm = torch.empty(B, D, dtype=torch.float32, device=x.device)
    S = torch.empty(B, D, dtype=torch.float32, device=x.device)
    self._ck(self.lib.er_fm_fwd(_p(x), B, F, D, x.stride(0), _p(fm), _p(S), _stream()), 'er_fm_fwd')
    return fm, S

  def fm_bwd(self, x, S, g, F, D, into=None, accumulate=False):
    B = x.shape[0]
    dx = torch.empty(B, F * D, dtype=torch.float32, device=x.device) if into is None else into
    assert dx.shape == (B, F * D) and dx.stride(1) == 1
    self._ck(
        self.lib.er_fm_bwd(_p(x), _p(S), _p(_f32c(g)), B, F, D, x.stride(0), _p(dx), dx.stride(0), int(accumulate),
                           _stream()), 'er_fm_bwd')
    return dx

  def auc_update(self, probs, labels, weights, thresholds, counts):
    """counts int64 [2, T + 1] += histogram of (label != 0, #thresholds below the prediction); see core/metrics.py."""
    probs, labels = _f32c(probs.reshape(-1)), _f32c(labels.reshape(-1))
    assert probs.numel() == labels.numel() and counts.dtype == torch.int64 and counts.is_contiguous()
    assert counts.numel() == 2 * (thresholds.numel() + 1)
    w = None if weights is None else _f32c(weights.reshape(-1))
    self._ck(self.lib.er_auc_update(_p(probs), _p(labels), _p(w), ctypes.c_int64(probs.numel()), _p(thresholds),
                                    ctypes.c_int32(thresholds.numel()), _p(counts), _stream()), 'er_auc_update')

  def grouped_auc(self, keys, preds, labels, reduction):
    """gAUC / session AUC of the accumulated rows (int64 keys, fp32 predictions, labels != 0 positive) -> (sum of
    w * AUC over the keys with both classes, sum of w, number of such keys); the ordering by (key, prediction) is two
    stable device sorts, the ranks and the reduction are er_grouped_auc (one host read at the end)."""
    n = keys.numel()
    preds, labels = _f32c(preds.reshape(-1)), _f32c(labels.reshape(-1))
    assert preds.numel() == n and labels.numel() == n and keys.dtype == torch.int64
    out = torch.zeros(3, dtype=torch.float64, device=keys.device)
    if n:
      by_pred = torch.sort(preds, stable=True).indices
      order = by_pred[torch.sort(keys.reshape(-1)[by_pred], stable=True).indices]
      k, p, y = keys.reshape(-1)[order].contiguous(), preds[order].contiguous(), labels[order].contiguous()
      work = torch.zeros(3 * n, dtype=torch.float64, device=keys.device)
      self._ck(self.lib.er_grouped_auc(_p(k), _p(p), _p(y), ctypes.c_int64(n), ctypes.c_int32(int(reduction)), _p(work), _p(out),
                                       _stream()), 'er_grouped_auc')
    return tuple(float(x) for x in out.cpu().tolist())

  def dot_interaction_fwd(self, x, F, D, self_interaction):
    """x [B, F*D] -> [B, P] pairwise dot products in the order of model/dlrm.py:51-57."""
    B = x.shape[0]
    P = F * (F - 1) // 2 + (F if self_interaction else 0)
    out = torch.empty(B, P, dtype=torch.float32, device=x.device)
    self._ck(self.lib.er_dot_interaction_fwd(_p(x), B, F, D, x.stride(0), int(bool(self_interaction)), _p(out),
                                             out.stride(0), _stream()), 'er_dot_interaction_fwd')
    return out

  def dot_interaction_bwd(self, x, g, F, D, self_interaction):
    B = x.shape[0]
    dx = torch.empty(B, F * D, dtype=torch.float32, device=x.device)
    g = _f32c(g)
    self._ck(self.lib.er_dot_interaction_bwd(_p(x), _p(g), B, F, D, x.stride(0), int(bool(self_interaction)),
                                             g.stride(0), _p(dx), dx.stride(0), 0, _stream()),
             'er_dot_interaction_bwd')
    return dx

  def rowsum_fwd(self, x, n):
    B = x.shape[0]
    out = torch.empty(B, 1, dtype=torch.float32, device=x.device)
    self._ck(self.lib.er_rowsum_fwd(_p(x), B, n, x.stride(0), _p(out), _stream()), 'er_rowsum_fwd')
    return out

  fused_wide_fm = os.environ.get('EASYREC_AMD_FUSED_WIDE_FM', '1') != '0'  # A/B switch

  def wide_fm_concat(self, wide, fm_x, F, D, deep):
    """[sum(wide) | FM(fm_x) | deep] at a 16-byte row pitch and the FM field sums ([B, D]) in one launch."""
    B, n_w, n_d = wide.shape[0], wide.shape[1], deep.shape[1]
    assert wide.stride(1) == 1 and fm_x.stride(1) == 1 and deep.stride(1) == 1 and fm_x.shape[1] >= F * D
    width = 1 + D + n_d
    out = torch.empty(B, (width + 3) // 4 * 4, dtype=torch.float32, device=wide.device)[:, :width]
    S = torch.empty(B, D, dtype=torch.float32, device=wide.device)
    self._ck(self.lib.er_wide_fm_concat(_p(wide), n_w, wide.stride(0), _p(fm_x), F, D, fm_x.stride(0), _p(deep), n_d,
                                        deep.stride(0), B, _p(out), out.stride(0), _p(S), _stream()), 'er_wide_fm_concat')
    return out, S

  def bn_apply_wide_fm(self, pend, wide, fm_x, F, D):
    """er_bn_apply_wide_fm: the deferred BatchNorm finalize + apply `pend` (LinearBNActFn(defer_apply=True)) and
    [sum(wide) | FM(fm_x) | y] in ONE launch -> (out, S) as wide_fm_concat, or None when the library wants the two launches."""
    z, y = pend['z'], pend['y']
    B, N = z.shape
    n_w = wide.shape[1]
    assert wide.stride(1) == 1 and fm_x.stride(1) == 1 and fm_x.shape[1] >= F * D and z.is_contiguous() and y.is_contiguous()
    width = 1 + D + N
    out = torch.empty(B, (width + 3) // 4 * 4, dtype=torch.float32, device=wide.device)[:, :width]
    S = torch.empty(B, D, dtype=torch.float32, device=wide.device)
    rc = self.lib.er_bn_apply_wide_fm(_p(z), _p(pend['stats']), ctypes.c_int32(int(pend['chunks'])), _p(pend['gamma']),
                                      _p(pend['beta']), B, N, ctypes.c_float(pend['eps']), ctypes.c_float(pend['momentum']),
                                      _p(pend['moving_mean']), _p(pend['moving_var']), int(pend['act']), _p(y),
                                      _p(pend['mean']), _p(pend['invstd']), _p(wide), n_w, wide.stride(0), _p(fm_x), F, D,
                                      fm_x.stride(0), _p(out), out.stride(0), _p(S), _stream())
    if rc == 3:
      return None
    self._ck(rc, 'er_bn_apply_wide_fm')
    return out, S

  def bn_apply_pending(self, pend):
    """the deferred BatchNorm finalize + apply as the launch of its own it would have been"""
    z = pend['z']
    B, N = z.shape
    self._ck(self.lib.er_bn_apply_from_stats_b16(_p(z), None, _p(pend['stats']), ctypes.c_int32(int(pend['chunks'])),
                                                 _p(pend['gamma']), _p(pend['beta']), B, N, ctypes.c_float(pend['eps']),
                                                 ctypes.c_float(pend['momentum']), _p(pend['moving_mean']),
                                                 _p(pend['moving_var']), int(pend['act']), _p(pend['y']), _p(pend['mean']),
                                                 _p(pend['invstd']), None, ctypes.c_int32(0), _stream()),
             'er_bn_apply_from_stats')

  # the last BatchNorm apply of DeepFM's deep tower inside the [sum(wide) | FM | deep] launch (er_bn_apply_wide_fm) - A/B switch
  defer_bn_apply = os.environ.get('EASYREC_AMD_DEFER_BN_APPLY', '1') != '0'

  # a TALL layer's BatchNorm apply inside the NEXT contraction's staging (er_bn_finalize_from_stats + er_gemm_f32_bn_a): the
  # [B * L, units] activations of DIN's attention MLP are not read back by an apply launch of their own - A/B switch
  bn_in_staging = os.environ.get('EASYREC_AMD_BN_IN_STAGING', '1') != '0'
  BN_IN_STAGING_MIN_ROWS = 256 * 64 + 1  # (layers whose column statistics take the merge launch: more than 256 row tiles)

  def bn_a_ok(self, pend, w):
    """May the deferred BatchNorm apply `pend` run inside the staging of the contraction with w [K, N]?"""
    z, y = pend['z'], pend['y']
    K = z.shape[1]
    return (K % 4 == 0 and K <= 256 and z.is_contiguous() and y.is_contiguous() and w.dim() == 2 and w.stride(1) == 1 and
            (z.data_ptr() | y.data_ptr()) % 16 == 0)

  # the bias + frozen BatchNorm + activation of a multi-task model's expert layers inside the grouped contraction's epilogue
  # (er_gemm_problem.fz_*: one launch writes z and y; the depth's BatchNorm launch disappears) - A/B switch
  frozen_bn_epilogue = os.environ.get('EASYREC_AMD_FROZEN_BN_EPILOGUE', '1') != '0'

  # ... and their elementwise backward (dz = gamma * invstd * masked dy) in the epilogue of the input-gradient contraction of
  # the layer ABOVE (er_gemm_problem.bn_dz_out): the experts' BatchNorm-backward passes of all but the last depth disappear
  frozen_dz_epilogue = os.environ.get('EASYREC_AMD_FROZEN_DZ_EPILOGUE', '1') != '0'

  # tall projections onto <= 4 columns (DIN's attention scores [B x L, 32] -> [B x L, 1]) by er_gemv_f32_bn_a - A/B switch
  tall_gemv = os.environ.get('EASYREC_AMD_TALL_GEMV', '1') != '0'

  def gemv_ok(self, x, M, N, K):
    return (self.tall_gemv and N <= 4 and M >= self.BN_IN_STAGING_MIN_ROWS and K % 4 == 0 and x.stride(1) == 1 and
            x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)

  def _gemv(self, x, pend, w, bias, out=None):
    """out [M, N <= 4] = x . w (+ bias), or act(BatchNorm(z)) . w for a deferred apply `pend` (x = its z; statistics already
    finalized)"""
    M, K = x.shape
    N = w.shape[1]
    if out is None:
      out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    if self.op_log is not None:
      self.op_log.append(('er::gemv_bna_kernel<%d>' % N, 2.0 * M * N * K))
    bn = pend is not None
    y = pend['y'] if bn else None
    self._ck(self.lib.er_gemv_f32_bn_a(M, N, K, _p(x), ctypes.c_int32(x.stride(0)), _p(pend['mean']) if bn else None,
                                       _p(pend['invstd']) if bn else None, _p(pend['gamma']) if bn else None,
                                       _p(pend['beta']) if bn else None, int(pend['act']) if bn else 0, _p(y),
                                       ctypes.c_int32(y.stride(0) if bn else 0), _p(w), ctypes.c_int32(w.stride(0)), _p(out),
                                       ctypes.c_int32(out.stride(0)), _p(bias), _stream()), 'er_gemv_f32_bn_a')
    return out

  def wgrad_tall_narrow_ok(self, x, dz):
    return (self.tall_gemv and x.dim() == 2 and dz.dim() == 2 and dz.shape[1] <= 4 and x.shape[0] >= self.BN_IN_STAGING_MIN_ROWS and
            x.shape[1] % 4 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and dz.stride(1) == 1 and
            x.dtype == torch.float32 and dz.dtype == torch.float32)

  def wgrad_tall_narrow(self, x, dz, out, accumulate=True, bias_grad=None):
    """out [K, N <= 4] (+)= x^T . dz for a tall x [rows, K] (er_wgrad_tall_narrow: the weight gradient of DIN's attention score
    projection, which stalled the grouped weight-gradient launch on its [rows, 1] operand)"""
    rows, K = x.shape
    N = dz.shape[1]
    assert out.shape == (K, N) and out.stride(1) == 1 and dz.shape[0] == rows
    assert bias_grad is None or (bias_grad.is_contiguous() and bias_grad.numel() == N)
    scratch = torch.empty(513 * (K + 1) * N, dtype=torch.float32, device=x.device)
    if self.op_log is not None:
      self.op_log.append(('er::wgrad_narrow_partial_kernel<%d>' % N, 2.0 * rows * N * K))
    self._ck(self.lib.er_wgrad_tall_narrow(rows, K, N, _p(x), ctypes.c_int32(x.stride(0)), _p(dz), ctypes.c_int32(dz.stride(0)),
                                           _p(out), ctypes.c_int32(out.stride(0)), _p(bias_grad), int(bool(accumulate)), _p(scratch),
                                           ctypes.c_int64(scratch.numel()), _stream()), 'er_wgrad_tall_narrow')
    return out

  def gemm_bn_a(self, pend, w, bias, col_stats=None):
    """out [M, N] = act(BatchNorm(z)) . w (+ bias) for the deferred BatchNorm apply `pend` (LinearBNActFn(defer_apply=True)):
    the statistics are finalized by a launch of their own, the apply runs while the contraction stages its A tiles and leaves
    pend['y'] behind for the backward pass.  The same bits as bn_apply_pending + gemm."""
    z, y = pend['z'], pend['y']
    M, K = z.shape
    N = w.shape[1]
    self._ck(self.lib.er_bn_finalize_from_stats(_p(pend['stats']), ctypes.c_int32(int(pend['chunks'])), M, K,
                                                ctypes.c_float(pend['eps']), ctypes.c_float(pend['momentum']),
                                                _p(pend['moving_mean']), _p(pend['moving_var']), _p(pend['mean']),
                                                _p(pend['invstd']), _stream()), 'er_bn_finalize_from_stats')
    if col_stats is None and self.gemv_ok(z, M, N, K):
      return self._gemv(z, pend, w, bias)
    out = torch.empty(M, N, dtype=torch.float32, device=z.device)
    if col_stats is not None:
      assert col_stats.numel() >= self.gemm_row_tiles(M) * N * 3 and col_stats.dtype == torch.float32
    self._log_gemm('gemm_f32_bna_kernel', None, M, N, K)
    self._ck(self.lib.er_gemm_f32_bn_a(M, N, K, _p(z), ctypes.c_int32(z.stride(0)), _p(pend['mean']), _p(pend['invstd']),
                                       _p(pend['gamma']), _p(pend['beta']), int(pend['act']), _p(y),
                                       ctypes.c_int32(y.stride(0)), _p(w), ctypes.c_int32(w.stride(0)), _p(out),
                                       ctypes.c_int32(out.stride(0)), _p(bias), _p(col_stats), _stream()), 'er_gemm_f32_bn_a')
    return out

  def rowsum_bwd(self, g, n, into=None, accumulate=False):
    B = g.shape[0]
    dx = torch.empty(B, n, dtype=torch.float32, device=g.device) if into is None else into
    assert dx.shape == (B, n) and dx.stride(1) == 1
    self._ck(self.lib.er_rowsum_bwd(_p(_f32c(g)), B, n, _p(dx), dx.stride(0), int(accumulate), _stream()),
             'er_rowsum_bwd')
    return dx

  def group_grad_finish(self, groups):
    """groups: [(dout, out, lambda, has_base, [terms])], term = ('rowsum', g, col0, width) | ('fm', g, saved, col0,
    width, dim); g: [B] / [B, 1] / [B, dim] with unit inner stride.  One launch (er_group_grad_finish)."""
    arr, _keep = self._grad_groups(groups)
    self._ck(self.lib.er_group_grad_finish(arr, len(groups), _stream()), 'er_group_grad_finish')

  # the fused single-GPU embedding step: er_emb_front (build + sort + heads [+ decay table] in one launch, catch-up from
  # the heads) and er_emb_bwd_fused (finish + reduce + row update in one launch) - A/B switch
  fused_emb = os.environ.get('EASYREC_AMD_FUSED_EMB', '1') != '0'

  # the catch-up of lazy dense decay in registers (er_emb_fwd_lazy / er_emb_bwd_fused) instead of a launch of its own - A/B switch
  defer_catch_up = os.environ.get('EASYREC_AMD_DEFER_CATCH_UP', '1') != '0'

  def emb_front(self, groups, hyper, skip_one_row, defer=False):
    """-> False when the groups need the general path (nothing launched).  defer: no catch-up launch - the step's lookup
    must then be emb_fwd_lazy."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    flags = (1 if skip_one_row else 0) | (2 if defer else 0)
    rc = self.lib.er_emb_front(gh, n, ctypes.c_int(flags), _p(hyper), _stream())
    if rc == 3:
      return False
    self._ck(rc, 'er_emb_front')
    return True

  def emb_fwd_lazy(self, plan, groups, hyper, sumsq_partials=None):
    """emb_fwd whose lookups into the lazily decaying `groups` catch their rows up in registers (after emb_front(defer))."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    self._ck(self.lib.er_emb_fwd_lazy(plan['handle'], gh, n, _p(hyper), _p(sumsq_partials), _stream()), 'er_emb_fwd_lazy')

  def emb_front_fwd(self, groups, plan, hyper, skip_one_row, sumsq_partials=None):
    """emb_front(defer=True) + emb_fwd_lazy in one call (one launch when the prologue builds the replay table).
    -> False when the groups need the general path (nothing launched)."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    rc = self.lib.er_emb_front_fwd(gh, n, ctypes.c_int((1 if skip_one_row else 0) | 2), plan['handle'], _p(hyper),
                                   _p(sumsq_partials), _stream())
    if rc == 3:
      return False
    self._ck(rc, 'er_emb_front_fwd')
    return True

  # the step's lag-1 replay table built by the prologue launch, so that sort and lookup share a launch - A/B switch
  prologue_tables = os.environ.get('EASYREC_AMD_PROLOGUE_TABLES', '1') != '0'

  def decay_tables_set_prologue_build(self, tabs, on):
    self._ck(self.lib.er_decay_tables_set_prologue_build(tabs['handle'], int(bool(on))), 'er_decay_tables_set_prologue_build')
    tabs['prologue_build'] = bool(on)

  def decay_tables_sync(self, tabs):
    self._ck(self.lib.er_decay_tables_sync(tabs['handle'], _stream()), 'er_decay_tables_sync')

  def decay_tables_error(self, tabs):
    flag = ctypes.c_int32(0)
    self._ck(self.lib.er_decay_tables_error(tabs['handle'], ctypes.byref(flag)), 'er_decay_tables_error')
    return bool(flag.value)

  def emb_bwd_fused(self, groups, finish, opt_kind, hyper, wgrads=None, dense_opt=None):
    """finish: group_grad_finish's descriptors, one per feature-group gradient buffer the groups' lookups write.
    wgrads: the step's queued weight gradients (take_wgrads()'s fp32 list, wgrads_fit_the_tail) - contracted in the same
    grid (er_emb_bwd_fused_tail), a deferred loss tail (loss_tail(defer=True)) as one more workgroup of it; dense_opt:
    dense_opt_step's arguments (w, m, v, grad, l2coef, opt_kind, hyper, l2_partials) - the dense optimizer behind the
    cross-tile fix, finishing the k-split weight gradients while it reads them (dense_opt_fits_the_tail).  Returns True
    when the optimizer ran here."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    arr, _keep = self._grad_groups(finish)
    if wgrads:
      pr = self._gemm_problems(GEMM_TN, wgrads, log_as='emb_bwd_own_wgrad_kernel')
      self.tail_launches = getattr(self, 'tail_launches', 0) + 1  # (tests: the fused tail is what ran)
      lt = self._deferred_loss_tail
      self._deferred_loss_tail = None
      oj = None
      if dense_opt is not None:
        w, m, v, grad, l2coef, kind, hyp, l2p = dense_opt
        oj = DenseOptJob(_ptr(w), _ptr(m), _ptr(v), _ptr(grad), _ptr(l2coef), w.numel(), int(kind), _ptr(hyp), _ptr(l2p))
      self._ck(self.lib.er_emb_bwd_fused_tail(gh, n, arr, len(finish), ctypes.c_int(opt_kind), _p(hyper), pr, len(wgrads),
                                              ctypes.c_int32(int(self.tail_wgrad_blocks)),
                                              ctypes.byref(lt[0]) if lt is not None else None,
                                              ctypes.byref(oj) if oj is not None else None, _stream()),
               'er_emb_bwd_fused_tail')
      if oj is not None:
        st = self._bf16_state_of(dense_opt[0])
        if st is not None:
          st.refresh()  # (a bf16 step whose dense optimizer ran in the tail: the weights' bf16 shadows follow the masters)
      return oj is not None
    self._ck(self.lib.er_emb_bwd_fused(gh, n, arr, len(finish), ctypes.c_int(opt_kind), _p(hyper), _stream()),
             'er_emb_bwd_fused')
    return False

  # the riders of the fused tail (er_emb_bwd_fused_tail); A/B switch
  tail_riders = os.environ.get('EASYREC_AMD_TAIL_RIDERS', '1') != '0'

  def dense_opt_fits_the_tail(self, wgrads, w, grad):
    """every queued weight gradient is a contiguous block of the flat gradient buffer (the optimizer finishes the k-split
    ones while it reads them), fp32 masters without bf16 shadows to refresh"""
    if self._bf16_state_of(w) is not None:
      return False
    lo, hi = grad.data_ptr(), grad.data_ptr() + 4 * grad.numel()
    return all(pr[2].is_contiguous() and lo <= pr[2].data_ptr() and pr[2].data_ptr() + 4 * pr[2].numel() <= hi for pr in wgrads)

  def _grad_groups(self, groups):
    arr = (GradGroup * len(groups))()
    keep = []
    for q, (dout, out, lam, has_base, terms) in zip(arr, groups):
      assert dout.shape == out.shape and dout.stride() == out.stride() and dout.stride(1) == 1 and len(terms) <= 4
      q.dout, q.out, q.ld = dout.data_ptr(), out.data_ptr(), dout.stride(0)
      q.batch, q.width, q.has_base, q.n_terms, q.lam = dout.shape[0], dout.shape[1], int(bool(has_base)), len(terms), lam
      for t, term in zip(q.terms, terms):
        g = term[1]
        assert g.dtype == torch.float32 and (g.dim() == 1 or g.stride(-1) == 1)
        t.g, t.g_ld = g.data_ptr(), (g.stride(0) if g.dim() >= 1 else 1)
        if term[0] == 'rowsum':
          t.kind, t.col0, t.width, t.dim = GRAD_TERM_ROWSUM, term[2], term[3], 1
        else:
          saved = _f32c(term[2])
          keep.append(saved)
          t.kind, t.col0, t.width, t.dim, t.saved = GRAD_TERM_FM, term[3], term[4], term[5], saved.data_ptr()
    return arr, keep

  def axpy2d(self, x, alpha, y, accumulate=True):
    """y (+)= alpha * x on 2-D views with unit inner stride."""
    rows, cols = x.shape
    assert x.stride(1) == 1 and y.stride(1) == 1 and y.shape == x.shape
    self._ck(
        self.lib.er_axpy2d(_p(x), x.stride(0), ctypes.c_float(alpha), _p(y), y.stride(0), rows, cols,
                           int(accumulate), _stream()), 'er_axpy2d')

  # -- K6 / K7 cross
  def cross_v1_fwd(self, x0, w, b):
    B, d = x0.shape
    L = w.shape[0]
    out = torch.empty_like(x0)
    dots = torch.empty(B, L, dtype=torch.float32, device=x0.device)
    self._ck(self.lib.er_cross_v1_fwd(_p(_f32c(x0)), _p(_f32c(w)), _p(_f32c(b)), B, d, L, _p(out), _p(dots),
                                      _stream()), 'er_cross_v1_fwd')
    return out, dots

  def cross_v1_bwd(self, x0, w, b, dots, dout):
    B, d = x0.shape
    L = w.shape[0]
    npart = self.lib.er_cross_v1_bwd_partials(B)
    dx0 = torch.empty_like(x0)
    dwp = torch.empty(npart, L * d, dtype=torch.float32, device=x0.device)
    dbp = torch.empty(npart, L * d, dtype=torch.float32, device=x0.device)
    self._ck(
        self.lib.er_cross_v1_bwd(_p(x0), _p(w), _p(b), _p(dots), _p(_f32c(dout)), B, d, L, _p(dx0), _p(dwp),
                                 _p(dbp), _stream()), 'er_cross_v1_bwd')
    dw = self.colsum(dwp).view(L, d)
    db = self.colsum(dbp).view(L, d)
    return dx0, dw, db

  def cross_v2_fwd(self, x0, x, u, bias, diag_scale):
    B, d = x0.shape
    out = torch.empty_like(x0)
    self._ck(
        self.lib.er_cross_v2_epilogue_fwd(_p(_f32c(x0)), _p(_f32c(x)), _p(_f32c(u)), _p(bias),
                                          ctypes.c_float(diag_scale), B, d, _p(out), _stream()),
        'er_cross_v2_epilogue_fwd')
    return out

  def cross_v2_bwd(self, x0, x, u, bias, diag_scale, dout):
    B, d = x0.shape
    dx0, dx, du = torch.empty_like(x0), torch.empty_like(x0), torch.empty_like(x0)
    self._ck(
        self.lib.er_cross_v2_epilogue_bwd(_p(x0), _p(x), _p(u), _p(bias), ctypes.c_float(diag_scale),
                                          _p(_f32c(dout)), B, d, _p(dx0), 0, _p(dx), _p(du), _stream()),
        'er_cross_v2_epilogue_bwd')
    return dx0, dx, du

  # The DCN-v2 cross layer as ONE launch forward and one or two backward (north_star: "DCN-cross ... as fused HIP kernels";
  # reference layers/keras/interaction.py:249-286): the elementwise part rides in the epilogue of the layer's contraction
  # (er_gemm_f32_cross / er_gemm_bf16_nt_epi).  A/B switch: '0' = GEMM + er_cross_v2_epilogue_* launches.
  fused_cross = os.environ.get('EASYREC_AMD_FUSED_CROSS', '1') != '0'

  def _cross_b16(self, w, bf16, *rows):
    """The Bf16Shadows a bf16 cross contraction over weight w runs on (None: the fp32 kernel), or False: bf16 was asked
    for but er_gemm_bf16_nt_epi does not take these operands (the caller uses the unfused launches)."""
    if not bf16:
      return None
    st = self._bf16_state_of(w) if (self.bf16_nt and self.bf16_epilogues) else None
    if st is None or w.shape[0] % 4 or w.shape[1] % 4 or not self._nt_rows_ok(*rows):
      return False
    return st

  def cross_fwd_fused(self, x0, x, w, bias, diag, bf16):
    """(out, u) = (x0 * (x . w + bias + diag * x) + x, x . w) in one launch; None: not applicable."""
    B, d = x.shape
    if w.shape != (d, d) or x0.shape != x.shape or x0.stride(1) != 1 or x.stride(1) != 1:
      return None
    out = torch.empty(B, d, dtype=torch.float32, device=x.device)
    u = torch.empty(B, d, dtype=torch.float32, device=x.device)
    st = self._cross_b16(w, bf16, x0, x, out, u)
    if st is False:
      return None
    epi = GemmEpilogue(kind=EPI_CROSS_FWD, diag=float(diag), x0=x0.data_ptr(), xl=x.data_ptr(), u=u.data_ptr(),
                       ld_x0=x0.stride(0), ld_xl=x.stride(0), ld_u=u.stride(0))
    if st is not None:
      _, plain, t = st.weight(w)
      self.gemm_bf16_nt(st.act(x, cache=True), t, B, d, d, out=out, out_bf16=st.new_copy(out), bias=bias, epi=epi)
    else:
      self._log_gemm('gemm_f32_cross_kernel<true, false, 3>', None, B, d, d)
      self._ck(self.lib.er_gemm_f32_cross(ctypes.c_int(GEMM_NN), B, d, d, _p(x), ctypes.c_int32(x.stride(0)), _p(w),
                                          ctypes.c_int32(w.stride(0)), _p(out), ctypes.c_int32(out.stride(0)), _p(bias), 0,
                                          ctypes.byref(epi), _stream()), 'er_gemm_f32_cross')
    return out, u

  def cross_bwd_top(self, x0, x, u, bias, diag, dout, dx0, acc0, bf16, w):
    """The elementwise backward of a cross layer whose dout comes from outside the stack: du = dout * x0 (and its bf16 copy
    in a bf16 step), dx0 (+)= dout * (u + bias + diag * x), partial [row tiles, d] = per-tile column sums of du."""
    B, d = x0.shape
    du = torch.empty(B, d, dtype=torch.float32, device=x0.device)
    partial = torch.empty(self.gemm_row_tiles(B), d, dtype=torch.float32, device=x0.device)
    st = self._cross_b16(w, bf16, du) if bf16 else None
    dub = None
    if st:
      dub = torch.empty(B, st.pad8(d), dtype=torch.bfloat16, device=x0.device)  # (the kernel zeroes the k-tail columns)
      st.register(du, dub)
    self._ck(self.lib.er_cross_v2_bwd_top(_p(x0), _p(x), _p(u), _p(bias), ctypes.c_float(diag), _p(dout),
                                          ctypes.c_int32(dout.stride(0)), B, d, _p(dx0), ctypes.c_int32(dx0.stride(0)),
                                          int(bool(acc0)), _p(du), _p(dub), ctypes.c_int32(0 if dub is None else dub.stride(0)),
                                          _p(partial), _stream()), 'er_cross_v2_bwd_top')
    return du, partial

  def cross_dgrad_fused(self, du, w, dout, diag, bf16, dst, acc, prev=None):
    """dst (+)= du . w^T + dout + diag * du - the whole gradient of the layer's input x_{l-1} - in one launch; with prev =
    dict(x0, u, bias, xl, dx0, acc0) (x_{l-1} is the output of the cross layer below) the same launch runs that layer's
    elementwise backward on it: returns (du_prev, partial_prev) (its du = dst * x0 and the per-tile column sums), else
    (None, None).  False: not applicable (bf16 operands the epilogue does not take)."""
    B, d = du.shape
    st = self._cross_b16(w, bf16, du, dout, dst, *( [prev['x0'], prev['u'], prev['dx0']] if prev else []))
    if st is False:
      return False
    epi = GemmEpilogue(kind=EPI_CROSS_BWD, diag=float(diag), dout=dout.data_ptr(), ld_dout=dout.stride(0))
    if diag != 0:
      epi.du_in, epi.ld_du_in = du.data_ptr(), du.stride(0)
    du_prev = partial = None
    if prev is not None:
      du_prev = torch.empty(B, d, dtype=torch.float32, device=du.device)
      partial = torch.empty(self.gemm_row_tiles(B), d, dtype=torch.float32, device=du.device)
      epi.x0, epi.ld_x0 = prev['x0'].data_ptr(), prev['x0'].stride(0)
      epi.prev_u, epi.ld_prev_u = prev['u'].data_ptr(), prev['u'].stride(0)
      epi.prev_bias = _ptr(prev['bias'])
      if diag != 0:
        epi.xl, epi.ld_xl = prev['xl'].data_ptr(), prev['xl'].stride(0)
      epi.dx0, epi.ld_dx0, epi.accumulate_dx0 = prev['dx0'].data_ptr(), prev['dx0'].stride(0), int(bool(prev['acc0']))
      epi.du_out, epi.ld_du_out = du_prev.data_ptr(), du_prev.stride(0)
      epi.partial = partial.data_ptr()
      if st is not None:
        dpb = st.new_copy(du_prev)
        epi.du_out_bf16, epi.ld_du_out_bf16 = dpb.data_ptr(), dpb.stride(0)
    if st is not None:
      _, plain, t = st.weight(w)
      self.gemm_bf16_nt(st.act(du), plain, B, d, d, out=dst, accumulate=acc, epi=epi)
    else:
      self._log_gemm('gemm_f32_cross_kernel<true, true, 4>', None, B, d, d)
      self._ck(self.lib.er_gemm_f32_cross(ctypes.c_int(GEMM_NT), B, d, d, _p(du), ctypes.c_int32(du.stride(0)), _p(w),
                                          ctypes.c_int32(w.stride(0)), _p(dst), ctypes.c_int32(dst.stride(0)), None,
                                          int(bool(acc)), ctypes.byref(epi), _stream()), 'er_gemm_f32_cross')
    return du_prev, partial

  def cross_v2_bwd_acc(self, x0, x, u, bias, diag_scale, dout, dx0, acc0, dx, accx):
    """er_cross_v2_epilogue_bwd_acc: dx0 / dx are 2-D destinations (unit inner stride; dx None: x is x0) written or, with
    their flag, accumulated into.  Returns du."""
    B, d = x0.shape
    du = torch.empty_like(x0)
    assert dx0.stride(1) == 1 and (dx is None or dx.stride(1) == 1)
    assert dout.dim() == 2 and dout.stride(1) == 1 and dout.dtype == torch.float32
    self._ck(self.lib.er_cross_v2_epilogue_bwd_acc(_p(x0), _p(x), _p(u), _p(bias), ctypes.c_float(diag_scale), _p(dout),
                                                   ctypes.c_int32(dout.stride(0)), B, d, _p(dx0), ctypes.c_int32(dx0.stride(0)), int(bool(acc0)), _p(dx),
                                                   ctypes.c_int32(dx.stride(0) if dx is not None else 0), int(bool(accx)),
                                                   _p(du), _stream()), 'er_cross_v2_epilogue_bwd_acc')
    return du

  # -- K8 DIN
  def din_concat_fwd(self, q, h):
    B, L, E = h.shape
    out = torch.empty(B, L, 4 * E, dtype=torch.float32, device=h.device)
    self._ck(self.lib.er_din_concat_fwd(_p(_f32c(q)), _p(_f32c(h)), B, L, E, _p(out), _stream()),
             'er_din_concat_fwd')
    return out

  def din_concat_bwd(self, q, h, dout, dh=None, acc_h=False):
    """dh given: the history's gradient buffer of the step (kernels.grad_slot); acc_h: add into it."""
    B, L, E = h.shape
    dq = torch.empty_like(q)
    if dh is None:
      assert not acc_h
      dh = torch.empty_like(h)
    assert dh.shape == h.shape and dh.is_contiguous()
    self._ck(
        self.lib.er_din_concat_bwd(_p(q), _p(h), _p(_f32c(dout)), B, L, E, _p(dq), 0, _p(dh), int(bool(acc_h)), _stream()),
        'er_din_concat_bwd')
    return dq, dh

  # DIN's first attention layer on the GENERATED [q, h, q - h, q * h] operand (er_din_gemm_*): the [B, L, 4E] block exists
  # neither forward nor backward.  A/B switch: '0' = er_din_concat + the ordinary contractions.
  din_fused = os.environ.get('EASYREC_AMD_DIN_FUSED', '1') != '0'

  @staticmethod
  def din_gemm_ok(q, h, w):
    """q [B, E], h [B, L, E] contiguous fp32, w [4E, N]: shapes er_din_gemm_* take (E % 16 == 0, 16-byte aligned rows)."""
    if h.dim() != 3 or q.dim() != 2:
      return False
    B, L, E = h.shape
    return (E % 16 == 0 and L >= 2 and q.shape == (B, E) and q.is_contiguous() and h.is_contiguous() and w.shape[0] == 4 * E and
            w.stride(1) == 1 and w.stride(0) % 4 == 0 and q.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0 and
            w.data_ptr() % 16 == 0 and B * L * L < 2 ** 32)

  def din_gemm_fwd(self, q, h, w, bias, col_stats=None):
    """z [B * L, N] = [q, h, q - h, q * h] . w (+ bias) without building the block; col_stats as gemm()'s."""
    B, L, E = h.shape
    N = w.shape[1]
    z = torch.empty(B * L, N, dtype=torch.float32, device=h.device)
    self._log_gemm('gemm_f32_din_kernel<true, false, 1>', None, B * L, N, 4 * E)
    self._ck(self.lib.er_din_gemm_fwd(_p(q), ctypes.c_int32(q.stride(0)), _p(h), ctypes.c_int32(E), B, L, E, _p(w),
                                      ctypes.c_int32(w.stride(0)), N, _p(bias), _p(z), ctypes.c_int32(N), _p(col_stats),
                                      _stream()), 'er_din_gemm_fwd')
    return z

  def din_gemm_wgrad(self, q, h, dz, out, accumulate=True):
    """out [4E, N] (+)= [q, h, q - h, q * h]^T . dz"""
    B, L, E = h.shape
    N = dz.shape[1]
    assert dz.shape[0] == B * L and dz.stride(1) == 1 and out.shape == (4 * E, N) and out.stride(1) == 1
    self._log_gemm('gemm_f32_din_kernel<false, false, 1>', None, 4 * E, N, B * L)
    self._ck(self.lib.er_din_gemm_wgrad(_p(q), ctypes.c_int32(q.stride(0)), _p(h), ctypes.c_int32(E), B, L, E, _p(dz),
                                        ctypes.c_int32(dz.stride(0)), N, _p(out), ctypes.c_int32(out.stride(0)),
                                        int(bool(accumulate)), _stream()), 'er_din_gemm_wgrad')
    return out

  def din_gemm_dgrad(self, dz, w, q, h, dh=None, acc_h=False):
    """(dq [B, E], dh [B, L, E]) from dz [B * L, N] and w [4E, N] without building dcat = dz . w^T; dh given: the history's
    gradient buffer of the step (kernels.grad_slot), acc_h: add into it."""
    B, L, E = h.shape
    N = dz.shape[1]
    dq = torch.empty(B, E, dtype=torch.float32, device=h.device)
    if dh is None:
      assert not acc_h
      dh = torch.empty_like(h)
    assert dh.shape == h.shape and dh.is_contiguous()
    self.lib.er_din_dq_partial_floats.restype = ctypes.c_int64
    n_part = int(self.lib.er_din_dq_partial_floats(B, L, E))
    partial = torch.empty(n_part, dtype=torch.float32, device=h.device)
    self._log_gemm('gemm_f32_din_kernel<true, true, 2>', None, B * L, 4 * E, N)
    self._ck(self.lib.er_din_gemm_dgrad(_p(dz), ctypes.c_int32(dz.stride(0)), N, _p(w), ctypes.c_int32(w.stride(0)), _p(q),
                                        ctypes.c_int32(q.stride(0)), _p(h), ctypes.c_int32(E), B, L, E, _p(dq), ctypes.c_int32(E),
                                        _p(dh), ctypes.c_int32(E), int(bool(acc_h)), _p(partial), _stream()), 'er_din_gemm_dgrad')
    return dq, dh

  def din_pool_fwd(self, scores, hist, seq_len, scale=1.0):
    B, L, E = hist.shape
    probs = torch.empty(B, L, dtype=torch.float32, device=hist.device)
    out = torch.empty(B, E, dtype=torch.float32, device=hist.device)
    self._ck(
        self.lib.er_din_pool_fwd(_p(_f32c(scores)), _p(_f32c(hist)), _p(seq_len), B, L, E, ctypes.c_float(scale),
                                 _p(probs), _p(out), _stream()), 'er_din_pool_fwd')
    return out, probs

  def din_pool_bwd(self, probs, hist, seq_len, dout, scale=1.0, dhist=None, acc_h=False):
    B, L, E = hist.shape
    dscores = torch.empty(B, L, dtype=torch.float32, device=hist.device)
    if dhist is None:
      assert not acc_h
      dhist = torch.empty_like(hist)
    assert dhist.shape == hist.shape and dhist.is_contiguous()
    self._ck(
        self.lib.er_din_pool_bwd(_p(probs), _p(hist), _p(seq_len), _p(_f32c(dout)), B, L, E,
                                 ctypes.c_float(scale), _p(dscores), _p(dhist), int(bool(acc_h)), _stream()), 'er_din_pool_bwd')
    return dscores, dhist

  # -- K1b hash-table (KV) embedding tables
  def kv_create(self, var_rows, capacity, seed, init_mean, init_stddev, filter_freq=0, steps_to_live=0, step=None):
    """The map of one KV table whose arena is `var_rows` ([capacity, dim] view of the table group's storage).
    filter_freq > 1: keys get their row once seen that often (the map then tracks every id SEEN: 4x the slots);
    steps_to_live > 0: every training lookup stamps the key with `step` (int64 device counter) for evict at save time."""
    dev = var_rows.device
    filter_freq, steps_to_live = int(filter_freq), int(steps_to_live)
    slots = 1
    while slots < (8 if filter_freq > 1 else 2) * int(capacity):
      slots *= 2
    assert slots <= (1 << 31), 'hash-table embedding: %d map slots (the kernels count slots and rows in 32 bits)' % slots
    kv = {'keys': torch.full((slots,), -1, dtype=torch.int64, device=dev),
          'rows': torch.full((slots,), -1, dtype=torch.int32, device=dev),
          'next_row': torch.zeros(1, dtype=torch.int32, device=dev),
          'overflow': torch.zeros(1, dtype=torch.int32, device=dev),
          'capacity': int(capacity), 'var': var_rows, 'dim': int(var_rows.shape[1]), 'seed': int(seed) & ((1 << 63) - 1),
          'mean': float(init_mean), 'stddev': float(init_stddev), 'filter_freq': filter_freq,
          'steps_to_live': steps_to_live, 'freq': None, 'version': None, 'n_keys': None, 'step': None}
    if filter_freq > 1:
      kv['freq'] = torch.zeros(slots, dtype=torch.int32, device=dev)
      kv['n_keys'] = torch.zeros(1, dtype=torch.int32, device=dev)
    if steps_to_live > 0:
      assert step is not None and step.dtype == torch.int64, 'steps_to_live needs the device step counter'
      kv['version'] = torch.zeros(slots, dtype=torch.int32, device=dev)
      kv['step'] = step
    return kv

  @staticmethod
  def _kv_job(kv, ids, rows_out, limit=None):
    opt = lambda t: 0 if t is None else t.data_ptr()
    return KvJob(ids.data_ptr(), ids.numel(), kv['keys'].data_ptr(), kv['rows'].data_ptr(), kv['keys'].numel(),
                 kv['next_row'].data_ptr(), kv['var'].data_ptr(), kv['seed'], rows_out.data_ptr(),
                 kv['overflow'].data_ptr(), kv['capacity'], kv['dim'], kv['mean'], kv['stddev'], opt(limit),
                 opt(kv['freq']), opt(kv['version']), opt(kv['n_keys']), opt(kv['step']), kv['filter_freq'],
                 kv['var'].stride(0) if kv['var'].stride(0) != kv['dim'] else 0)

  def kv_translate(self, kv, ids, rows_out, insert):
    """rows_out[i] = arena row of ids[i] (-1: no row); insert: a training lookup (unseen ids get a row, or - filtered
    tables - a count)."""
    assert ids.dtype == torch.int64 and rows_out.dtype == torch.int64 and ids.is_contiguous() and rows_out.is_contiguous()
    assert ids.numel() == rows_out.numel()
    job = self._kv_job(kv, ids, rows_out)
    self._ck(self.lib.er_kv_translate_job(ctypes.byref(job), int(bool(insert)), _stream()), 'er_kv_translate_job')

  def kv_jobs_create(self, jobs):
    """jobs: [(kv, ids, rows_out[, n_limit])] -> the device-resident descriptor table of er_kv_translate_multi (built
    once).  n_limit: int32 device scalar = the number of valid ids of the step (ragged lists in fixed buffers)."""
    n = len(jobs)
    arr = (KvJob * n)()
    starts = [0]
    for i, job in enumerate(jobs):
      kv, ids, rows_out = job[:3]
      limit = job[3] if len(job) > 3 else None
      assert limit is None or (limit.dtype == torch.int32 and limit.numel() == 1)
      assert ids.dtype == torch.int64 and rows_out.dtype == torch.int64 and ids.is_contiguous() and rows_out.is_contiguous()
      arr[i] = self._kv_job(kv, ids, rows_out, limit)
      starts.append(starts[-1] + (ids.numel() + 255) // 256)
    dev = jobs[0][1].device
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    blk = torch.tensor(starts, dtype=torch.int32, device=dev)
    return {'table': table, 'blk_start': blk, 'n': n, 'blocks': starts[-1], 'jobs': jobs}

  def kv_translate_multi(self, handle, insert):
    if handle['blocks'] == 0:
      return
    self._ck(self.lib.er_kv_translate_multi(_p(handle['table']), _p(handle['blk_start']), handle['n'], handle['blocks'],
                                            int(bool(insert)), _stream()), 'er_kv_translate_multi')

  def kv_route_create(self, jobs, world):
    """Embedding-parallel hash tables: jobs [(ids, rows_out[, n_limit])] -> the buffers and descriptor table of
    er_kv_bucket / er_kv_unbucket.  send / recv / owner_rows / back: int64 [world, C] with C = the jobs' id counts
    summed (job j's region of every owner's block is [off_j, off_j + n_j): a job fits even if all its ids have one owner)."""
    n = len(jobs)
    dev = jobs[0][0].device
    arr = (KvRouteJob * n)()
    starts, offs, slots = [0], [0], []
    for i, job in enumerate(jobs):
      ids, rows_out = job[:2]
      limit = job[2] if len(job) > 2 else None
      assert ids.dtype == torch.int64 and rows_out.dtype == torch.int64 and ids.is_contiguous() and rows_out.is_contiguous()
      assert ids.numel() == rows_out.numel() and (limit is None or (limit.dtype == torch.int32 and limit.numel() == 1))
      slot = torch.full((ids.numel(),), -1, dtype=torch.int32, device=dev)
      slots.append(slot)
      arr[i] = KvRouteJob(ids.data_ptr(), ids.numel(), 0 if limit is None else limit.data_ptr(), rows_out.data_ptr(),
                          slot.data_ptr(), offs[-1])
      starts.append(starts[-1] + (ids.numel() + 255) // 256)
      offs.append(offs[-1] + ids.numel())
    C = max(offs[-1], 1)
    buf = lambda: torch.full((world, C), -1, dtype=torch.int64, device=dev)
    return {'table': torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev),
            'blk_start': torch.tensor(starts, dtype=torch.int32, device=dev), 'n': n, 'blocks': starts[-1], 'world': int(world),
            'C': C, 'offs': offs, 'slots': slots, 'jobs': jobs, 'send': buf(), 'recv': buf(), 'owner_rows': buf(), 'back': buf(),
            'counts': torch.zeros(n * world, dtype=torch.int32, device=dev)}

  def kv_bucket(self, h):
    if h['blocks']:
      self._ck(self.lib.er_kv_bucket(_p(h['table']), _p(h['blk_start']), h['n'], h['blocks'], h['world'], ctypes.c_int64(h['C']),
                                     _p(h['send']), _p(h['counts']), _stream()), 'er_kv_bucket')

  def kv_unbucket(self, h):
    if h['blocks']:
      self._ck(self.lib.er_kv_unbucket(_p(h['table']), _p(h['blk_start']), h['n'], h['blocks'], h['world'],
                                       ctypes.c_int64(h['C']), _p(h['back']), _stream()), 'er_kv_unbucket')

  def kv_export(self, kv):
    """(keys ascending, arena rows) of the table's materialised ids (host sync)."""
    n_max = kv['capacity']
    keys = torch.empty(n_max, dtype=torch.int64, device=kv['keys'].device)
    rows = torch.empty(n_max, dtype=torch.int32, device=kv['keys'].device)
    count = torch.zeros(1, dtype=torch.int32, device=kv['keys'].device)
    self._ck(self.lib.er_kv_export(_p(kv['keys']), _p(kv['rows']), ctypes.c_int64(kv['keys'].numel()), _p(keys), _p(rows),
                                   _p(count), _stream()), 'er_kv_export')
    n = int(count.item())
    keys, rows = keys[:n], rows[:n]
    order = torch.argsort(keys)
    return keys[order], rows[order].to(torch.int64)

  def kv_export_all(self, kv):
    """Every key the map holds, ascending: (keys, arena rows (-1: no row yet), freq, version) (host sync)."""
    slots, dev = kv['keys'].numel(), kv['keys'].device
    # (one record per map slot: a launch that crosses the load-factor threshold may claim more than slots / 2 of them while
    # raising only the sticky overflow flag - the export must stay in bounds whatever the map holds)
    n_max = slots
    keys = torch.empty(n_max, dtype=torch.int64, device=dev)
    rows, freq, version = (torch.empty(n_max, dtype=torch.int32, device=dev) for _ in range(3))
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    opt = lambda t: None if t is None else _p(t)
    self._ck(self.lib.er_kv_export_all(_p(kv['keys']), _p(kv['rows']), opt(kv['freq']), opt(kv['version']),
                                       ctypes.c_int64(slots), _p(keys), _p(rows), _p(freq), _p(version), _p(count), _stream()),
             'er_kv_export_all')
    n = int(count.item())
    order = torch.argsort(keys[:n])
    return keys[:n][order], rows[:n][order].to(torch.int64), freq[:n][order], version[:n][order]

  def kv_rebuild(self, kv, keys, rows, freq=None, version=None):
    """Replace the map's content: distinct `keys` with their arena `rows` (-1: tracked, no row yet), counts and stamps.
    The rows in use become 0 .. (number of rows >= 0) - 1: the caller has laid the arena out that way."""
    dev = kv['keys'].device
    keys = keys.to(dev, torch.int64).contiguous()
    rows32 = rows.to(dev, torch.int32).contiguous()
    kv['keys'].fill_(-1)
    kv['rows'].fill_(-1)
    opt = lambda t: None if t is None else _p(t)
    if kv['freq'] is not None:
      kv['freq'].zero_()
      freq = None if freq is None else freq.to(dev, torch.int32).contiguous()
    if kv['version'] is not None:
      kv['version'].zero_()
      version = None if version is None else version.to(dev, torch.int32).contiguous()
    self._ck(self.lib.er_kv_rebuild(opt(keys), opt(rows32), opt(freq) if kv['freq'] is not None else None,
                                    opt(version) if kv['version'] is not None else None, ctypes.c_int64(keys.numel()),
                                    _p(kv['keys']), _p(kv['rows']), opt(kv['freq']), opt(kv['version']),
                                    ctypes.c_int64(kv['keys'].numel()), _p(kv['overflow']), _stream()), 'er_kv_rebuild')
    kv['next_row'].fill_(int((rows32 >= 0).sum().item()))
    if kv['n_keys'] is not None:
      kv['n_keys'].fill_(keys.numel())

  # -- K9b CIN (xDeepFM)
  def cin_outer_fwd(self, xi, strides, H, x0, z):
    """z[(b, d), h * H0 + m] = xi[b, h, d] * x0[b, m, d]; xi addressed by strides = (stride_b, stride_h, stride_d)."""
    B, H0, D = x0.shape
    assert z.shape == (B * D, H * H0) and z.is_contiguous() and x0.is_contiguous()
    self._ck(self.lib.er_cin_outer_fwd(_p(xi), ctypes.c_int64(strides[0]), ctypes.c_int32(strides[1]),
                                       ctypes.c_int32(strides[2]), ctypes.c_int32(H), _p(x0), ctypes.c_int32(H0),
                                       ctypes.c_int32(D), ctypes.c_int64(B), _p(z), _stream()), 'er_cin_outer_fwd')

  def cin_act_pool_fwd(self, c, bias, B, D, pooled, col0):
    """c [B * D, N] <- relu(c + bias) in place; pooled[:, col0 : col0 + N] = its sum over d."""
    N = c.shape[1]
    assert c.is_contiguous() and pooled.stride(1) == 1
    self._ck(self.lib.er_cin_act_pool_fwd(_p(c), _p(bias), ctypes.c_int64(B), ctypes.c_int32(D), ctypes.c_int32(N),
                                          _p(pooled), ctypes.c_int32(pooled.stride(0)), ctypes.c_int32(col0), _stream()),
             'er_cin_act_pool_fwd')

  def cin_act_pool_bwd(self, fm, dpooled, col0, dnext, B, D, dc):
    N = fm.shape[1]
    assert fm.is_contiguous() and dc.is_contiguous() and dpooled.stride(1) == 1 and (dnext is None or dnext.is_contiguous())
    self._ck(self.lib.er_cin_act_pool_bwd(_p(fm), _p(dpooled), ctypes.c_int32(dpooled.stride(0)), ctypes.c_int32(col0),
                                          _p(dnext), ctypes.c_int64(B), ctypes.c_int32(D), ctypes.c_int32(N), _p(dc),
                                          _stream()), 'er_cin_act_pool_bwd')

  def cin_outer_bwd(self, dz, xi, strides, H, x0, dxi, add_xi, dx0):
    B, H0, D = x0.shape
    assert dz.shape == (B * D, H * H0) and dz.is_contiguous() and dx0.is_contiguous()
    self._ck(self.lib.er_cin_outer_bwd(_p(dz), _p(xi), ctypes.c_int64(strides[0]), ctypes.c_int32(strides[1]),
                                       ctypes.c_int32(strides[2]), ctypes.c_int32(H), _p(x0), ctypes.c_int32(H0),
                                       ctypes.c_int32(D), ctypes.c_int64(B), _p(dxi), int(bool(add_xi)), _p(dx0),
                                       _stream()), 'er_cin_outer_bwd')

  # -- K9 MLP pieces
  def bn_act_fwd(self, x, bias, gamma, beta, use_bn, eps, momentum, moving_mean, moving_var, act):
    B, N = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device) if use_bn else None
    invstd = torch.empty(N, dtype=torch.float32, device=x.device) if use_bn else None
    self._ck(
        self.lib.er_bn_act_fwd(_p(_f32c(x)), _p(bias), _p(gamma), _p(beta), B, N, int(use_bn),
                               ctypes.c_float(eps), ctypes.c_float(momentum), _p(moving_mean), _p(moving_var),
                               int(act), _p(y), _p(mean), _p(invstd), _stream()), 'er_bn_act_fwd')
    return y, mean, invstd

  def copy_multi(self, pairs):
    """[(dst, src)] device tensors (same byte size, contiguous) -> all copies in one launch (er_copy_multi)."""
    n = len(pairs)
    if n == 0:
      return
    srcs = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in pairs])
    dsts = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    nb = (ctypes.c_int64 * n)(*[s_.numel() * s_.element_size() for _, s_ in pairs])
    self._ck(self.lib.er_copy_multi(srcs, dsts, nb, n, _stream()), 'er_copy_multi')

  concat_pitch = os.environ.get('EASYREC_AMD_CONCAT_PITCH', '1') != '0'  # A/B switch

  def concat_cols(self, parts, bf16_state=None):
    """torch.cat(parts, dim=1) of 2-D fp32 blocks (unit inner stride) as one library launch.  bf16_state (a Bf16Shadows):
    the launch also writes the bf16 copy the next contraction reads, registered there."""
    n = len(parts)
    B = parts[0].shape[0]
    for t in parts:
      assert t.dim() == 2 and t.shape[0] == B and t.stride(1) == 1 and t.dtype == torch.float32
    # rows at a pitch that is a multiple of four floats (DeepFM's [wide | fm | deep] is 81 wide): the consumers' GEMMs
    # then see a 16-byte aligned operand and take their branch-free 16-byte loads instead of the masked scalar path
    width = sum(t.shape[1] for t in parts)
    pitch = (width + 3) // 4 * 4 if self.concat_pitch else width
    out = torch.empty(B, pitch, dtype=torch.float32, device=parts[0].device)[:, :width]
    outb = None
    if bf16_state is not None and n <= 8:
      outb = torch.empty(B, bf16_state.pad8(width), dtype=torch.bfloat16, device=out.device)  # (k-tail zeroed by the kernel)
      bf16_state.register(out, outb)
    for i in range(0, n, 8):  # (more than 8 parts: several launches into column blocks of `out`)
      chunk = parts[i:i + 8]
      col0 = sum(t.shape[1] for t in parts[:i])
      dst = out[:, col0:]
      pp = (ctypes.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
      ww = (ctypes.c_int32 * len(chunk))(*[t.shape[1] for t in chunk])
      ll = (ctypes.c_int32 * len(chunk))(*[t.stride(0) for t in chunk])
      self._ck(self.lib.er_concat_cols_b16(pp, ww, ll, len(chunk), B, _p(dst), ctypes.c_int32(out.stride(0)), _p(outb),
                                           ctypes.c_int32(0 if outb is None else outb.stride(0)), _stream()),
               'er_concat_cols')
    return out

  def bn_act_bwd(self, x, bias, gamma, y, mean, invstd, dy, use_bn, act, need_bias, need_affine, into=None,
                 partial=None, beta=None, bf16_state=None):
    """into = (dbias_buf, dgamma_buf, dbeta_buf) (each may be None): accumulate the parameter gradients
    into those buffers (slices of the flat gradient buffer) instead of returning new tensors.
    partial: column sums [gemm_row_tiles(B)][N][2] already produced by gemm_bn_bwd (skips that pass).
    bf16_state (a Bf16Shadows): the launch also writes dx's bf16 copy, registered there for the input-gradient contraction."""
    B, N = x.shape
    dx = torch.empty_like(x)
    dev = x.device
    dxb = bf16_state.new_copy(dx) if bf16_state is not None else None
    ldb = ctypes.c_int32(0 if dxb is None else dxb.stride(0))
    acc = into is not None
    if acc:
      dbias, dgamma, dbeta = into
    else:
      dbias = torch.empty(N, dtype=torch.float32, device=dev) if need_bias else None
      dgamma = torch.empty(N, dtype=torch.float32, device=dev) if need_affine else None
      dbeta = torch.empty(N, dtype=torch.float32, device=dev) if need_affine else None
    assert dy.dim() == 2 and dy.stride(1) == 1 and dy.dtype == torch.float32
    if partial is not None:
      self._ck(
          self.lib.er_bn_act_bwd_from_partials_ld_b16(_p(x), _p(bias), _p(gamma), _p(y), _p(mean), _p(invstd), _p(dy),
                                                      ctypes.c_int32(dy.stride(0)), B, N, int(use_bn), int(act), _p(partial),
                                                      ctypes.c_int32(self.gemm_row_tiles(B)), _p(dx), _p(dbias), _p(dgamma),
                                                      _p(dbeta), int(acc), _p(dxb), ldb, _stream()),
          'er_bn_act_bwd_from_partials_ld')
    else:  # (dy may be a column block of a wider gradient - ConcatFn's backward -: read in place)
      self._ck(
          self.lib.er_bn_act_bwd_ld_b16(_p(x), _p(bias), _p(gamma), _p(y), _p(mean), _p(invstd), _p(dy),
                                        ctypes.c_int32(dy.stride(0)), B, N, int(use_bn), int(act), _p(dx), _p(dbias),
                                        _p(dgamma), _p(dbeta), int(acc), _p(dxb), ldb, _stream()), 'er_bn_act_bwd_ld')
    if acc:
      return dx, None, None, None
    return dx, dbias, dgamma, dbeta

  def colsum(self, x, out=None, accumulate=False):
    """out[j] (+)= sum_i x[i, j]; out: e.g. a bias' slice of the flat gradient buffer."""
    rows, cols = x.shape
    if out is None:
      assert not accumulate
      out = torch.empty(cols, dtype=torch.float32, device=x.device)
    self._ck(self.lib.er_colsum_acc(_p(x), rows, cols, x.stride(0), _p(out), int(bool(accumulate)), _stream()),
             'er_colsum_acc')
    return out

  def dice_fwd(self, x, alpha, eps, momentum, moving_mean, moving_var):
    B, N = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device)
    invstd = torch.empty(N, dtype=torch.float32, device=x.device)
    self._ck(
        self.lib.er_dice_fwd(_p(_f32c(x)), _p(alpha), B, N, ctypes.c_float(eps), ctypes.c_float(momentum),
                             _p(moving_mean), _p(moving_var), _p(y), _p(mean), _p(invstd), _stream()),
        'er_dice_fwd')
    return y, mean, invstd

  def dice_bwd(self, x, alpha, mean, invstd, dy):
    B, N = x.shape
    dx = torch.empty_like(x)
    dalpha = torch.empty(N, dtype=torch.float32, device=x.device)
    self._ck(
        self.lib.er_dice_bwd(_p(x), _p(alpha), _p(mean), _p(invstd), _p(_f32c(dy)), B, N, _p(dx), _p(dalpha),
                             _stream()), 'er_dice_bwd')
    return dx, dalpha

  # -- K10 loss / scalars
  def sigmoid_ce(self, logits, labels, weights, loss_scale=1.0):
    """Returns (loss [1], dlogits [B], probs [B])."""
    B = logits.numel()
    dev = logits.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dlogits = torch.empty(B, dtype=torch.float32, device=dev)
    probs = torch.empty(B, dtype=torch.float32, device=dev)
    self._ck(
        self.lib.er_sigmoid_ce_fwd_bwd(_p(_f32c(logits)), _p(_f32c(labels)), _p(weights), B,
                                       ctypes.c_float(loss_scale), _p(loss), _p(dlogits), _p(probs), _stream()),
        'er_sigmoid_ce_fwd_bwd')
    return loss, dlogits, probs

  def sigmoid_ce_multi(self, heads):
    """heads: [(logits, labels, weights or None, loss_scale)] -> [(loss [1], dlogits [B])] in one launch."""
    arr = (CeHead * len(heads))()
    outs, keep = [], []
    for q, (logits, labels, weights, scale) in zip(arr, heads):
      z, y = _f32c(logits), _f32c(labels)
      B = z.numel()
      loss = torch.empty(1, dtype=torch.float32, device=z.device)
      dz = torch.empty(B, dtype=torch.float32, device=z.device)
      q.logits, q.labels, q.weights = z.data_ptr(), y.data_ptr(), _ptr(weights)
      q.B, q.loss_scale = B, float(scale)
      q.loss_out, q.dlogits, q.probs_out = loss.data_ptr(), dz.data_ptr(), None
      outs.append((loss, dz))
      keep.append((z, y))
    self._ck(self.lib.er_sigmoid_ce_multi(arr, len(heads), _stream()), 'er_sigmoid_ce_multi')
    return outs

  def total_loss(self, reg_emb, reg_dense, losses, reports, reg_out, total_out):
    n = len(losses)
    src = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in losses])
    dst = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in reports])
    self._ck(self.lib.er_total_loss(_p(reg_emb), _p(reg_dense), src, dst, n, _p(reg_out), _p(total_out), _stream()),
             'er_total_loss')

  def reg_total_loss(self, emb_partials, emb_scale, dense_partials, losses, reports, reg_out, total_out):
    """reg_out = emb_scale * sum(emb_partials) + sum(dense_partials); total_out = reg_out + sum(losses); reports[i] =
    losses[i].  Either partial array may be None.  One launch (er_reg_total_loss)."""
    n = len(losses)
    assert n <= 8
    src = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in losses])
    dst = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in reports])
    n_part = 0 if emb_partials is None else emb_partials.numel()
    n_dense = 0 if dense_partials is None else dense_partials.numel()
    self._ck(self.lib.er_reg_total_loss(_p(emb_partials), ctypes.c_int32(n_part), ctypes.c_float(emb_scale),
                                        _p(dense_partials), ctypes.c_int32(n_dense), src, dst, ctypes.c_int32(n),
                                        _p(reg_out), _p(total_out), _stream()), 'er_reg_total_loss')

  # the binary head of a rank model (dense(K -> 1) + sigmoid cross entropy + their gradients) as ONE launch, its dW / db
  # partial sums folded into the loss tail's launch: layers/dnn.py dense(head=True), builders/loss_builder.py
  fused_head = os.environ.get('EASYREC_AMD_FUSED_HEAD', '1') != '0'  # A/B switch

  def head_sigmoid_ce(self, x, w, b, labels, loss_scale, src=None, logits=None):
    """er_head_sigmoid_ce: x [B, K] (row stride >= K), w [K, 1], b [1] or None, labels [B] -> dict(logits [B, 1], probs [B],
    dlogits [B], dx [B, K], loss_partials [T], wb_partials [T, K + 1], bn_partials [T, K, 2] or None); the loss is
    loss_scale * sum(loss_partials) / B (loss_tail).  src: the BnSource of x (x is its activation output)."""
    B, K = x.shape
    dev = x.device
    T = (B + 63) // 64
    f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
    out = {'logits': logits if logits is not None else f(B, 1), 'probs': f(B), 'dlogits': f(B), 'dx': f(B, K), 'loss_partials': f(T), 'wb_partials': f(T, K + 1),
           'bn_partials': None}
    sz = smean = sinv = None
    sld, sact = 0, ACT_NONE
    if src is not None:
      assert src.y is not None and src.y.data_ptr() == x.data_ptr() and src.zbias is None
      out['bn_partials'] = f(T, K, 2)
      sz, smean, sinv, sld, sact = src.z, src.mean, src.invstd, src.z.stride(0), src.act
    self._ck(self.lib.er_head_sigmoid_ce(_p(x), ctypes.c_int32(x.stride(0)), _p(w), _p(b), _p(_f32c(labels)), ctypes.c_int32(B),
                                         ctypes.c_int32(K), ctypes.c_float(loss_scale), _p(out['logits']), _p(out['probs']),
                                         _p(out['dlogits']), _p(out['dx']), _p(out['loss_partials']), _p(out['wb_partials']),
                                         _p(sz), ctypes.c_int32(sld), _p(smean), _p(sinv), ctypes.c_int32(int(sact)),
                                         _p(out['bn_partials']), _stream()), 'er_head_sigmoid_ce')
    return out

  def loss_tail(self, emb_partials, emb_scale, dense_partials, losses, reports, reg_out, total_out, jobs=(), defer=False):
    """reg_total_loss whose task losses may be PartialLoss records (per-workgroup partial sums left by head_sigmoid_ce)
    and which also runs small column-sum jobs [(partial [P, ld], dst [n_cols], n_cols)]: dst[j] += sum_p partial[p, j].
    defer: not launched now - the step's fused tail runs it as one workgroup of its first launch (emb_bwd_fused(tail=...));
    whoever does not get that far calls flush_loss_tail() before anything reads the losses or the gradient buffer."""
    n = len(losses)
    assert n <= 8 and len(jobs) <= 4
    src = (ctypes.c_void_p * max(n, 1))()
    dst = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in reports])
    parts = (ctypes.c_int32 * max(n, 1))()
    scales = (ctypes.c_float * max(n, 1))()
    divs = (ctypes.c_float * max(n, 1))()
    values = (ctypes.c_void_p * max(n, 1))()
    for i, t in enumerate(losses):
      pl = getattr(t, '_er_partials', None)
      if pl is not None:
        src[i], parts[i], scales[i], divs[i], values[i] = pl[0].data_ptr(), pl[0].numel(), float(pl[1]), float(pl[2]), t.data_ptr()
      else:
        src[i], parts[i], scales[i], divs[i], values[i] = t.data_ptr(), 0, 1.0, 1.0, None
    arr = (TailJob * max(len(jobs), 1))()
    for q, (partial, d, n_cols) in zip(arr, jobs):
      assert partial.dim() == 2 and partial.stride(1) == 1 and d.is_contiguous() and d.numel() == n_cols
      q.partial, q.dst, q.n_parts, q.n_cols, q.ld = partial.data_ptr(), d.data_ptr(), partial.shape[0], int(n_cols), partial.stride(0)
    n_part = 0 if emb_partials is None else emb_partials.numel()
    n_dense = 0 if dense_partials is None else dense_partials.numel()
    if defer:
      assert self._deferred_loss_tail is None, 'the previous deferred loss tail was never run'
      vp = ctypes.c_void_p
      job = LossTailJob(_ptr(emb_partials), n_part, float(emb_scale), _ptr(dense_partials), n_dense,
                        ctypes.cast(src, vp), ctypes.cast(dst, vp), ctypes.cast(parts, vp), ctypes.cast(scales, vp),
                        ctypes.cast(divs, vp), ctypes.cast(values, vp), n, ctypes.cast(arr, vp), len(jobs),
                        _ptr(reg_out), _ptr(total_out))
      # (the record points into the ctypes arrays and the tensors: both stay alive with it)
      self._deferred_loss_tail = (job, (src, dst, parts, scales, divs, values, arr, emb_partials, dense_partials, list(losses),
                                        list(reports), reg_out, total_out, list(jobs)))
      return
    self._ck(self.lib.er_loss_tail(_p(emb_partials), ctypes.c_int32(n_part), ctypes.c_float(emb_scale), _p(dense_partials),
                                   ctypes.c_int32(n_dense), src, dst, parts, scales, divs, values, ctypes.c_int32(n), arr,
                                   ctypes.c_int32(len(jobs)), _p(reg_out), _p(total_out), _stream()), 'er_loss_tail')

  # the deferred loss tail is per THREAD: the embedding-parallel tests run their ranks as threads over this one backend
  _tail_tls = threading.local()

  @property
  def _deferred_loss_tail(self):
    return getattr(HipBackend._tail_tls, 'pending', None)

  @_deferred_loss_tail.setter
  def _deferred_loss_tail(self, value):
    HipBackend._tail_tls.pending = value

  def discard_loss_tail(self):
    """Forget a deferred loss tail nobody ran (the step that queued it raised before its tail): called at the start of
    every step."""
    self._deferred_loss_tail = None

  def flush_loss_tail(self):
    """Run a deferred loss tail on its own (a step whose tail did not take it)."""
    pending = self._deferred_loss_tail
    if pending is None:
      return
    self._deferred_loss_tail = None
    j = pending[0]
    I32, F32, VP = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_void_p)
    self._ck(self.lib.er_loss_tail(ctypes.c_void_p(j.emb_partials), ctypes.c_int32(j.n_partials), ctypes.c_float(j.emb_scale),
                                   ctypes.c_void_p(j.dense_partials), ctypes.c_int32(j.n_dense), ctypes.cast(j.losses, VP),
                                   ctypes.cast(j.report, VP), ctypes.cast(j.loss_parts, I32), ctypes.cast(j.loss_scales, F32),
                                   ctypes.cast(j.loss_divs, F32), ctypes.cast(j.loss_values, VP), ctypes.c_int32(j.n_losses),
                                   ctypes.cast(j.jobs, ctypes.POINTER(TailJob)), ctypes.c_int32(j.n_jobs),
                                   ctypes.c_void_p(j.reg_out), ctypes.c_void_p(j.total_out), _stream()), 'er_loss_tail')

  def l2_partials(self, w, coef, partials):
    """partials[b] = sum over weights [256 b, 256 b + 256) of 0.5 * coef * w^2 (what dense_opt_step(l2_partials=) keeps
    current from then on)."""
    assert partials.numel() == (w.numel() + 255) // 256
    self._ck(self.lib.er_l2_partials(_p(w), _p(coef), ctypes.c_int64(w.numel()), _p(partials), _stream()),
             'er_l2_partials')

  def reduce_sum(self, partials, scale, out, accumulate=False):
    self._ck(
        self.lib.er_reduce_sum(_p(partials), partials.numel(), ctypes.c_float(scale), _p(out), int(accumulate),
                               _stream()), 'er_reduce_sum')

  def l2_loss(self, w, coef, out, accumulate=False):
    self._ck(self.lib.er_l2_loss(_p(w), _p(coef), ctypes.c_int64(w.numel()), _p(out), int(accumulate), _stream()),
             'er_l2_loss')

  # -- K11 MMoE
  def mmoe_mix_fwd(self, experts, gate_logits):
    E, B, H = experts.shape
    T = gate_logits.shape[0]
    gates = torch.empty_like(gate_logits)
    out = torch.empty(T, B, H, dtype=torch.float32, device=experts.device)
    self._ck(
        self.lib.er_mmoe_mix_fwd(_p(_f32c(experts)), _p(_f32c(gate_logits)), T, E, B, H, _p(gates), _p(out),
                                 _stream()), 'er_mmoe_mix_fwd')
    return out, gates

  def mmoe_mix_bwd(self, experts, gates, dout):
    E, B, H = experts.shape
    T = gates.shape[0]
    dexperts = torch.empty_like(experts)
    dlogits = torch.empty_like(gates)
    self._ck(
        self.lib.er_mmoe_mix_bwd(_p(experts), _p(gates), _p(_f32c(dout)), T, E, B, H, _p(dexperts), _p(dlogits),
                                 _stream()), 'er_mmoe_mix_bwd')
    return dexperts, dlogits

  def hyper_select(self, table, counter, out, history=None, history_index=HYPER_LR_T):
    """history: fp32 [2 * capacity]: value of step s at [s], the running maximum of the values at [capacity + s]."""
    n_slots = table.shape[0]
    cap = 0 if history is None else history.numel() // 2
    self._ck(self.lib.er_hyper_select(_p(table), _p(counter), n_slots, table[0].numel(), _p(out), _p(history),
                                      ctypes.c_int64(cap), ctypes.c_int32(history_index), _stream()),
             'er_hyper_select')

  def step_prologue(self, table, counter, out, history=None, zero=None, history_index=HYPER_LR_T, decay_tables=None,
                    hash_job=None):
    """hyper_select + zeroing of `zero` (the flat gradient buffer) in one launch.  decay_tables (decay_tables_create):
    the same launch appends the step's entry to the closed-form replay's per-step table.  hash_job = (bytes, offsets,
    n_per_col, num_buckets, drop_empty, out) (hash_bucket_fast's arguments): the batch's id strings are hashed by further
    workgroups of the same launch."""
    n_slots = table.shape[0]
    cap = 0 if history is None else history.numel() // 2
    nz = 0 if zero is None else zero.numel()
    assert zero is None or (zero.dtype == torch.float32 and zero.is_contiguous())
    hb = ho = hk = hout = None
    hn, hpc, hdrop = 0, 1, 0
    if hash_job is not None:
      hb, ho, hpc, hk, hdrop, hout = hash_job
      hn = ho.numel() - 1
      assert hout.dtype == torch.int64 and hout.numel() >= hn
    self._ck(self.lib.er_step_prologue_hash(_p(table), _p(counter), n_slots, table[0].numel(), _p(out), _p(history),
                                            ctypes.c_int64(cap), ctypes.c_int32(history_index), _p(zero),
                                            ctypes.c_int64(nz), decay_tables['handle'] if decay_tables else None,
                                            _p(hb), _p(ho), ctypes.c_int64(hn), ctypes.c_int64(int(hpc)), _p(hk),
                                            ctypes.c_int(int(hdrop)), _p(hout), _stream()), 'er_step_prologue_hash')

  # -- closed-form replay of TF-Adam's decay-only steps (csrc/er_decay.h)
  def decay_tables_create(self, lr_hist, step_counter, beta1, beta2):
    """The tables of the closed-form replay for the history buffer `lr_hist` ([2 * capacity]) - or None when the
    betas are outside its range (the groups then keep the exact step-by-step replay)."""
    self.lib.er_decay_tables_bytes.restype = ctypes.c_int64
    if self.lib.er_decay_tables_supported(ctypes.c_float(beta1), ctypes.c_float(beta2)) <= 0:
      return None
    cap = lr_hist.numel() // 2
    nbytes = int(self.lib.er_decay_tables_bytes(ctypes.c_int64(cap)))
    buf = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=lr_hist.device)
    handle = ctypes.c_void_p()
    self._ck(self.lib.er_decay_tables_create(_p(buf), ctypes.c_int64(cap), _p(lr_hist), _p(step_counter),
                                             ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.byref(handle)),
             'er_decay_tables_create')
    return {'handle': handle, 'buffer': buf, 'lr_hist': lr_hist, 'step_counter': step_counter,
            'betas': (float(beta1), float(beta2))}

  def decay_tables_destroy(self, tabs):
    if tabs is not None and tabs.get('handle') is not None:
      self.lib.er_decay_tables_destroy(tabs['handle'])
      tabs['handle'] = None

  def emb_group_set_decay_tables(self, group, tabs):
    self._ck(self.lib.er_emb_group_set_decay_tables(group['handle'], tabs['handle'] if tabs else None),
             'er_emb_group_set_decay_tables')
    group['decay_tables'] = tabs

  # -- TF-exact Adam without the sweep (lazy dense decay)
  def emb_group_enable_lazy_decay(self, group, last_step, lr_hist, step_counter):
    """lr_hist: the [2 * capacity] history buffer of hyper_select (values | running maxima)."""
    assert last_step.dtype == torch.int32 and lr_hist.dtype == torch.float32 and step_counter.dtype == torch.int64
    self._ck(self.lib.er_emb_group_enable_lazy_decay(group['handle'], _p(last_step), _p(lr_hist), _p(step_counter)),
             'er_emb_group_enable_lazy_decay')
    cap = lr_hist.numel() // 2
    if True:  # (the absorbed regime of the replay)
      self._ck(self.lib.er_emb_group_set_lr_max(group['handle'], _p(lr_hist[cap:])), 'er_emb_group_set_lr_max')
    group['last_step'], group['lr_hist'], group['step_counter'] = last_step, lr_hist, step_counter
    self._set_row_pitch(group)

  def emb_flush_window(self, groups, n_windows, hyper, lag=0, max_blocks=0):
    """Rolling flush: this step's window (step mod n_windows) of up to 4 table groups, one launch.  lag 1: the
    variant that runs concurrently with the step (after its catch-up, on another stream)."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    self._ck(self.lib.er_emb_flush_window(gh, n, ctypes.c_int32(int(n_windows)), ctypes.c_int32(int(lag)),
                                          ctypes.c_int32(int(max_blocks)), _p(hyper), _stream()),
             'er_emb_flush_window')

  def emb_catch_up(self, group, unique_keys, n_unique, hyper):
    self._ck(self.lib.er_emb_catch_up(group['handle'], _p(unique_keys), _p(n_unique), _p(hyper), _stream()),
             'er_emb_catch_up')

  def emb_catch_up_multi(self, groups, unique_keys, n_unique, hyper):
    """er_emb_catch_up for several table groups in one launch (same results)."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    uk = (ctypes.c_void_p * n)(*[t.data_ptr() for t in unique_keys])
    nu = (ctypes.c_void_p * n)(*[t.data_ptr() for t in n_unique])
    self._ck(self.lib.er_emb_catch_up_multi(gh, uk, nu, n, _p(hyper), _stream()), 'er_emb_catch_up_multi')

  def emb_bwd_update_multi(self, groups, opt_kind, hyper):
    """er_emb_bwd_update for several table groups: one tile launch and one fix launch for all (same results)."""
    n = len(groups)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    self._ck(self.lib.er_emb_bwd_update_multi(gh, n, ctypes.c_int(opt_kind), _p(hyper), _stream()),
             'er_emb_bwd_update_multi')

  # the end of an embedding-parallel step - owner fix | replicated apply | dense optimizer - as one launch - A/B switch
  ep_update_tail = os.environ.get('EASYREC_AMD_EP_UPDATE_TAIL', '1') != '0'

  def emb_owner_update_tail(self, groups, opt_kind, hyper, tables, dense_opt):
    """er_emb_owner_update_tail: emb_bwd_update_multi(groups) whose fix launch also carries emb_dense_apply(tables) and
    dense_opt_step(*dense_opt) (dense_opt = (w, m, v, grad, l2coef, kind, hyper, l2_partials)); <= 4 groups, <= 4 tables."""
    n, nt = len(groups), len(tables)
    gh = (ctypes.c_void_p * n)(*[g['handle'] for g in groups])
    descs = (DenseApplyDesc * max(nt, 1))()
    for i, (var, m, v, dense) in enumerate(tables):
      assert var.stride(1) == 1 and dense.stride(1) == 1 and dense.shape[0] == var.shape[0]
      assert all(t is None or t.stride() == var.stride() for t in (m, v))
      descs[i] = DenseApplyDesc(var.data_ptr(), None if m is None else m.data_ptr(), None if v is None else v.data_ptr(),
                                dense.data_ptr(), dense.stride(0), var.shape[1], var.shape[0], var.stride(0))
    w, m, v, grad, l2coef, kind, hyp, l2p = dense_opt
    oj = DenseOptJob(_ptr(w), _ptr(m), _ptr(v), _ptr(grad), _ptr(l2coef), w.numel(), int(kind), _ptr(hyp), _ptr(l2p))
    self._ck(self.lib.er_emb_owner_update_tail(gh, n, ctypes.c_int(opt_kind), _p(hyper), descs if nt else None, nt,
                                               ctypes.byref(oj), _stream()), 'er_emb_owner_update_tail')
    st = self._bf16_state_of(w)
    if st is not None:
      st.refresh()  # the weights' bf16 shadows follow the masters (one launch)

  def emb_flush_decay(self, group, hyper):
    self._ck(self.lib.er_emb_flush_decay(group['handle'], _p(hyper), _stream()), 'er_emb_flush_decay')

  # -- gradient clipping by global norm (compat/optimizers.py:365-376, 453-481)
  def gradsq_rows(self, x, cols, weight, acc, accumulate, counts=None, seg_stride=None):
    """acc[0] (+)= weight * sum of x[r, :cols]^2 over the valid rows: all of them, or (counts int32 [n_seg]) the rows
    with (r % seg_stride) < counts[r // seg_stride]; seg_stride None = the whole buffer is one segment."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and acc.dtype == torch.float32
    rows = x.shape[0]
    n_seg = 0 if counts is None else counts.numel()
    assert counts is None or counts.dtype == torch.int32
    stride = rows if seg_stride is None else int(seg_stride)
    self._ck(self.lib.er_gradsq_rows(_p(x), ctypes.c_int64(rows), ctypes.c_int32(int(cols)), ctypes.c_int32(x.stride(0)),
                                     _p(counts), ctypes.c_int32(n_seg), ctypes.c_int64(max(stride, 1)),
                                     ctypes.c_float(weight), _p(acc), int(bool(accumulate)), _stream()), 'er_gradsq_rows')

  def gradsq_dense(self, w, grad, l2coef, hyper, acc, accumulate=False):
    """acc[0] (+)= sum (hyper.grad_scale * grad + l2coef * w)^2: the dense gradient as er_dense_opt_step sees it."""
    self._ck(self.lib.er_gradsq_dense(_p(w), _p(grad), _p(l2coef), ctypes.c_int64(w.numel()), _p(hyper), _p(acc),
                                      int(bool(accumulate)), _stream()), 'er_gradsq_dense')

  def clip_scale(self, normsq, clip_norm, records, norm_out=None):
    """records [n, HYPER_FLOATS] (device): records[:, HYPER_CLIP] = clip_norm * min(1 / norm, 1 / clip_norm)."""
    assert records.dim() == 2 and records.shape[1] == HYPER_FLOATS and records.is_contiguous()
    self._ck(self.lib.er_clip_scale(_p(normsq), ctypes.c_float(clip_norm), _p(records), ctypes.c_int32(records.shape[0]),
                                    _p(norm_out), _stream()), 'er_clip_scale')

  def emb_apply_unique(self, group, keys, grads, n_unique, opt_kind, hyper):
    """The row-wise optimizer of emb_bwd_update on ready-made de-duplicated row sums (emb_bwd_reduce[_routed])."""
    assert keys.dtype == torch.int32 and n_unique.dtype == torch.int32 and grads.dim() == 2 and grads.stride(1) == 1
    self._ck(self.lib.er_emb_apply_unique(group['handle'], _p(keys), _p(grads), ctypes.c_int32(grads.stride(0)),
                                          _p(n_unique), ctypes.c_int(opt_kind), _p(hyper), _stream()), 'er_emb_apply_unique')

  # -- dense optimizer
  def dense_opt_step(self, w, m, v, grad, l2coef, opt_kind, hyper, l2_partials=None):
    """l2_partials: left holding the per-256-weight sums of 0.5 * l2coef * w_new^2 (the next step's kernel-L2 loss)."""
    self._ck(
        self.lib.er_dense_opt_step_l2(_p(w), _p(m), _p(v), _p(grad), _p(l2coef), ctypes.c_int64(w.numel()),
                                      ctypes.c_int(opt_kind), _p(hyper), _p(l2_partials), _stream()),
        'er_dense_opt_step_l2')
    st = self._bf16_state_of(w)
    if st is not None:
      st.refresh()  # the weights' bf16 shadows follow the masters (one launch)


_BACKEND = None


def hip():
  """The kernel backend used by every layer of this package."""
  global _BACKEND
  if _BACKEND is None:
    _BACKEND = HipBackend()
  return _BACKEND


# ---------------------------------------------------------------------------------------------
# autograd glue: dense activations flow through torch.autograd; each Function is one fused kernel
# forward and one backward.
# ---------------------------------------------------------------------------------------------
def _take_pending_bn(be, x, src, w, bf16):
  """x may be the output of a layer whose BatchNorm apply was deferred to its consumer (LinearBNActFn / DINFirstLayerFn with
  defer_apply): -> the pending record when the apply can run inside the staging of the contraction x . w
  (HipBackend.gemm_bn_a), else None - after running the apply as the launch of its own it would have been."""
  pin = src.pending if isinstance(src, BnSource) else None
  if pin is None:
    return None
  src.pending = None
  if x.dim() == 2 and x.stride(-1) == 1 and not bf16 and getattr(be, 'bn_in_staging', False) and \
      pin['y'].data_ptr() == x.data_ptr() and pin['y'].shape == x.shape and be.bn_a_ok(pin, w):
    return pin
  be.bn_apply_pending(pin)
  return None


class LinearFn(torch.autograd.Function):
  """y = x . W (+ b): tf.layers.dense without activation (reference layers/dnn.py:57-62).  Forward and both
  gradients are hand-written MFMA GEMMs (er_gemm_*).  The weight/bias gradients are accumulated straight
  into the variables' slices of the flat gradient buffer (`w_grad`, `b_grad`; zeroed once per step by
  VarStore.zero_grad), so autograd issues no per-variable add kernels; when no buffer is given they are
  returned the normal way."""

  @staticmethod
  def forward(ctx, x, w, b, w_grad, b_grad, bf16, src=None, sink=None):
    be = hip()
    pin = _take_pending_bn(be, x, src, w, bf16)
    x2 = x if x.stride(-1) == 1 else x.contiguous()
    y = be.gemm_bn_a(pin, w, b) if pin is not None else be.gemm(GEMM_NN, x2, w, bias=b, bf16=bf16)
    ctx.save_for_backward(x2, w)
    ctx.has_bias = b is not None
    ctx.w_grad, ctx.b_grad, ctx.bf16 = w_grad, b_grad, bf16
    ctx.sink = be.wgrad_sink()
    ctx.src = src if (src is not None and x2 is x and _bn_bwd_fusable(be, bf16) and src.fused) else None
    ctx.gsink = sink if x2 is x else None
    # (x's other consumers may share its gradient buffer: grad_slot)
    ctx.slots = grad_slots_of_step() if (x2 is x and x.dim() == 2 and x.is_contiguous()) else None
    return y

  @staticmethod
  def backward(ctx, dy):
    be = hip()
    x, w = ctx.saved_tensors
    dy = dy if dy.stride(-1) == 1 and dy.dim() == 2 else dy.contiguous()
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
      dx = _dgrad(be, dy, w, ctx.src, ctx.bf16, ctx.gsink, x, ctx.slots)
    with_bias = False
    if ctx.needs_input_grad[1]:
      # (a tall projection onto <= 4 columns: weight AND bias gradient from one pass over dy - HipBackend.wgrad_tall_narrow)
      with_bias = bool(ctx.has_bias and ctx.needs_input_grad[2] and ctx.b_grad is not None and ctx.w_grad is not None and
                       not ctx.bf16 and hasattr(be, 'wgrad_tall_narrow_ok') and be.wgrad_tall_narrow_ok(x, dy) and
                       ctx.w_grad.stride(-1) == 1 and ctx.b_grad.is_contiguous())
      if with_bias:
        be.wgrad_tall_narrow(x, dy, ctx.w_grad, accumulate=True, bias_grad=ctx.b_grad)
      else:
        dw = _wgrad(be, x, dy, ctx.w_grad, ctx.bf16, ctx.sink)
    if ctx.has_bias and ctx.needs_input_grad[2] and not with_bias:
      if ctx.b_grad is not None:
        be.colsum(dy, out=ctx.b_grad, accumulate=True)  # straight into the flat gradient buffer
      else:
        db = be.colsum(dy)
    return dx, dw, db, None, None, None, None, None


class HeadState(object):
  """A rank model's `output` projection (dense(K -> 1)) whose logits have not been computed yet: layers/dnn.py dense(head=True)
  registers it; builders/loss_builder.py either runs the fused head launch on it (HipBackend.head_sigmoid_ce: logits, loss,
  and the projection's whole backward) or materialises the logits with an ordinary GEMM (materialize_head)."""
  __slots__ = ('x', 'w', 'b', 'w_grad', 'b_grad', 'src', 'logits', 'state', 'dx', 'dz', 'bf16')

  def __init__(self, x, w, b, w_grad, b_grad, src, logits, bf16):
    self.x, self.w, self.b, self.w_grad, self.b_grad, self.src, self.logits, self.bf16 = x, w, b, w_grad, b_grad, src, logits, bf16
    self.state, self.dx, self.dz = 'pending', None, None

  @property
  def fusable(self):
    K = self.x.shape[1]
    return (self.state == 'pending' and self.w_grad is not None and (self.b is None or self.b_grad is not None) and
            K % 4 == 0 and K <= 256 and self.x.stride(0) % 4 == 0 and self.x.data_ptr() % 16 == 0 and self.w.data_ptr() % 16 == 0)


def pending_head(pred):
  """The HeadState whose logits buffer `pred` views (same storage start), while it is still pending; else None."""
  from easyrec_amd.core import context
  stack = context._stack()
  heads = getattr(stack[-1], 'heads', None) if stack else None
  if not heads:
    return None
  st = heads.get(pred.data_ptr())
  return st if (st is not None and st.state == 'pending' and pred.numel() == st.logits.numel()) else None


def materialize_head(st):
  """logits = x . w + b by the ordinary GEMM (the loss is not the fused sigmoid cross entropy)."""
  if st.state == 'pending':
    hip().gemm(GEMM_NN, st.x, st.w.detach(), out=st.logits, bias=None if st.b is None else st.b.detach(), bf16=st.bf16)
    st.state = 'materialized'


def materialize_pending_heads():
  from easyrec_amd.core import context
  stack = context._stack()
  for st in list(getattr(stack[-1], 'heads', {}).values()) if stack else []:
    materialize_head(st)


class HeadFn(torch.autograd.Function):
  """LinearFn for the logit head of a rank model (units = 1): the forward only RESERVES the logits; the loss builder fills
  them - together with the loss and this layer's complete backward (HeadState / HipBackend.head_sigmoid_ce) - or falls back
  to the GEMMs of LinearFn."""

  @staticmethod
  def forward(ctx, x, w, b, w_grad, b_grad, bf16, src, heads):
    x2 = x if x.stride(-1) == 1 else x.contiguous()
    logits = torch.empty(x2.shape[0], 1, dtype=torch.float32, device=x2.device)
    be = hip()
    # (the head launch itself is fp32 arithmetic on fp32 activations, whatever the dense dtype)
    fused = src is not None and x2 is x and src.fused and _bn_bwd_fusable(be, bf16)
    # (the state keeps a DETACHED alias of the logits: they are written after this forward returned, outside autograd's view)
    st = HeadState(x2, w.detach(), None if b is None else b.detach(), w_grad, b_grad, src if fused else None, logits.detach(), bf16)
    heads[logits.data_ptr()] = st
    ctx.st = st
    ctx.save_for_backward(x2, w)
    ctx.has_bias = b is not None
    ctx.sink = be.wgrad_sink()
    return logits

  @staticmethod
  def backward(ctx, dy):
    be = hip()
    st = ctx.st
    x, w = ctx.saved_tensors
    if st.state == 'fused':
      # the loss launch already produced dx and queued dW / db for the loss tail: dy must be the gradient it computed
      # (autograd may hand over a copy of the seed; build_loss_graph seeds exactly one gradient per logits tensor)
      assert dy.numel() == st.dz.numel()
      return st.dx, None, None, None, None, None, None, None
    assert st.state == 'materialized', 'the logits of a rank model\'s head were never computed (kernels.materialize_head)'
    dy = dy if dy.stride(-1) == 1 and dy.dim() == 2 else dy.contiguous()
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
      dx = _dgrad(be, dy, w, st.src, st.bf16, None)
    if ctx.needs_input_grad[1]:
      dw = _wgrad(be, x, dy, st.w_grad, st.bf16, ctx.sink)
    if ctx.has_bias and ctx.needs_input_grad[2]:
      if st.b_grad is not None:
        be.colsum(dy, out=st.b_grad, accumulate=True)
      else:
        db = be.colsum(dy)
    return dx, dw, db, None, None, None, None, None


def _bn_bwd_fusable(be, bf16):
  """May a consumer's input-gradient contraction emit the producer's BatchNorm-backward column sums?  fp32: er_gemm_f32_bn_bwd;
  bf16: the same epilogue of er_gemm_bf16_nt_epi (bf16 operands in HBM)."""
  if not getattr(be, 'fused_bn_bwd', False):
    return False
  return (not bf16) or bool(getattr(be, 'bf16_nt', False) and getattr(be, 'bf16_epilogues', False))


def _wgrad(be, x, dz, w_grad, bf16, sink):
  """dW = x^T . dz: queued for the grouped launch (accumulating into w_grad, a slice of the flat gradient buffer), else
  launched here.  Returns dW only without w_grad."""
  if w_grad is not None:
    if not bf16 and hasattr(be, 'wgrad_tall_narrow_ok') and be.wgrad_tall_narrow_ok(x, dz) and w_grad.stride(-1) == 1:
      be.wgrad_tall_narrow(x, dz, w_grad, accumulate=True)  # (a [rows, 1] operand would stall the grouped launch: see there)
    elif not sink.put(x, dz, w_grad, bf16):
      be.gemm(GEMM_TN, x, dz, out=w_grad, accumulate=True, bf16=bf16)
    return None
  return be.gemm(GEMM_TN, x, dz, bf16=bf16)


def _dgrad(be, dz, w, src, bf16, sink=None, x=None, slots=None):
  """dx = dz . W^T; when the input was the output of a fused dense + BatchNorm layer (src), the GEMM's epilogue also
  leaves that layer's BatchNorm-backward column sums in src.partial; when it was an embedding group output (sink),
  the GEMM accumulates straight into the group's gradient buffer and nothing is returned to autograd."""
  if sink is not None and sink.covers(0, w.shape[0]):
    dst, acc = sink.target(0, w.shape[0])
    be.gemm(GEMM_NT, dz, w, out=dst, accumulate=acc, bf16=bf16)
    sink.done()
    return None
  if src is None:
    slot = grad_slot(slots, x) if (x is not None and slots is not None) else None
    if slot is not None and not slot[2]:
      # another consumer of x already started its gradient (a cross layer's epilogue): this GEMM accumulates into it
      be.gemm(GEMM_NT, dz, w, out=slot[0], accumulate=True, bf16=bf16)
      return None
    if slot is not None:
      return be.gemm(GEMM_NT, dz, w, out=slot[0], bf16=bf16)
    return be.gemm(GEMM_NT, dz, w, bf16=bf16)
  M, N = dz.shape[0], w.shape[0]
  if isinstance(src, BnColsView):
    inner = src.src
    n_src = inner.y.shape[1]
    partial = torch.empty(be.gemm_row_tiles(M) * n_src * 2, dtype=torch.float32, device=dz.device)
    dx = be.gemm_bn_bwd(GEMM_NT, dz, w, inner, partial, col0=src.col0, **({'bf16': True} if bf16 else {}))
    if dx is None:  # (bf16: shapes the epilogue does not take)
      return be.gemm(GEMM_NT, dz, w, bf16=bf16)
    inner.partial, inner.dx_ptr = partial, dx.data_ptr() + 4 * src.col0  # (what the block's view of dx starts at)
    return dx
  partial = torch.empty(be.gemm_row_tiles(M) * N * 2, dtype=torch.float32, device=dz.device)
  dx = be.gemm_bn_bwd(GEMM_NT, dz, w, src, partial, **({'bf16': True} if bf16 else {}))
  if dx is None:
    return be.gemm(GEMM_NT, dz, w, bf16=bf16)
  src.partial, src.dx_ptr = partial, dx.data_ptr()
  return dx


class LinearBNActFn(torch.autograd.Function):
  """dense (+ bias) -> BatchNorm(train) -> activation of one DNN layer (reference layers/dnn.py:57-79) as TWO
  launches forward: the MFMA GEMM, whose epilogue adds the bias and emits per-row-tile column statistics, and
  one fused finalize + normalise + ReLU kernel.  Backward: column sums, fused finalize + dz, then the two
  gradient GEMMs; parameter gradients accumulate straight into the flat gradient buffer (`grad_bufs` =
  (kernel.grad, gamma.grad, beta.grad)).  Under BatchNorm d(loss)/d(bias) == 0, so the bias gets none."""

  @staticmethod
  def forward(ctx, x, w, b, gamma, beta, moving_mean, moving_var, eps, momentum, act, bf16, grad_bufs, src=None,
              sink=None, defer_apply=False):
    be = hip()
    pin = _take_pending_bn(be, x, src, w, bf16)
    x2 = x if x.stride(-1) == 1 else x.contiguous()
    M, N = x2.shape[0], w.shape[1]
    chunks = be.gemm_row_tiles(M)
    stats = torch.empty(chunks * N * 3, dtype=torch.float32, device=x2.device)
    if pin is not None:
      z = be.gemm_bn_a(pin, w, b, col_stats=stats)
    else:
      z = be.gemm(GEMM_NN, x2, w, bias=b, bf16=bf16, col_stats=stats)
    # (bf16: the BatchNorm launch writes the bf16 copy the next contraction reads; its backward the one the dgrad reads)
    ctx.b16 = be._bf16_state_of(w) if (bf16 and getattr(be, 'bf16_nt', False) and getattr(be, 'bf16_epilogues', False)) else None
    # defer_apply: the caller's NEXT op on y is the one consumer that runs this layer's BatchNorm finalize + apply inside its
    # own launch (DeepFM: WideFmConcatFn).  Until then y / mean / invstd are unwritten buffers.
    pend = None
    # (defer_apply == 'staging': the consumer is the next layer's LinearBNActFn - a tall layer, HipBackend.bn_in_staging)
    if defer_apply and not bf16 and _bn_bwd_fusable(be, bf16) and z.is_contiguous() and \
        getattr(be, 'bn_in_staging' if defer_apply == 'staging' else 'defer_bn_apply', False):
      y = torch.empty_like(z)
      mean = torch.empty(N, dtype=torch.float32, device=z.device)
      invstd = torch.empty(N, dtype=torch.float32, device=z.device)
      pend = dict(z=z, stats=stats, chunks=chunks, gamma=gamma.detach(), beta=beta.detach(), eps=eps, momentum=momentum,
                  moving_mean=moving_mean, moving_var=moving_var, act=act, y=y, mean=mean, invstd=invstd)
    else:
      y, mean, invstd = be.bn_apply_from_stats(z, None, stats, chunks, gamma, beta, eps, momentum, moving_mean,
                                               moving_var, act, **({'bf16_state': ctx.b16} if ctx.b16 is not None else {}))
    ctx.save_for_backward(x2, w, gamma, beta, z, y, mean, invstd)
    ctx.act, ctx.bf16, ctx.grad_bufs = act, bf16, grad_bufs
    ctx.sink = be.wgrad_sink()
    fused = _bn_bwd_fusable(be, bf16)
    ctx.src = src if (src is not None and x2 is x and fused and src.fused) else None
    ctx.gsink = sink if x2 is x else None
    gb = None if grad_bufs is None else (grad_bufs[1], grad_bufs[2])
    # (z already carries the bias)
    ctx.own = BnSource(z, None, y, mean, invstd, act, gamma, gb, beta=beta, fused=fused) if fused else None
    if pend is not None:
      ctx.own.pending = pend
    _bn_tls.last = ctx.own
    return y

  @staticmethod
  def backward(ctx, dy):
    be = hip()
    x, w, gamma, beta, z, y, mean, invstd = ctx.saved_tensors
    assert ctx.own is None or ctx.own.pending is None, 'a deferred BatchNorm apply was never run (kernels.finish_pending_bn)'
    wg, gg, betag = ctx.grad_bufs if ctx.grad_bufs is not None else (None, None, None)
    direct = gg is not None and betag is not None
    # (a column block of a wider gradient - ConcatFn's backward - is read in place by the BatchNorm backward)
    dyc = dy if (dy.dim() == 2 and dy.stride(1) == 1) else dy.contiguous()
    own, partial = ctx.own, None
    dgamma = dbeta = None
    if own is not None:
      # the consumer's dgrad GEMM already reduced the column sums, provided dy is exactly its output
      if own.partial is not None and dyc.data_ptr() == own.dx_ptr:
        partial = own.partial
      own.partial = None
    dz, _, dgamma, dbeta = be.bn_act_bwd(z, None, gamma, y, mean, invstd, dyc, 1, ctx.act, False, True,
                                         into=(None, gg, betag) if direct else None, partial=partial, beta=beta,
                                         **({'bf16_state': ctx.b16} if (ctx.b16 is not None and ctx.needs_input_grad[0]) else {}))
    dx = dw = None
    if ctx.needs_input_grad[0]:
      dx = _dgrad(be, dz, w, ctx.src, ctx.bf16, ctx.gsink)
    if ctx.needs_input_grad[1]:
      dw = _wgrad(be, x, dz, wg, ctx.bf16, ctx.sink)
    return dx, dw, None, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def finish_pending_bn(t):
  """Run the deferred BatchNorm finalize + apply behind tensor t (LinearBNActFn(defer_apply=True)) as a launch of its own, if
  it is still pending: what a consumer that cannot fold it calls before it reads t."""
  src = bn_source_of(t)
  if src is not None and src.pending is not None:
    hip().bn_apply_pending(src.pending)
    src.pending = None


class GroupedLinearFn(torch.autograd.Function):
  """The same-depth dense layers z_e = x_e . W_e (+ b_e) of E PARALLEL stacks - MMoE's experts, the task towers, the
  gates (reference layers/mmoe.py:62-83, model/multi_task_model.py:33-100, which issue them one after the other) - as ONE
  grouped launch forward (er_gemm_grouped_f32: bias in the epilogue, optionally the column statistics a BatchNorm needs)
  and ONE grouped launch for the input gradients; weight gradients join the step's grouped weight-gradient launch, bias
  gradients are column sums into the flat gradient buffer.  Layers that read the SAME input (the first depth of MMoE:
  every expert and gate reads the shared features) accumulate their input gradients into one buffer - or straight into the
  embedding group's gradient buffer (`sinks`) - one GEMM after the other, as the sequential form does.
  apply(E, stats_mask, sinks, x_0..x_{E-1}, W_0.., b_0.. (None ok)) -> (z_0 .. z_{E-1}, stats_0 .. stats_{E-1}) (stats_e:
  an empty tensor unless stats_mask[e])."""

  @staticmethod
  def forward(ctx, E, stats_mask, sinks, srcs, *args):
    be = hip()
    xs, ws, bs = args[:E], args[E:2 * E], args[2 * E:3 * E]
    # fzs[e]: None, or dict(bias, gamma, beta, moving_mean, moving_var, eps, act) - layer e's output also goes through the frozen
    # BatchNorm + activation in the launch's epilogue; the outputs then carry (y_e, save_e = [mean | invstd]) behind the statistics
    fzs = args[3 * E] if len(args) > 3 * E else None
    ctx.has_fzs = len(args) > 3 * E
    ctx.gsinks = [sk if (sk is not None and x.stride(-1) == 1) else None for sk, x in zip(sinks, xs)]
    # srcs[e]: the BnSource of x_e when it IS the output of a dense + BatchNorm(train) layer: the input-gradient launch
    # then also emits that layer's BatchNorm-backward column sums (er_gemm_problem.bn_*)
    fused = getattr(be, 'fused_bn_bwd', False)
    ctx.srcs = [sc if (fused and sc is not None and sc.fused and x.stride(-1) == 1) else None
                for sc, x in zip(srcs, xs)]
    xs = [x if x.stride(-1) == 1 else x.contiguous() for x in xs]
    zs, stats, problems, pre = [], [], [], []
    for e in range(E):
      M, N = xs[e].shape[0], ws[e].shape[1]
      z = torch.empty(M, N, dtype=torch.float32, device=xs[e].device)
      st = torch.empty(be.gemm_row_tiles(M) * N * 3, dtype=torch.float32, device=z.device) if stats_mask[e] else None
      zs.append(z)
      stats.append(st if st is not None else torch.empty(0, device=z.device))
      fz = None
      if fzs is not None and fzs[e] is not None:
        assert bs[e] is None and st is None
        fz = dict(fzs[e], y=torch.empty_like(z), save=torch.empty(2, N, dtype=torch.float32, device=z.device))
        pre += [fz['y'], fz['save']]
      elif fzs is not None:
        pre += [torch.empty(0, device=z.device), torch.empty(0, device=z.device)]
      problems.append((xs[e], ws[e].detach(), z, None if bs[e] is None else bs[e].detach(), False, st, None, fz))
    be.gemm_grouped(GEMM_NN, problems)
    ctx.save_for_backward(*xs, *ws)
    ctx.E, ctx.bs = E, bs
    ctx.wgrads = [w.grad if (w.requires_grad and w.grad is not None) else None for w in ws]  # slices of the flat buffer
    ctx.stats_mask = tuple(stats_mask)
    ctx.set_materialize_grads(False)  # (the statistics outputs get no gradient: no zero tensors to be filled for them)
    ctx.sink = be.wgrad_sink()
    ctx.mark_non_differentiable(*stats, *pre)
    return tuple(zs) + tuple(stats) + tuple(pre)

  @staticmethod
  def backward(ctx, *grads):
    be = hip()
    E = ctx.E
    saved = ctx.saved_tensors
    xs, ws = saved[:E], saved[E:2 * E]
    dzs = [None if g is None else (g if (g.dim() == 2 and g.stride(1) == 1) else g.contiguous()) for g in grads[:E]]
    # a stack output left out of the loss (a tower no loss reads) arrives as None (set_materialize_grads(False)): that
    # layer contributes nothing - no input gradient, no weight gradient, no bias gradient - as in the one-by-one form
    dxs = [None] * E
    need = [e for e in range(E) if ctx.needs_input_grad[4 + e] and dzs[e] is not None]
    by_input = {}
    for e in need:
      # readers of ONE input share an accumulation buffer: same storage start AND same shape / strides (a row prefix
      # x[:k] of x shares its base pointer and is a different input)
      by_input.setdefault((xs[e].data_ptr(), tuple(xs[e].shape), tuple(xs[e].stride())), []).append(e)
    solo = [es[0] for es in by_input.values() if len(es) == 1 and ctx.gsinks[es[0]] is None]
    if solo:
      problems = []
      for e in solo:
        dxs[e] = torch.empty_like(xs[e])
        src = ctx.srcs[e]
        if src is None:
          problems.append((dzs[e], ws[e], dxs[e], None, False))
        else:
          M, N = dxs[e].shape
          partial = torch.empty(be.gemm_row_tiles(M) * N * 2, dtype=torch.float32, device=dxs[e].device)
          # (a layer on the moving statistics: the launch also runs its elementwise backward - dxs[e] leaves as dz)
          dz_out = bool(getattr(src, 'frozen', False) and getattr(be, 'frozen_dz_epilogue', False) and not src.dz_done)
          problems.append((dzs[e], ws[e], dxs[e], None, False, None, (src, partial, dz_out)))
          src.partial, src.dx_ptr = partial, dxs[e].data_ptr()
          if dz_out:
            src.dz_done = True
      be.gemm_grouped(GEMM_NT, problems)
    for es in by_input.values():
      if len(es) == 1 and ctx.gsinks[es[0]] is None:
        continue
      sink = ctx.gsinks[es[0]]
      to_sink = sink is not None and sink.covers(0, ws[es[0]].shape[0])  # an embedding group output: into its gradient buffer
      if len(es) > 1 and _CAT_DGRAD and all(dzs[e].stride(1) == 1 for e in es):
        # readers of ONE input (MMoE's first depth: every expert and gate reads the shared features): dx = sum_e dz_e W_e^T
        # is ONE contraction over the layers' output columns laid side by side - the input-sized gradient is written once
        # instead of being read and re-written by a GEMM per layer (8 x 36 MB at B = 8192), for two small concat launches.
        # (The same for the weight gradients - x^T . [dz_1 | dz_2 | ...] as one TN GEMM outside the step's grouped launch,
        # then added to the layers' gradient slices - measured SLOWER: 2.23 against 2.11 ms,
        # profiles/r04_mmoe_cat_wgrad_ab.txt; the grouped launch's split-K scheduling is the better one.)
        dz_cat = be.concat_cols([dzs[e] for e in es])
        w_cat = be.concat_cols([ws[e].detach() for e in es])
        if to_sink:
          _dgrad(be, dz_cat, w_cat, None, False, sink)
        else:
          dxs[es[0]] = be.gemm(GEMM_NT, dz_cat, w_cat)
        continue
      if to_sink:
        for e in es:
          _dgrad(be, dzs[e], ws[e], None, False, sink)
        continue
      acc = torch.empty_like(xs[es[0]])  # readers of one input: one buffer, accumulated GEMM by GEMM; the sum is returned
      for j, e in enumerate(es):         # for the first of them, the others contribute nothing more
        be.gemm(GEMM_NT, dzs[e], ws[e], out=acc, accumulate=j > 0)
      dxs[es[0]] = acc
    dws, dbs = [None] * E, [None] * E
    narrow = []  # the bias gradients of narrow layers (tower heads [B, 1], gates [B, experts]): ONE launch for all of them
    for e in range(E):
      if dzs[e] is None:
        continue
      if ctx.needs_input_grad[4 + E + e]:
        dws[e] = _wgrad(be, xs[e], dzs[e], ctx.wgrads[e], False, ctx.sink)
      b = ctx.bs[e]
      # (a layer whose output feeds BatchNorm on batch statistics - stats_mask - has a bias gradient of exactly zero)
      if b is not None and not ctx.stats_mask[e] and ctx.needs_input_grad[4 + 2 * E + e]:
        if b.grad is not None:
          if hasattr(be, 'colsum_narrow_multi') and be.colsum_is_narrow(dzs[e]) and b.grad.is_contiguous():
            narrow.append((dzs[e], b.grad))
          else:
            be.colsum(dzs[e], out=b.grad, accumulate=True)
        else:
          dbs[e] = be.colsum(dzs[e])
    if len(narrow) == 1:
      be.colsum(narrow[0][0], out=narrow[0][1], accumulate=True)
    elif narrow:
      be.colsum_narrow_multi(narrow, accumulate=True)
    return (None, None, None, None) + tuple(dxs) + tuple(dws) + tuple(dbs) + ((None,) if ctx.has_fzs else ())


class BNFromStatsFn(torch.autograd.Function):
  """BatchNorm(train) + activation of a GEMM output whose per-row-tile column statistics its launch already produced
  (GroupedLinearFn): er_bn_apply_from_stats forward, er_bn_act_bwd backward (parameter gradients into the flat buffer)."""

  @staticmethod
  def forward(ctx, z, stats, gamma, beta, moving_mean, moving_var, eps, momentum, act, grad_bufs):
    be = hip()
    y, mean, invstd = be.bn_apply_from_stats(z, None, stats, be.gemm_row_tiles(z.shape[0]), gamma, beta, eps, momentum,
                                             moving_mean, moving_var, act)
    ctx.save_for_backward(z, gamma, y, mean, invstd)
    ctx.act, ctx.grad_bufs = act, grad_bufs
    ctx.beta = None if beta is None else beta.detach()  # (read by the backward's mask recomputation only)
    fused = getattr(be, 'fused_bn_bwd', False)
    ctx.own = BnSource(z, None, y, mean, invstd, act, gamma, grad_bufs, beta=beta, fused=fused) if fused else None
    _bn_tls.last = ctx.own
    return y

  @staticmethod
  def backward(ctx, dy):
    be = hip()
    z, gamma, y, mean, invstd = ctx.saved_tensors
    gg, betag = ctx.grad_bufs if ctx.grad_bufs is not None else (None, None)
    direct = gg is not None and betag is not None
    dyc = dy if (dy.dim() == 2 and dy.stride(1) == 1) else dy.contiguous()
    own, partial = ctx.own, None
    if own is not None:
      # the consumer's input-gradient launch already reduced the column sums, provided dy is exactly its output
      if own.partial is not None and dyc.data_ptr() == own.dx_ptr:
        partial = own.partial
      own.partial = None
    dz, _, dgamma, dbeta = be.bn_act_bwd(z, None, gamma, y, mean, invstd, dyc, 1, ctx.act, False, True,
                                         into=(None, gg, betag) if direct else None, partial=partial, beta=ctx.beta)
    return dz, None, dgamma, dbeta, None, None, None, None, None, None


class GroupedBNActFn(torch.autograd.Function):
  """The bias / BatchNorm / activation kernels that follow the same-depth dense layers of E parallel stacks
  (GroupedLinearFn) as ONE launch forward and two backward: layer e is BNFromStatsFn (cfg mode BN_BATCH: the statistics its
  GEMM's epilogue produced) or BNActFn (BN_NONE, BN_FROZEN) - same kernels' bodies, same bits, E times fewer launches.
  apply(E, cfgs, z_0.., stats_0.., bias_0.., gamma_0.., beta_0..) -> y_0 .. y_{E-1}; cfgs[e] = (mode, act, moving_mean,
  moving_var, eps, momentum, grad_bufs) with grad_bufs = (bias.grad, gamma.grad, beta.grad) slices of the flat gradient
  buffer or None."""

  @staticmethod
  def forward(ctx, E, cfgs, *args):
    be = hip()
    zs, stats, biases, gammas, betas = (args[i * E:(i + 1) * E] for i in range(5))
    # pres[e] = (y, save [2, N]): layer e was already applied by its contraction's epilogue (GroupedLinearFn, fzs) - only its
    # backward is this Function's
    pres = args[5 * E:6 * E] if len(args) >= 6 * E else (None,) * E
    layers, todo = [], []
    for e in range(E):
      mode, act, mm, mv, eps, momentum, _ = cfgs[e]
      if pres[e] is not None:
        continue
      todo.append(e)
      layers.append(dict(x=zs[e], bias=biases[e], gamma=gammas[e], beta=betas[e], moving_mean=mm, moving_var=mv,
                         col_stats=stats[e], use_bn=mode, act=act, eps=eps, momentum=momentum))
    done = be.bn_fwd_multi(layers) if layers else []
    outs = [None] * E
    for e, o in zip(todo, done):
      outs[e] = o
    for e in range(E):
      if pres[e] is not None:
        y, save = pres[e]
        outs[e] = (y.detach(), save[0], save[1])  # (a new tensor object over the same memory: this Function's output)
    fused = getattr(be, 'fused_bn_bwd', False)
    ctx.owns, saved = [], []
    for e in range(E):
      y, mean, invstd = outs[e]
      mode, act = cfgs[e][0], cfgs[e][1]
      gb = cfgs[e][6]
      own = None
      if fused and mode == BN_BATCH:
        own = BnSource(zs[e], None, y, mean, invstd, act, gammas[e], None if gb is None else (gb[1], gb[2]), beta=betas[e],
                       fused=True)
      elif fused and mode == BN_FROZEN and getattr(be, 'frozen_dz_epilogue', False) and gammas[e] is not None:
        # (z is the bare contraction: the bias enters through zbias)
        own = BnSource(zs[e], None if biases[e] is None else biases[e].detach(), y, mean, invstd, act, gammas[e].detach(), None,
                       beta=betas[e], fused=True)
        own.frozen = True
      ctx.owns.append(own)
      saved += [zs[e], biases[e], gammas[e], betas[e], y, mean, invstd]
    ctx.save_for_backward(*saved)
    ctx.E, ctx.cfgs = E, cfgs
    ctx.n_extra = len(args) - 5 * E
    _bn_tls.last = ctx.owns if any(o is not None for o in ctx.owns) else None  # (the caller tags the outputs)
    return tuple(o[0] for o in outs)

  @staticmethod
  def backward(ctx, *dys):
    be = hip()
    E = ctx.E
    saved = ctx.saved_tensors
    layers = []
    for e in range(E):
      z, bias, gamma, beta, y, mean, invstd = saved[7 * e:7 * e + 7]
      mode, act, gb = ctx.cfgs[e][0], ctx.cfgs[e][1], ctx.cfgs[e][6]
      dy = dys[e] if (dys[e].dim() == 2 and dys[e].stride(1) == 1) else dys[e].contiguous()
      own, partial, dx_done = ctx.owns[e], None, False
      if own is not None:
        # the consumer's input-gradient launch already reduced the column sums, provided dy is exactly its output
        if own.partial is not None and dy.data_ptr() == own.dx_ptr and dy.stride(0) == dy.shape[1]:
          partial = own.partial
          dx_done = own.dz_done  # (... and, on frozen statistics, turned dy into dz: only the parameter gradients are left)
        else:
          assert not own.dz_done, 'a contraction wrote dz for a frozen BatchNorm layer whose backward received another tensor'
        own.partial, own.dz_done = None, False
      into = None
      if gb is not None:
        bg, gg, betag = gb
        if (bias is None or bg is not None) and (gamma is None or (gg is not None and betag is not None)):
          into = (bg if bias is not None else None, gg, betag)
      layers.append(dict(x=z, bias=bias, gamma=gamma, beta=beta, y=y, mean=mean, invstd=invstd, dy=dy, use_bn=mode, act=act,
                         partial=partial, into=into, dx_done=dx_done))
    outs = be.bn_bwd_multi(layers)
    dzs = tuple(o[0] for o in outs)
    none = (None,) * E
    return (None, None) + dzs + none + tuple(o[1] for o in outs) + tuple(o[2] for o in outs) + tuple(o[3] for o in outs) + \
        (None,) * ctx.n_extra


class FMFn(torch.autograd.Function):
  """reference layers/fm.py:20-26 over a [B, F*D] block of the input-layer output."""

  @staticmethod
  def forward(ctx, x, F, D, sink=None, col0=0):
    fm, S = hip().fm_fwd(x, F, D)
    ctx.save_for_backward(x, S)
    ctx.F, ctx.D = F, D
    ctx.sink, ctx.col0 = sink, col0
    return fm

  @staticmethod
  def backward(ctx, g):
    x, S = ctx.saved_tensors
    if ctx.sink is not None:
      # x is (a column block of) an embedding group output: its gradient g * (S - x) is added when the group's gradient
      # buffer is finished (er_group_grad_finish: one launch for all groups and terms)
      ctx.sink.defer(('fm', g if g.stride(-1) == 1 else g.contiguous(), S, ctx.col0, ctx.F * ctx.D, ctx.D))
      return None, None, None, None, None
    dx = hip().fm_bwd(x, S, g.contiguous(), ctx.F, ctx.D)
    if x.shape[1] != ctx.F * ctx.D:
      full = torch.zeros_like(x)
      full[:, :ctx.F * ctx.D] = dx
      dx = full
    return dx, None, None, None, None


class WideFmConcatFn(torch.autograd.Function):
  """DeepFM's [reduce_sum(wide) | FM(fields) | deep] (reference model/deepfm.py:60-83) as one launch; the wide and FM
  blocks live in embedding group outputs, so their gradients are the deferred terms RowSumFn / FMFn would register
  (finished with the group's gradient buffer), and `deep` gets its column block of the incoming gradient as a view."""

  @staticmethod
  def forward(ctx, wide, fm_x, deep, F, D, wide_sink, fm_sink, col0):
    be = hip()
    src = bn_source_of(deep)
    res = None
    if src is not None and src.pending is not None:
      # the deep tower's last BatchNorm finalize + apply rides in this launch (er_bn_apply_wide_fm)
      res = be.bn_apply_wide_fm(src.pending, wide, fm_x, F, D)
      if res is None:
        be.bn_apply_pending(src.pending)
      src.pending = None
    out, S = res if res is not None else be.wide_fm_concat(wide, fm_x, F, D, deep if deep.stride(-1) == 1 else deep.contiguous())
    ctx.save_for_backward(S)
    ctx.F, ctx.D, ctx.n_w = F, D, wide.shape[1]
    ctx.wide_sink, ctx.fm_sink, ctx.col0 = wide_sink, fm_sink, col0
    return out

  @staticmethod
  def backward(ctx, g):
    S, = ctx.saved_tensors
    D = ctx.D
    g = g if g.stride(-1) == 1 else g.contiguous()
    ctx.wide_sink.defer(('rowsum', g[:, 0:1], 0, ctx.n_w))
    ctx.fm_sink.defer(('fm', g[:, 1:1 + D], S, ctx.col0, ctx.F * D, D))
    return None, None, g[:, 1 + D:], None, None, None, None, None


class ConcatFn(torch.autograd.Function):
  """tf.concat(parts, axis=1) (model/deepfm.py:75-83 and the towers' joins): ONE library launch forward; the backward
  hands every producer the column block of the incoming gradient as a VIEW - the fused backward kernels read strided
  gradients in place (no copies)."""

  @staticmethod
  def forward(ctx, *parts):
    ctx.widths = [int(t.shape[1]) for t in parts]
    be = hip()
    st = _bf16_step_state(be)
    ps = [t if t.stride(-1) == 1 else t.contiguous() for t in parts]
    return be.concat_cols(ps, bf16_state=st) if st is not None else be.concat_cols(ps)

  @staticmethod
  def backward(ctx, g):
    out, c = [], 0
    for w in ctx.widths:
      out.append(g[:, c:c + w])
      c += w
    return tuple(out)


def _bf16_step_state(be):
  """The Bf16Shadows of the model being run when its dense part is bf16 with producer-written operands, else None."""
  if not (getattr(be, 'bf16_nt', False) and getattr(be, 'bf16_epilogues', False)):
    return None
  from easyrec_amd.core import context
  stack = context._stack()
  ctx = stack[-1] if stack else None
  if ctx is None or getattr(ctx, 'dense_dtype', 'f32') != 'bf16':
    return None
  return getattr(ctx, 'bf16_state', None)


def concat_cols(parts):
  """Differentiable concat along dim 1 through the library (2-D fp32 tensors)."""
  parts = list(parts)
  if len(parts) == 1:
    return parts[0]
  if any(t.requires_grad for t in parts) and torch.is_grad_enabled():
    out = ConcatFn.apply(*parts)
    # a part that is the output of a fused dense + BatchNorm layer (DCN-v2's deep tower beside the cross stack): the
    # consumer's input-gradient contraction emits that layer's BatchNorm-backward sums from its column block
    c = 0
    for t in parts:
      src = bn_source_of(t)
      if src is not None and src.fused:
        tag_bn_cols(out, t, c)
        break
      c += int(t.shape[1])
    return out
  return hip().concat_cols([t if t.stride(-1) == 1 else t.contiguous() for t in parts])


class DotInteractionFn(torch.autograd.Function):
  """DLRM's pairwise feature interaction (reference model/dlrm.py:44-57) over a contiguous [B, F*D] block."""

  @staticmethod
  def forward(ctx, x, F, D, self_interaction):
    x = x if (x.stride(-1) == 1 and x.dim() == 2) else x.contiguous()
    out = hip().dot_interaction_fwd(x, F, D, self_interaction)
    ctx.save_for_backward(x)
    ctx.F, ctx.D, ctx.self_interaction = F, D, self_interaction
    return out

  @staticmethod
  def backward(ctx, g):
    x, = ctx.saved_tensors
    dx = hip().dot_interaction_bwd(x, g.contiguous(), ctx.F, ctx.D, ctx.self_interaction)
    return dx, None, None, None


class RowSumFn(torch.autograd.Function):
  """reference model/deepfm.py:62-63 (reduce_sum(wide, axis=1, keepdims=True))."""

  @staticmethod
  def forward(ctx, x, sink=None):
    ctx.n = x.shape[1]
    ctx.sink = sink
    return hip().rowsum_fwd(x, x.shape[1])

  @staticmethod
  def backward(ctx, g):
    if ctx.sink is not None:
      ctx.sink.defer(('rowsum', g if g.stride(-1) == 1 else g.contiguous(), 0, ctx.n))
      return None, None
    return hip().rowsum_bwd(g.contiguous(), ctx.n), None


class BNActFn(torch.autograd.Function):
  """bias + BatchNorm(train) + activation: reference layers/dnn.py:57-79.  `grad_bufs` = (bias.grad,
  gamma.grad, beta.grad) slices of the flat gradient buffer: the backward kernel accumulates into them
  directly (no autograd add kernels)."""

  @staticmethod
  def forward(ctx, x, bias, gamma, beta, moving_mean, moving_var, use_bn, eps, momentum, act, training,
              grad_bufs=None):
    mode = BN_NONE
    if use_bn:
      # training=False: the moving statistics normalise and stay as they are - evaluation, and the experts of the
      # reference's MMoE / DBMTL models in TRAINING too (model/mmoe.py:37-47 builds layers/mmoe.py MMOE without
      # is_training); gamma, beta, the bias and the input still receive gradients (er_bn_act_bwd, ER_BN_FROZEN)
      mode = BN_BATCH if training else BN_FROZEN
      assert training or (moving_mean is not None and moving_var is not None)
    y, mean, invstd = hip().bn_act_fwd(x, bias, gamma, beta, mode, eps, momentum, moving_mean, moving_var, act)
    ctx.save_for_backward(x, bias, gamma, y, mean, invstd)
    ctx.cfg = (mode, act)
    ctx.grad_bufs = grad_bufs
    ctx.beta = None if beta is None else beta.detach()  # (read by the backward's mask recomputation only)
    return y

  @staticmethod
  def backward(ctx, dy):
    x, bias, gamma, y, mean, invstd = ctx.saved_tensors
    use_bn, act = ctx.cfg
    into = None
    if ctx.grad_bufs is not None:
      bg, gg, betag = ctx.grad_bufs
      ok = (bias is None or bg is not None) and (gamma is None or (gg is not None and betag is not None))
      if ok:
        into = (bg, gg, betag)
    dx, dbias, dgamma, dbeta = hip().bn_act_bwd(x, bias, gamma, y, mean, invstd, dy.contiguous(), use_bn, act,
                                                bias is not None, gamma is not None, into=into, beta=ctx.beta)
    return dx, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None


class DiceFn(torch.autograd.Function):
  """reference layers/keras/activation.py:47-70 / utils/activation.py:14-44."""

  @staticmethod
  def forward(ctx, x, alpha, moving_mean, moving_var, eps, momentum, alpha_grad=None):
    y, mean, invstd = hip().dice_fwd(x, alpha, eps, momentum, moving_mean, moving_var)
    ctx.save_for_backward(x, alpha, mean, invstd)
    ctx.alpha_grad = alpha_grad
    return y

  @staticmethod
  def backward(ctx, dy):
    x, alpha, mean, invstd = ctx.saved_tensors
    dx, dalpha = hip().dice_bwd(x, alpha, mean, invstd, dy.contiguous())
    if ctx.alpha_grad is not None:
      ctx.alpha_grad.add_(dalpha)
      dalpha = None
    return dx, dalpha, None, None, None, None, None


class CrossV1Fn(torch.autograd.Function):
  """reference model/dcn.py:32-45, all layers in one launch.  `grad_bufs` = ([w_l.grad], [b_l.grad]): the
  per-layer slices of the flat gradient buffer (accumulated into directly; no AccumulateGrad nodes)."""

  @staticmethod
  def forward(ctx, x0, w, b, grad_bufs=None):
    out, dots = hip().cross_v1_fwd(x0.contiguous(), w, b)
    ctx.save_for_backward(x0, w, b, dots)
    ctx.grad_bufs = grad_bufs
    return out

  @staticmethod
  def backward(ctx, dout):
    x0, w, b, dots = ctx.saved_tensors
    dx0, dw, db = hip().cross_v1_bwd(x0.contiguous(), w, b, dots, dout.contiguous())
    if ctx.grad_bufs is not None:
      wg, bg = ctx.grad_bufs
      for i in range(len(wg)):
        wg[i].add_(dw[i])
        bg[i].add_(db[i])
      dw = db = None
    return dx0, dw, db, None


class CrossV2EpilogueFn(torch.autograd.Function):
  """reference layers/keras/interaction.py:276-286: x0 * (u + bias + diag*x) + x.  The backward ADDS its gradients of x0
  and x into the buffers their other consumers use (the embedding group's gradient buffer when x0 is a group output, else
  the tensors' grad_slot): a stack of L cross layers sums nothing through autograd."""

  @staticmethod
  def forward(ctx, x0, x, u, bias, diag_scale, bias_grad=None, sink=None):
    x0c, xc = x0.contiguous(), x.contiguous()
    out = hip().cross_v2_fwd(x0c, xc, u.contiguous(), bias, diag_scale)
    ctx.save_for_backward(x0c, xc, u, bias)
    ctx.diag = diag_scale
    ctx.bias_grad = bias_grad
    ctx.same = x0c.data_ptr() == xc.data_ptr()
    ctx.sink = sink if (sink is not None and x0c is x0) else None
    # (a contiguous copy is another tensor: its consumers do not share a slot)
    ctx.slots = grad_slots_of_step() if (x0c is x0 and xc is x) else None
    return out

  @staticmethod
  def backward(ctx, dout):
    be = hip()
    x0, x, u, bias = ctx.saved_tensors
    d = x0.shape[1]
    ret0 = retx = None
    if not hasattr(be, 'cross_v2_bwd_acc') or ctx.slots is None:
      dx0, dx, du = be.cross_v2_bwd(x0, x, u.contiguous(), bias, ctx.diag, dout.contiguous())
      ret0, retx = dx0, dx
    else:
      # destination of d/dx0: the group's gradient buffer (a first deposit must cover it: here it does), else x0's slot
      sink = ctx.sink
      if sink is not None and sink.covers(0, d):
        t0, acc0 = sink.target(0, d)
      else:
        sink = None
        t0, acc0, first0 = grad_slot(ctx.slots, x0)
        ret0 = t0 if first0 else None
      tx = accx = None
      if not ctx.same:
        tx, accx, firstx = grad_slot(ctx.slots, x)
        retx = tx if firstx else None
      dg = dout if (dout.dim() == 2 and dout.stride(1) == 1) else dout.contiguous()  # (a column block of tf.concat's gradient: in place)
      du = be.cross_v2_bwd_acc(x0, x, u.contiguous(), bias, ctx.diag, dg, t0, acc0, tx, accx)
      if sink is not None:
        sink.done()
    dbias = None
    if bias is not None:
      if ctx.bias_grad is not None:  # the column sums straight into the bias' slice of the flat gradient buffer
        be.colsum(du, out=ctx.bias_grad, accumulate=True)
      else:
        dbias = be.colsum(du)
    return ret0, retx, du, dbias, None, None, None


class CrossSrc(object):
  """What the input-gradient contraction of the cross layer ABOVE needs in order to run THIS layer's elementwise backward in
  its epilogue (HipBackend.cross_dgrad_fused(prev=...)), and what it leaves behind for this layer's backward."""
  __slots__ = ('x0', 'x', 'u', 'bias', 'diag', 'bias_grad', 'out', 'sink', 'slots', 'wsink', 'done', 'dout_ptr', 'du', 'ret0')

  def __init__(self, x0, x, u, bias, diag, bias_grad, out, sink, slots, wsink):
    self.x0, self.x, self.u, self.bias, self.diag, self.bias_grad, self.out = x0, x, u, bias, diag, bias_grad, out
    self.sink, self.slots, self.wsink = sink, slots, wsink
    self.done, self.dout_ptr, self.du, self.ret0 = False, 0, None, None

  def dx0_target(self):
    """(tensor, accumulate, what to hand autograd for x0 - the slot tensor on its first use - or None, the sink to mark)"""
    d = self.x0.shape[1]
    sink = self.sink
    if sink is not None and sink.covers(0, d):
      t0, acc0 = sink.target(0, d)
      return t0, acc0, None, sink
    t0, acc0, first0 = grad_slot(self.slots, self.x0)
    return t0, acc0, (t0 if first0 else None), None


_cross_tls = threading.local()


def take_last_cross_source():
  src, _cross_tls.last = getattr(_cross_tls, 'last', None), None
  return src


def cross_source_of(x):
  """The CrossSrc of x if x IS the untouched output of a fused cross layer."""
  src = getattr(x, '_er_cross_src', None)
  if src is None or x.dim() != 2 or x.data_ptr() != src.out.data_ptr() or x.shape != src.out.shape or x.stride() != src.out.stride():
    return None
  return src


class CrossLayerFn(torch.autograd.Function):
  """One full-rank DCN-v2 cross layer x_{l+1} = x0 * (x_l . W + b + diag * x_l) + x_l (reference layers/keras/interaction.py:
  249-286) as ONE launch forward - the contraction with the elementwise part in its epilogue - and, in a stack, ONE launch
  per layer backward: the input-gradient contraction du_l . W^T adds dout_l in its epilogue (the whole gradient of x_{l-1})
  and runs the elementwise backward of the layer below on it (du_{l-1}, d/dx0, the bias gradient's per-tile column sums);
  only the top layer of a stack, whose dout comes from outside, needs its own elementwise launch.  Weight gradients join the
  step's grouped launch; the bias gradients are finished by one launch for all layers (HipBackend.queue_colsum).  d/dx0 goes
  into the embedding group's gradient buffer (sink) or x0's gradient slot."""

  @staticmethod
  def forward(ctx, x0, x, w, bias, diag, w_grad, b_grad, sink, bf16, prev):
    be = hip()
    out, u = be.cross_fwd_fused(x0, x, w, bias, diag, bf16)
    ctx.save_for_backward(x0, x, u, w, bias)
    ctx.diag, ctx.bf16, ctx.w_grad = diag, bf16, w_grad
    ctx.same = x0.data_ptr() == x.data_ptr()
    ctx.prev = prev
    ctx.src = CrossSrc(x0, x, u, bias, diag, b_grad, out, sink, grad_slots_of_step(), be.wgrad_sink())
    _cross_tls.last = ctx.src
    return out

  @staticmethod
  def backward(ctx, dout):
    be = hip()
    x0, x, u, w, bias = ctx.saved_tensors
    src, prev = ctx.src, ctx.prev
    dg = dout if (dout.dim() == 2 and dout.stride(1) == 1) else dout.contiguous()
    ret0 = None
    # 1. this layer's elementwise backward - already done by the contraction of the layer above?
    if src.done and dg.data_ptr() == src.dout_ptr:
      du = src.du
    else:
      t0, acc0, r0, sk = src.dx0_target()
      du, partial = be.cross_bwd_top(x0, x, u, bias, ctx.diag, dg, t0, acc0, ctx.bf16, w)
      if sk is not None:
        sk.done()
      ret0 = r0
      if bias is not None:
        be.queue_colsum(src.wsink, partial, src.bias_grad, bias.numel())
    src.done, src.du = False, None
    # 2. the input-gradient contraction; its epilogue completes the gradient of x_{l-1} and runs the layer below
    retx = None
    if ctx.same:  # x_{l-1} is x0: the gradient joins d/dx0
      t0, acc0, r0, sk = src.dx0_target()
      assert acc0, 'the elementwise backward deposits into d/dx0 first'
      res = be.cross_dgrad_fused(du, w, dg, ctx.diag, ctx.bf16, t0, True)
      assert res is not False
      if sk is not None:
        sk.done()
      ret0 = ret0 if ret0 is not None else r0
    else:
      retx = torch.empty_like(x)
      pv = None
      if prev is not None:
        p0, pacc0, pr0, psk = prev.dx0_target()
        pv = dict(x0=prev.x0, u=prev.u, bias=prev.bias, xl=prev.x, dx0=p0, acc0=pacc0)
      res = be.cross_dgrad_fused(du, w, dg, ctx.diag, ctx.bf16, retx, False, prev=pv)
      assert res is not False
      if prev is not None:
        du_prev, partial_prev = res
        if psk is not None:
          psk.done()
        ret0 = ret0 if ret0 is not None else pr0
        if prev.bias is not None:
          be.queue_colsum(prev.wsink, partial_prev, prev.bias_grad, prev.bias.numel())
        prev.done, prev.dout_ptr, prev.du = True, retx.data_ptr(), du_prev
    # 3. the weight gradient x_{l-1}^T . du joins the step's grouped launch
    dw = None
    if ctx.needs_input_grad[2]:
      dw = _wgrad(be, x, du, ctx.w_grad, ctx.bf16, src.wsink)
    return ret0, retx, dw, None, None, None, None, None, None, None


class CINFn(torch.autograd.Function):
  """xDeepFM's compressed interaction network, all layers (reference layers/keras/interaction.py:370-409).
  x0 [B, H0, D]; kernels[k] [H_k+1, H_k, H0], biases[k] [H_k+1]; returns [B, sum_k H_k+1].  The parameter gradients are
  accumulated into kernel_grads / bias_grads (the flat gradient buffer's views)."""

  @staticmethod
  def forward(ctx, x0, n_layers, *params):
    be = hip()
    kernels, biases = params[:n_layers], params[n_layers:2 * n_layers]
    ctx.kernel_grads, ctx.bias_grads = params[2 * n_layers:3 * n_layers], params[3 * n_layers:4 * n_layers]
    x0 = x0.contiguous()
    B, H0, D = x0.shape
    sizes = [int(w.shape[0]) for w in kernels]
    out = torch.empty(B, sum(sizes), dtype=torch.float32, device=x0.device)
    xi, strides, H = x0, (H0 * D, D, 1), H0
    zs, fms = [], []
    col = 0
    for w, b in zip(kernels, biases):
      N = int(w.shape[0])
      assert w.shape == (N, H, H0), 'cin kernel %s for a [%d x %d] interaction' % (tuple(w.shape), H, H0)
      z = torch.empty(B * D, H * H0, dtype=torch.float32, device=x0.device)
      be.cin_outer_fwd(xi, strides, H, x0, z)
      fm = be.gemm(GEMM_NT, z, w.detach().reshape(N, H * H0))  # [B * D, N] = x_{k+1} as [B, D, N]
      be.cin_act_pool_fwd(fm, b.detach(), B, D, out, col)
      zs.append(z)
      fms.append(fm)
      xi, strides, H = fm, (D * N, 1, N), N
      col += N
    ctx.save_for_backward(x0, *kernels, *zs, *fms)
    ctx.n_layers, ctx.sizes = n_layers, sizes
    return out

  @staticmethod
  def backward(ctx, dout):
    be = hip()
    n = ctx.n_layers
    saved = ctx.saved_tensors
    x0, kernels, zs, fms = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:1 + 3 * n]
    B, H0, D = x0.shape
    dout = dout.contiguous()
    dx0 = torch.zeros_like(x0)
    cols = [sum(ctx.sizes[:k]) for k in range(n)]
    dnext = None  # gradient of x_{k+1} from the layer above ([B * D, N])
    for k in reversed(range(n)):
      w, z, fm = kernels[k].detach(), zs[k], fms[k]
      N = ctx.sizes[k]
      H = H0 if k == 0 else ctx.sizes[k - 1]
      dc = torch.empty_like(fm)
      be.cin_act_pool_bwd(fm, dout, cols[k], dnext, B, D, dc)
      if ctx.bias_grads[k] is not None:
        be.colsum(dc, out=ctx.bias_grads[k], accumulate=True)
      be.gemm(GEMM_TN, dc, z, out=ctx.kernel_grads[k].view(N, H * H0), accumulate=True)  # dW = dc^T . z
      dz = be.gemm(GEMM_NN, dc, w.reshape(N, H * H0))
      if k == 0:
        be.cin_outer_bwd(dz, x0, (H0 * D, D, 1), H, x0, dx0, True, dx0)
        dnext = None
      else:
        prev = fms[k - 1]
        dprev = torch.empty_like(prev)
        be.cin_outer_bwd(dz, prev, (D * H, 1, H), H, x0, dprev, False, dx0)
        dnext = dprev
    return (dx0, None) + (None,) * (4 * n)


class DINConcatFn(torch.autograd.Function):
  """reference model/multi_tower_din.py:69-75: [q, h, q-h, q*h]."""

  @staticmethod
  def forward(ctx, q, h):
    ctx.save_for_backward(q, h)
    # (the history is also read by the pooling: both backward kernels write ONE gradient buffer - kernels.grad_slot)
    ctx.slots = grad_slots_of_step() if h.is_contiguous() else None
    return hip().din_concat_fwd(q.contiguous(), h.contiguous())

  @staticmethod
  def backward(ctx, dout):
    q, h = ctx.saved_tensors
    if ctx.slots is None:
      return hip().din_concat_bwd(q.contiguous(), h.contiguous(), dout.contiguous())
    buf, acc, first = grad_slot(ctx.slots, h)
    dq, _ = hip().din_concat_bwd(q.contiguous(), h, dout.contiguous(), dh=buf, acc_h=acc)
    return dq, (buf if first else None)


class DINFirstLayerFn(torch.autograd.Function):
  """dense (+ bias) -> BatchNorm(train) -> activation over DIN's attention input [q, h, q - h, q * h] ([B, L, 4E], reference
  model/multi_tower_din.py:62-80 feeding layers/dnn.py:57-79) WITHOUT building it, forward or backward: the contraction
  generates the block from (q, h) while staging (er_din_gemm_fwd), the weight gradient likewise (er_din_gemm_wgrad), and the
  input-gradient contraction reduces the block's gradient to dq / dh in its epilogue (er_din_gemm_dgrad).  Same variables
  and arithmetic as LinearBNActFn over DINConcatFn's output; the history's gradient lands in its gradient slot."""

  @staticmethod
  def forward(ctx, q, h, w, b, gamma, beta, moving_mean, moving_var, eps, momentum, act, grad_bufs, defer_apply=False):
    be = hip()
    B, L, E = h.shape
    N = w.shape[1]
    chunks = be.gemm_row_tiles(B * L)
    stats = torch.empty(chunks * N * 3, dtype=torch.float32, device=h.device)
    z = be.din_gemm_fwd(q, h, w, b, col_stats=stats)
    fused = getattr(be, 'fused_bn_bwd', False)
    pend = None
    if defer_apply and fused and getattr(be, 'bn_in_staging', False):
      # the next layer's contraction applies this BatchNorm while it stages its A tiles (LinearBNActFn, HipBackend.gemm_bn_a)
      y = torch.empty_like(z)
      mean = torch.empty(N, dtype=torch.float32, device=z.device)
      invstd = torch.empty(N, dtype=torch.float32, device=z.device)
      pend = dict(z=z, stats=stats, chunks=chunks, gamma=gamma.detach(), beta=beta.detach(), eps=eps, momentum=momentum,
                  moving_mean=moving_mean, moving_var=moving_var, act=act, y=y, mean=mean, invstd=invstd)
    else:
      y, mean, invstd = be.bn_apply_from_stats(z, None, stats, chunks, gamma, beta, eps, momentum, moving_mean, moving_var, act)
    ctx.save_for_backward(q, h, w, gamma, beta, z, y, mean, invstd)
    ctx.act, ctx.grad_bufs = act, grad_bufs
    ctx.slots = grad_slots_of_step()
    gb = None if grad_bufs is None else (grad_bufs[1], grad_bufs[2])
    ctx.own = BnSource(z, None, y, mean, invstd, act, gamma, gb, beta=beta, fused=fused) if fused else None
    if pend is not None:
      ctx.own.pending = pend
    _bn_tls.last = ctx.own
    return y

  @staticmethod
  def backward(ctx, dy):
    be = hip()
    q, h, w, gamma, beta, z, y, mean, invstd = ctx.saved_tensors
    assert ctx.own is None or ctx.own.pending is None, 'a deferred BatchNorm apply was never run (kernels.finish_pending_bn)'
    wg, gg, betag = ctx.grad_bufs if ctx.grad_bufs is not None else (None, None, None)
    direct = gg is not None and betag is not None
    dyc = dy if (dy.dim() == 2 and dy.stride(1) == 1) else dy.contiguous()
    own, partial = ctx.own, None
    if own is not None:
      if own.partial is not None and dyc.data_ptr() == own.dx_ptr:
        partial = own.partial
      own.partial = None
    dz, _, dgamma, dbeta = be.bn_act_bwd(z, None, gamma, y, mean, invstd, dyc, 1, ctx.act, False, True,
                                         into=(None, gg, betag) if direct else None, partial=partial, beta=beta)
    dq = dh = dw = None
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
      if ctx.slots is not None:
        buf, acc, first = grad_slot(ctx.slots, h)
        dq, _ = be.din_gemm_dgrad(dz, w, q, h, dh=buf, acc_h=acc)
        dh = buf if first else None
      else:
        dq, dh = be.din_gemm_dgrad(dz, w, q, h)
    if ctx.needs_input_grad[2]:
      if wg is not None:
        be.din_gemm_wgrad(q, h, dz, wg, accumulate=True)
      else:
        dw = be.din_gemm_wgrad(q, h, dz, torch.empty_like(w), accumulate=False)
    return dq, dh, dw, None, dgamma, dbeta, None, None, None, None, None, None, None


class DINPoolFn(torch.autograd.Function):
  """reference model/multi_tower_din.py:86-95: mask, softmax over time, scores @ hist."""

  @staticmethod
  def forward(ctx, scores, hist, seq_len, scale):
    out, probs = hip().din_pool_fwd(scores.contiguous(), hist.contiguous(), seq_len, scale)
    ctx.save_for_backward(probs, hist, seq_len)
    ctx.scale = scale
    ctx.slots = grad_slots_of_step() if hist.is_contiguous() else None
    return out

  @staticmethod
  def backward(ctx, dout):
    probs, hist, seq_len = ctx.saved_tensors
    if ctx.slots is None:
      dscores, dhist = hip().din_pool_bwd(probs, hist.contiguous(), seq_len, dout.contiguous(), ctx.scale)
      return dscores, dhist, None, None
    buf, acc, first = grad_slot(ctx.slots, hist)
    dscores, _ = hip().din_pool_bwd(probs, hist, seq_len, dout.contiguous(), ctx.scale, dhist=buf, acc_h=acc)
    return dscores, (buf if first else None), None, None


class MMoEMixManyFn(torch.autograd.Function):
  """MMoEMixFn over E + T separate tensors: apply(E, T, expert_0 .. expert_{E-1} [B, H], gate_logits_0 .. [B, E]) -> T
  mixtures [B, H].  The inputs are gathered into the kernels' [E, B, H] / [T, B, E] operands by ONE copy launch
  (er_copy_multi) and so are the T output gradients on the way back; the input gradients are views of the backward
  kernels' outputs (torch.stack / select through autograd was ~9 launches of zero-fill and slice copies per step)."""

  @staticmethod
  def forward(ctx, E, T, *args):
    be = hip()
    outs, gate_logits = args[:E], args[E:E + T]
    B, H = outs[0].shape
    dev = outs[0].device
    experts = torch.empty(E, B, H, dtype=torch.float32, device=dev)
    logits = torch.empty(T, B, E, dtype=torch.float32, device=dev)
    be.copy_multi([(experts[e], outs[e] if outs[e].is_contiguous() else outs[e].contiguous()) for e in range(E)] +
                  [(logits[t], gate_logits[t] if gate_logits[t].is_contiguous() else gate_logits[t].contiguous())
                   for t in range(T)])
    out, gates = be.mmoe_mix_fwd(experts, logits)
    ctx.save_for_backward(experts, gates)
    ctx.E, ctx.T = E, T
    return tuple(out[t] for t in range(T))

  @staticmethod
  def backward(ctx, *douts):
    be = hip()
    experts, gates = ctx.saved_tensors
    E, T = ctx.E, ctx.T
    dout = torch.empty(T, experts.shape[1], experts.shape[2], dtype=torch.float32, device=experts.device)
    be.copy_multi([(dout[t], douts[t] if douts[t].is_contiguous() else douts[t].contiguous()) for t in range(T)])
    dexperts, dlogits = be.mmoe_mix_bwd(experts, gates, dout)
    return (None, None) + tuple(dexperts[e] for e in range(E)) + tuple(dlogits[t] for t in range(T))


class MMoEMixFn(torch.autograd.Function):
  """reference layers/mmoe.py:73-82: softmax gates, weighted sum of experts, all tasks at once."""

  @staticmethod
  def forward(ctx, experts, gate_logits):
    out, gates = hip().mmoe_mix_fwd(experts.contiguous(), gate_logits.contiguous())
    ctx.save_for_backward(experts, gates)
    return out

  @staticmethod
  def backward(ctx, dout):
    experts, gates = ctx.saved_tensors
    return hip().mmoe_mix_bwd(experts.contiguous(), gates, dout.contiguous())

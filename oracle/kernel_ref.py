"""ORACLE - TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by easyrec_amd/).

Per-entry-point CPU restatement of the arithmetic behind include/easyrec_hip.h, written with plain
numpy / torch-CPU ops in the order the reference's TF ops run.  `RefBackend` has the same
tensor-level methods as `easyrec_amd.kernels.HipBackend`, so that
  * `-m gpu` tests compare every HIP kernel with the matching method on the same seeded inputs;
  * `-m "not gpu"` tests can drive the whole host logic (plan building, models, estimator) on CPU
    by monkeypatching the backend - that checks the host code, not the kernels.

Reference call sites restated (all under /root/reference/easy_rec/python):
  lookup/combiner  compat/embedding_ops.py:37-162 (safe_embedding_lookup_sparse),
                   compat/feature_column/feature_column.py:199-244 (sum / mean / sqrtn formulas)
  sparse Adam      compat/adam_s.py:185-213 and tf.train.AdamOptimizer._apply_sparse_shared
  dense Adam       training_ops.apply_adam (compat/adam_s.py:148-165 call site)
  FM               layers/fm.py:20-26          wide sum  model/deepfm.py:62-63
  cross v1 / v2    model/dcn.py:32-45 ; layers/keras/interaction.py:276-286
  DIN              model/multi_tower_din.py:62-97
  DNN / BN / Dice  layers/dnn.py:57-79 ; utils/activation.py:14-44
  loss             builders/loss_builder.py:35-39 ; compat/regularizers.py:76-108
  MMoE             layers/mmoe.py:62-83
"""
import numpy as np
import torch

from oracle import hashing

OPT_SGD, OPT_ADAM, OPT_LAZY_ADAM, OPT_ADAGRAD = 0, 1, 2, 3
ACT_NONE, ACT_RELU = 0, 1
BN_FROZEN = 2  # use_bn value: normalise with the moving statistics (include/easyrec_hip.h ER_BN_FROZEN)
(HYPER_LR, HYPER_LR_T, HYPER_BETA1, HYPER_BETA2, HYPER_OMB1, HYPER_OMB2, HYPER_EPS, HYPER_GSCALE) = range(8)
HYPER_CLIP = 8  # er_opt_hyper.clip_scale (0 = no clipping)


def _clip_of(h):
  return F32(h[HYPER_CLIP]) if (len(h) > HYPER_CLIP and h[HYPER_CLIP] != 0) else None

F32 = np.float32


# ---------------------------------------------------------------------------------------------
# embedding lookup
# ---------------------------------------------------------------------------------------------
def lookup_rows(table, ids, offsets, weights, combiner, n_rows):
  """safe_embedding_lookup_sparse for one column.  numpy fp32, sequential in id order.

  table [rows, dim]; ids int64 (dense mode: [n_rows]; ragged: [nnz] with offsets [n_rows+1]).
  Pruning: id < 0 (or >= rows) dropped; weight <= 0 dropped unless combiner == sum
  (compat/embedding_ops.py:95-110).  Empty rows -> zeros (:147-155).
  """
  rows, dim = table.shape
  out = np.zeros((n_rows, dim), dtype=np.float32)
  for r in range(n_rows):
    kb, ke = (r, r + 1) if offsets is None else (int(offsets[r]), int(offsets[r + 1]))
    acc = np.zeros(dim, dtype=np.float32)
    wsum, w2sum = F32(0), F32(0)
    for k in range(kb, ke):
      i = int(ids[k])
      if i < 0 or i >= rows:
        continue
      if weights is not None:
        w = F32(weights[k])
        if combiner != 0 and not (w > 0):
          continue
        acc = acc + table[i] * w  # embeddings *= weights ; segment_sum
        wsum = F32(wsum + w)
        w2sum = F32(w2sum + w * w)
      else:
        acc = acc + table[i]
        wsum = F32(wsum + F32(1))
        w2sum = F32(w2sum + F32(1))
    if combiner != 0 and wsum != 0:
      den = wsum if combiner == 1 else np.sqrt(w2sum, dtype=np.float32)
      acc = acc / den
    out[r] = acc
  return out


def lookup_rows_fast(table, ids, offsets, weights, combiner, n_rows):
  """Vectorised equivalent of `lookup_rows` for dense-mode lookups (one id per row)."""
  if offsets is not None:
    return lookup_rows(table, ids, offsets, weights, combiner, n_rows)
  rows, dim = table.shape
  ids = np.asarray(ids[:n_rows])
  ok = (ids >= 0) & (ids < rows)
  w = None
  if weights is not None:
    w = np.asarray(weights[:n_rows], dtype=np.float32)
    if combiner != 0:
      ok = ok & (w > 0)
  safe = np.where(ok, ids, 0)
  e = table[safe].astype(np.float32)
  if w is not None:
    e = e * w[:, None]
    if combiner == 1:
      e = e / np.where(ok, w, F32(1))[:, None]
    elif combiner == 2:
      e = e / np.where(ok, np.sqrt(w * w, dtype=np.float32), F32(1))[:, None]
  e[~ok] = 0
  return e.astype(np.float32)


def lookup_entries(spec):
  """All (key, out_row, scale) entries of a lookup, in entry order (backward bookkeeping)."""
  ids = spec.ids.cpu().numpy()
  offsets = None if spec.offsets is None else spec.offsets.cpu().numpy()
  weights = None if spec.weights is None else spec.weights.cpu().numpy()
  ents = []
  for r in range(spec.n_rows):
    kb, ke = (r, r + 1) if offsets is None else (int(offsets[r]), int(offsets[r + 1]))
    valid = []
    for k in range(kb, ke):
      i = int(ids[k])
      w = F32(1) if weights is None else F32(weights[k])
      if i < 0 or i >= spec.rows:
        continue
      if weights is not None and spec.combiner != 0 and not (w > 0):
        continue
      valid.append((i, w))
    den = F32(1)
    if spec.combiner != 0 and valid:
      wsum = F32(0)
      w2 = F32(0)
      for _, w in valid:
        wsum = F32(wsum + w)
        w2 = F32(w2 + w * w)
      den = wsum if spec.combiner == 1 else np.sqrt(w2, dtype=np.float32)
    for i, w in valid:
      ents.append((spec.key_base + i, r, F32(w / den)))
  return ents


def adam_row(var, m, v, g, h):
  """Row arithmetic of the sparse Adam apply (fp32, op by op; compat/adam_s.py:193-213)."""
  g = g.astype(np.float32)
  m_t = m * F32(h[HYPER_BETA1]) + g * F32(h[HYPER_OMB1])
  v_t = v * F32(h[HYPER_BETA2]) + (g * g) * F32(h[HYPER_OMB2])
  var_t = var - (F32(h[HYPER_LR_T]) * m_t) / (np.sqrt(v_t, dtype=np.float32) + F32(h[HYPER_EPS]))
  return var_t.astype(np.float32), m_t.astype(np.float32), v_t.astype(np.float32)


def sparse_grads(specs, dim):
  """De-duplicated gradient of a table group: {global_row: summed grad}, occurrences summed in
  ascending entry order (TF: unsorted_segment_sum over positions)."""
  acc = {}
  order = []
  for s in specs:
    dout = s.out.detach().cpu().numpy()
    for key, r, scale in lookup_entries(s):
      g = dout[r, s.out_col:s.out_col + dim].astype(np.float32) * scale
      if key in acc:
        acc[key] = (acc[key] + g).astype(np.float32)
      else:
        acc[key] = g.astype(np.float32)
        order.append(key)
  return acc


def apply_sparse(var, m, v, grads, opt_kind, h):
  """var/m/v numpy [total_rows, dim] updated in place."""
  gs = F32(h[HYPER_GSCALE])
  clip = _clip_of(h)
  touched = set()
  for key, g in grads.items():
    g = (g * gs).astype(np.float32)
    if clip is not None:
      g = (g * clip).astype(np.float32)
    touched.add(key)
    if opt_kind in (OPT_ADAM, OPT_LAZY_ADAM):
      var[key], m[key], v[key] = adam_row(var[key], m[key], v[key], g, h)
    elif opt_kind == OPT_ADAGRAD:
      v[key] = v[key] + g * g
      var[key] = var[key] - (g * F32(h[HYPER_LR])) / np.sqrt(v[key], dtype=np.float32)
    else:
      var[key] = var[key] - F32(h[HYPER_LR]) * g
  if opt_kind == OPT_ADAM:
    # tf.train.AdamOptimizer._apply_sparse_shared: m, v of EVERY row decay, every row moves
    mask = np.ones(var.shape[0], dtype=bool)
    if touched:
      mask[np.fromiter(touched, dtype=np.int64)] = False
    m[mask] = m[mask] * F32(h[HYPER_BETA1])
    v[mask] = v[mask] * F32(h[HYPER_BETA2])
    var[mask] = var[mask] - (F32(h[HYPER_LR_T]) * m[mask]) / (np.sqrt(v[mask], dtype=np.float32) +
                                                            F32(h[HYPER_EPS]))


def dense_opt(w, m, v, grad, l2coef, opt_kind, h):
  """training_ops.apply_adam: m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps)."""
  g = grad.astype(np.float32) * F32(h[HYPER_GSCALE])
  if l2coef is not None:
    g = np.where(l2coef != 0, g + l2coef * w, g).astype(np.float32)
  if _clip_of(h) is not None:
    g = (g * _clip_of(h)).astype(np.float32)
  if opt_kind in (OPT_ADAM, OPT_LAZY_ADAM):
    m[:] = m + (g - m) * F32(h[HYPER_OMB1])
    v[:] = v + (g * g - v) * F32(h[HYPER_OMB2])
    w[:] = w - (m * F32(h[HYPER_LR_T])) / (np.sqrt(v, dtype=np.float32) + F32(h[HYPER_EPS]))
  elif opt_kind == OPT_ADAGRAD:
    v[:] = v + g * g
    w[:] = w - (g * F32(h[HYPER_LR])) / np.sqrt(v, dtype=np.float32)
  else:
    w[:] = w - F32(h[HYPER_LR]) * g


# ---------------------------------------------------------------------------------------------
# backend with the HipBackend method surface (CPU tensors)
# ---------------------------------------------------------------------------------------------
class _NoWgradSink(object):
  """The queue object LinearFn asks its backend for; this backend never defers a weight gradient."""
  active = False
  queue = ()

  def put(self, x, dy, out, bf16, at=None):
    return False


class _Spec(object):
  """The attributes of a lookup this file reads (duck-typed stand-in for the product's LookupSpec)."""

  def __init__(self, **kw):
    self.__dict__.update(kw)


def _same_lookup_keys(group, leader):
  """Do two table groups see the same keys every step: same id / offset buffers, table geometry and routing?"""
  if group is leader or leader.get('sort_leader') is not None:
    return False
  a, b = group['specs'], leader['specs']
  if len(a) != len(b) or group.get('n_active', -1) != leader.get('n_active', -1):
    return False
  if any(group.get(k) != leader.get(k) for k in ('world', 'shard_stride', 'local_base')):
    return False
  addr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
  return all((addr(x.ids), addr(x.offsets), x.rows, x.key_base, x.n_rows, x.max_nnz) ==
             (addr(y.ids), addr(y.offsets), y.rows, y.key_base, y.n_rows, y.max_nnz) for x, y in zip(a, b))


class RefBackend(object):
  name = 'oracle'

  def reserve_scratch(self, floats):
    pass

  @staticmethod
  def require_device():
    pass

  def device_info(self):
    return {'cu_count': 0, 'wave_size': 0, 'arch': 'cpu-oracle'}

  # -- hashing
  def hash_bucket_fast_host(self, bytes_np, offsets_np, n_per_col, num_buckets, drop_empty):
    return hashing.hash_bucket_fast(bytes_np, offsets_np, n_per_col, num_buckets, drop_empty)

  def decode_csv_host(self, text, sep, kinds, max_rows, threads=0, out=None):
    """Line by line in Python (what tf.decode_csv does per record, input/csv_input.py:33-76)."""
    raw = np.ascontiguousarray(text, dtype=np.uint8).tobytes()
    sep_b = sep.encode('utf-8') if isinstance(sep, str) else bytes(sep)
    F = len(kinds)
    ints, flts = np.zeros((F, max_rows), dtype=np.int64), np.zeros((F, max_rows), dtype=np.float64)
    empty, begin = np.zeros((F, max_rows), dtype=np.uint8), np.zeros((F, max_rows), dtype=np.int64)
    length = np.zeros((F, max_rows), dtype=np.int32)
    pos = row = 0
    while pos < len(raw) and row < max_rows:
      nl = raw.find(b'\n', pos)
      if nl < 0:
        break
      eol = nl - 1 if nl > pos and raw[nl - 1:nl] == b'\r' else nl
      if eol > pos:
        cells = raw[pos:eol].split(sep_b)
        assert len(cells) == F, 'line %d has %d fields, expected %d' % (row, len(cells), F)
        b = pos
        for f, c in enumerate(cells):
          empty[f, row], begin[f, row], length[f, row] = len(c) == 0, b, len(c)
          if c and kinds[f] == 1:
            ints[f, row] = int(c)
            flts[f, row] = float(int(c))
          elif c and kinds[f] == 2:
            flts[f, row] = float(c)
          b += len(c) + 1
        row += 1
      pos = nl + 1
    return row, pos, ints, flts, empty, begin, length

  def pack_cells_host(self, text, begin, length):
    raw = np.ascontiguousarray(text, dtype=np.uint8).tobytes()
    cells = [raw[int(b):int(b) + int(n)] for b, n in zip(begin, length)]
    offsets = np.zeros(len(cells) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cells], out=offsets[1:])
    return np.frombuffer(b''.join(cells), dtype=np.uint8) if offsets[-1] else np.zeros(0, dtype=np.uint8), offsets

  def pack_int_decimal_host(self, values):
    enc = [str(int(v)).encode('ascii') for v in np.asarray(values).reshape(-1)]
    offsets = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=offsets[1:])
    return np.frombuffer(b''.join(enc), dtype=np.uint8) if enc else np.zeros(0, dtype=np.uint8), offsets

  def sparse_cross_hashed_host(self, bytes_np, offsets_np, n_rows, n_cols, num_buckets, hash_key=None):
    key = hashing.DEFAULT_CROSS_HASH_KEY if hash_key is None else hash_key
    return hashing.sparse_cross_hashed_columns(bytes_np, offsets_np, n_rows, n_cols, num_buckets, key)

  def hash_bucket_fast(self, bytes_t, offsets_t, n_per_col, num_buckets_t, drop_empty, out=None):
    r = hashing.hash_bucket_fast(bytes_t.cpu().numpy(), offsets_t.cpu().numpy(), n_per_col,
                                 num_buckets_t.cpu().numpy().astype(np.uint64), drop_empty)
    r = torch.from_numpy(r)
    if out is None:
      return r
    out.view(-1)[:r.numel()].copy_(r)
    return out

  def hash_bucket_fast_int64(self, values_t, n_per_col, num_buckets_t, out=None):
    vals = values_t.cpu().numpy().reshape(-1)
    strs = [str(int(x)).encode() for x in vals]
    data = np.frombuffer(b''.join(strs), dtype=np.uint8)
    offs = np.cumsum([0] + [len(s) for s in strs]).astype(np.int64)
    r = torch.from_numpy(
        hashing.hash_bucket_fast(data, offs, n_per_col, num_buckets_t.cpu().numpy().astype(np.uint64), False))
    if out is None:
      return r
    out.view(-1).copy_(r)
    return out

  # -- embeddings
  def emb_plan_create(self, specs):
    nblk = 0
    for s in specs:
      V = 4 if s.dim % 4 == 0 else 1
      G = 1
      while G < s.dim // V:
        G <<= 1
      nblk += (max(s.n_rows, 1) + 256 // G - 1) // (256 // G)
    return {'specs': list(specs), 'num_blocks': nblk}

  def emb_plan_destroy(self, plan):
    pass

  def emb_fwd(self, plan, sumsq_partials=None):
    blk = 0
    for s in plan['specs']:
      table = s.table.detach().cpu().numpy()
      ids = s.ids.cpu().numpy()
      offsets = None if s.offsets is None else s.offsets.cpu().numpy()
      weights = None if s.weights is None else s.weights.detach().cpu().numpy()
      res = lookup_rows_fast(table, ids, offsets, weights, s.combiner, s.n_rows)
      s.out[:s.n_rows, s.out_col:s.out_col + s.dim] = torch.from_numpy(res)
      V = 4 if s.dim % 4 == 0 else 1
      G = 1
      while G < s.dim // V:
        G <<= 1
      rpb = 256 // G
      nb = (max(s.n_rows, 1) + rpb - 1) // rpb
      if sumsq_partials is not None:
        sq = (res.astype(np.float64)**2).sum(axis=1)
        for b in range(nb):
          sumsq_partials[blk + b] = float(sq[b * rpb:(b + 1) * rpb].sum())
      blk += nb

  def emb_group_create(self, specs, dim, total_rows, var, m, v, bitmap):
    n_ent = sum((s.max_nnz if s.offsets is not None else s.n_rows) for s in specs)
    return {'specs': list(specs), 'dim': dim, 'total_rows': total_rows, 'var': var, 'm': m, 'v': v,
            'bitmap': bitmap, 'num_entries': n_ent}

  def emb_group_destroy(self, group):
    pass

  def emb_bwd_update(self, group, opt_kind, hyper):
    h = hyper.detach().cpu().numpy().reshape(-1)
    if group.get('n_active', -1) >= 0:
      grads = {}
      for key, _, s, r, scale in self._routed_entries(group):
        g = s.out.detach().cpu().numpy()[r, s.out_col:s.out_col + group['dim']].astype(np.float32) * scale
        grads[key] = (grads[key] + g).astype(np.float32) if key in grads else g.astype(np.float32)
    else:
      grads = sparse_grads(group['specs'], group['dim'])
    var = group['var'].detach().numpy()
    m = None if group['m'] is None else group['m'].numpy()
    v = None if group['v'] is None else group['v'].numpy()
    if opt_kind == OPT_ADAM and 'last_step' in group:
      apply_sparse(var, m, v, grads, OPT_LAZY_ADAM, h)  # touched rows; the others decay lazily (emb_catch_up)
      t = int(group['step_counter'].item()) - 1
      for key in grads:
        group['last_step'][int(key)] = t
      return
    apply_sparse(var, m, v, grads, opt_kind, h)

  def emb_bwd_reduce(self, group, out=None):
    grads = sparse_grads(group['specs'], group['dim'])
    keys = sorted(grads.keys())
    n = group['num_entries']
    k = torch.zeros(n, dtype=torch.int32)
    g = torch.zeros(n, group['dim'], dtype=torch.float32)
    for i, key in enumerate(keys):
      k[i] = key
      g[i] = torch.from_numpy(grads[key])
    return k, g, torch.tensor([len(keys)], dtype=torch.int32)

  # -- GEMM: the same torch-CPU fp32 matmul calls the model oracle makes (x @ w, dy @ w.T, x.T @ dy), so the
  #    host-logic tests stay bit-comparable; bf16=True rounds the operands to bfloat16 first, as the kernel does
  def gemm_reserve(self, floats):
    pass

  def gemm_row_tiles(self, M):
    return (int(M) + 63) // 64

  def bn_apply_from_stats(self, x, bias, col_stats, chunks, gamma, beta, eps, momentum, moving_mean, moving_var, act):
    # the statistics are recomputed from x: same values as the Welford merge up to rounding
    return self.bn_act_fwd(x, bias, gamma, beta, True, eps, momentum, moving_mean, moving_var, act)

  def gemm(self, layout, a, b, out=None, bias=None, accumulate=False, bf16=False, col_stats=None):
    def rnd(t):
      return t.to(torch.bfloat16).to(torch.float32) if bf16 else t
    A, Bm = rnd(a.detach()), rnd(b.detach())
    if layout == 0:
      r = A @ Bm
    elif layout == 1:
      r = A @ Bm.t()
    else:
      r = A.t() @ Bm
    if bias is not None:
      r = r + bias.detach()
    if out is None:
      return r
    if accumulate:
      out.add_(r)
    else:
      out.copy_(r)
    return out

  grouped_stacks = True  # layers/dnn.py run_parallel: the lock-step host logic runs on the stand-in too

  grouped_bn = True
  BN_MULTI_MAX_ROWS = 8192

  def bn_fwd_multi(self, layers):
    return [self.bn_act_fwd(l['x'], l.get('bias'), l.get('gamma'), l.get('beta'), l['use_bn'], l.get('eps', 0.0),
                            l.get('momentum', 0.0), l.get('moving_mean'), l.get('moving_var'), l['act']) for l in layers]

  def bn_bwd_multi(self, layers):
    return [self.bn_act_bwd(l['x'], l.get('bias'), l.get('gamma'), l['y'], l.get('mean'), l.get('invstd'), l['dy'],
                            l['use_bn'], l['act'], l.get('bias') is not None, l.get('gamma') is not None,
                            into=l.get('into')) for l in layers]

  def gemm_grouped(self, layout, problems, bf16=False):
    for pr in problems:
      a, b, out, bias, accumulate = pr[:5]
      self.gemm(layout, a, b, out=out, bias=bias, accumulate=accumulate, bf16=bf16)  # (column statistics: recomputed by bn_apply_from_stats)

  # -- embedding-parallel routing (reference compat/feature_column/feature_column.py:248-357:
  #    owner = id % world, local row = id // world)
  def emb_group_set_routing(self, group, world, shard_stride, local_base):
    group['world'], group['shard_stride'], group['local_base'] = world, shard_stride, list(local_base)

  def emb_group_set_active(self, group, n_rows):
    group['n_active'] = int(n_rows)

  def _routed_entries(self, group):
    """[(routed key, entry index, spec, out row, scale)] in entry order."""
    W = group.get('world', 1)
    ents, base = [], 0
    for li, s in enumerate(group['specs']):
      ids = s.ids.cpu().numpy()
      offsets = None if s.offsets is None else s.offsets.cpu().numpy()
      weights = None if s.weights is None else s.weights.cpu().numpy()
      n_rows = min(s.n_rows, group.get('n_active', s.n_rows)) if group.get('n_active', -1) >= 0 else s.n_rows
      for r in range(n_rows):
        kb, ke = (r, r + 1) if offsets is None else (int(offsets[r]), int(offsets[r + 1]))
        valid = []
        for k in range(kb, ke):
          i = int(ids[k])
          w = F32(1) if weights is None else F32(weights[k])
          if i < 0 or i >= s.rows or (weights is not None and s.combiner != 0 and not (w > 0)):
            continue
          valid.append((k, i, w))
        den = F32(1)
        if s.combiner != 0 and valid:
          wsum, w2 = F32(0), F32(0)
          for _, _, w in valid:
            wsum, w2 = F32(wsum + w), F32(w2 + w * w)
          den = wsum if s.combiner == 1 else np.sqrt(w2, dtype=np.float32)
        for k, i, w in valid:
          key = (s.key_base + i) if 'local_base' not in group else \
              ((i % W) * group['shard_stride'] + group['local_base'][li] + i // W)
          ents.append((key, base + k, s, r, F32(w / den)))
      base += s.max_nnz if s.offsets is not None else s.n_rows
    return ents

  # grouped weight-gradient launch of the HIP backend: the reference contracts every layer on its own
  def defer_wgrads(self):
    pass

  def wgrad_sink(self):
    return _NoWgradSink()  # never active: every gradient is computed where it arises

  def flush_wgrads(self):
    pass

  fused_tail = True  # (the host logic of the fused tail runs on the stand-in too: nothing is ever queued here)

  def take_wgrads(self):
    return [], []

  tail_riders = True

  def flush_loss_tail(self):
    pass  # (loss_tail(defer=True) runs at once here)

  def dense_opt_fits_the_tail(self, wgrads, w, grad):
    return False

  @staticmethod
  def wgrads_fit_the_tail(q):
    return False

  def emb_catch_up_multi(self, groups, unique_keys, n_unique, hyper):
    for g, uk, nu in zip(groups, unique_keys, n_unique):
      self.emb_catch_up(g, uk, nu, hyper)

  def emb_bwd_update_multi(self, groups, opt_kind, hyper):
    for g in groups:
      self.emb_bwd_update(g, opt_kind, hyper)

  fused_emb = True  # layers/input_layer.py: the fused single-GPU step's host logic runs on the stand-in too

  defer_catch_up = True

  def emb_fwd_lazy(self, plan, groups, hyper, sumsq_partials=None):
    """HipBackend.emb_fwd_lazy evaluates the pending decay of the rows it reads in registers and stores nothing; the
    stand-in brings them current in place (what the row update of the same step does anyway), then looks up."""
    lazy = [g for g in groups if g.get('last_step') is not None]
    if lazy:
      self.emb_catch_up_multi(lazy, [g['_front_keys'][0] for g in lazy], [g['_front_keys'][1] for g in lazy], hyper)
    self.emb_fwd(plan, sumsq_partials)

  prologue_tables = True

  def emb_front_fwd(self, groups, plan, hyper, skip_one_row, sumsq_partials=None):
    if not self.emb_front(groups, hyper, skip_one_row, defer=True):
      return False
    self.emb_fwd_lazy(plan, groups, hyper, sumsq_partials)
    return True

  def decay_tables_set_prologue_build(self, tabs, on):
    tabs['prologue_build'] = bool(on)

  def decay_tables_sync(self, tabs):
    tabs['lag'] = int(tabs['step_counter'].item())

  def decay_tables_error(self, tabs):
    return False

  def emb_front(self, groups, hyper, skip_one_row, defer=False):
    """easyrec_amd.kernels.HipBackend.emb_front restated with the general entry points: route every group (a follower of
    a shared sort after its leader), then bring the rows of the step current.  Eligibility as er_emb_front: dense-mode
    lookups on their own tables, closed-form (or no) lazy decay."""
    for g in groups:
      if any(sp.offsets is not None for sp in g['specs']) or 'local_base' in g:
        return False
    uks, nus = [], []
    for g in groups:
      leader = g.get('sort_leader')
      if leader is not None and any(leader is q for q in groups):
        self.emb_route(g, None, None, None, None)
        i = [k for k, q in enumerate(groups) if q is leader][0]
        uks.append(uks[i])
        nus.append(nus[i])
      else:
        uk = torch.zeros(max(g['num_entries'], 1), dtype=torch.int32)
        nu = torch.zeros(1, dtype=torch.int32)
        self.emb_route(g, uk, nu, None, None)
        uks.append(uk)
        nus.append(nu)
    for g, uk, nu in zip(groups, uks, nus):
      g['_front_keys'] = (uk, nu)
    lazy = [(g, uk, nu) for g, uk, nu in zip(groups, uks, nus) if g.get('last_step') is not None]
    if lazy and not defer:
      self.emb_catch_up_multi([x[0] for x in lazy], [x[1] for x in lazy], [x[2] for x in lazy], hyper)
    return True

  def emb_bwd_fused(self, groups, finish, opt_kind, hyper, wgrads=None, dense_opt=None):
    assert not wgrads and dense_opt is None
    self.group_grad_finish(finish)
    self.emb_bwd_update_multi(groups, opt_kind, hyper)

  def emb_group_share_sort(self, group, leader):
    if not _same_lookup_keys(group, leader):
      return False
    group['sort_leader'] = leader
    return True

  def emb_group_set_peer_capacity(self, group, peer_cap, count_header=True):
    group['peer_cap'], group['peer_hdr'], group['overflow'] = int(peer_cap), bool(count_header), False

  def emb_route_overflow(self, group):
    return bool(group.get('overflow', False))

  def emb_route(self, group, unique_keys, n_unique, entry_unique_index, owner_counts):
    ents = self._routed_entries(group)
    keys = sorted(set(e[0] for e in ents))
    if unique_keys is None:  # follower of a shared sort
      assert group.get('sort_leader') is not None and group['sort_leader'].get('_route_keys') == keys
      group['_route_keys'], group['_route_pos'] = keys, group['sort_leader']['_route_pos']
      return
    W = group.get('world', 1)
    stride = group['shard_stride'] if 'local_base' in group else group['total_rows']
    cap = group.get('peer_cap', 0)
    if cap:  # padded layout: owner w's keys at [w * cap, ...)
      pos, seen = {}, [0] * W
      for k in keys:
        w = k // stride
        if seen[w] < cap:
          pos[k] = w * cap + seen[w]
        else:
          group['overflow'] = True
        seen[w] += 1
    else:
      pos = {k: i for i, k in enumerate(keys)}
    if entry_unique_index is not None:
      entry_unique_index.fill_(-1)
      for key, j, _, _, _ in ents:
        entry_unique_index[j] = pos.get(key, -1)
    hdr = 1 if (cap and group.get('peer_hdr')) else 0
    per_owner = [sum(1 for k in keys if w * stride <= k < (w + 1) * stride) for w in range(W)]
    for k, i in pos.items():  # (i = the row slot; with a header the key sits 1 + owner slots further)
      unique_keys[i + (i // cap + 1 if hdr else 0)] = k
    if hdr:
      for w in range(W):
        unique_keys[w * (cap + 1)] = per_owner[w]
    n_unique[0] = len(keys)
    if owner_counts is not None:
      for w in range(W):
        owner_counts[w] = per_owner[w]
    group['_route_keys'], group['_route_pos'] = keys, pos

  def emb_bwd_reduce_routed(self, group, unique_grads):
    acc = {}
    for key, _, s, r, scale in self._routed_entries(group):
      g = s.out.detach().cpu().numpy()[r, s.out_col:s.out_col + group['dim']].astype(np.float32) * scale
      acc[key] = (acc[key] + g).astype(np.float32) if key in acc else g.astype(np.float32)
    for key, i in group['_route_pos'].items():
      unique_grads[i] = torch.from_numpy(acc[key])

  def emb_owner_ids(self, recv_keys, counts, n_runs, peer_cap, key_sub, ids, counts_out):
    hdr = 1 if counts is None else 0
    for q in range(n_runs):
      b = q * (peer_cap + hdr)
      c = int(recv_keys[b]) if hdr else int(counts[q])
      ids[q * peer_cap:(q + 1) * peer_cap] = -1
      ids[q * peer_cap:q * peer_cap + c] = recv_keys[b + hdr:b + hdr + c].to(torch.int64) - int(key_sub)
      counts_out[q] = c

  def emb_owner_merge_padded(self, group, counts, n_runs, peer_cap):
    assert group['num_entries'] == n_runs * peer_cap and group.get('n_active', -1) < 0

  def gather_rows(self, table, keys, n, key_sub, out):
    rows = keys[:n].to(torch.int64) - int(key_sub)
    out[:n] = table[rows]

  def scatter_unique(self, keys, grads, n_unique, capacity, dim, dense):
    n = int(n_unique.item())
    rows = keys[:n].to(torch.int64)
    dense[rows, :dim] = grads[:n]
    dense[rows, dim] = 1.0

  def emb_owner_merge(self, group, run_counts):
    assert sum(int(x) for x in run_counts) == group['n_active']

  def emb_owner_serve(self, groups, rows_out, hyper):
    """The sorted form: de-duplicate the received rows, catch them up, gather."""
    for g, out in zip(groups, rows_out):
      n = g['n_active'] if g.get('n_active', -1) >= 0 else g['num_entries']
      if n == 0:
        continue
      if g.get('last_step') is not None and hyper is not None:  # (hyper None: inference, the rows were flushed)
        uk = torch.zeros(g['num_entries'], dtype=torch.int32)
        nu = torch.zeros(1, dtype=torch.int32)
        self.emb_route(g, uk, nu, None, None)
        self.emb_catch_up(g, uk, nu, hyper)
      spec = g['specs'][0]
      ok = spec.ids[:n] >= 0  # (padding of the fixed-capacity exchange)
      out[:n][ok] = g['var'].detach()[spec.key_base + spec.ids[:n][ok]]

  ep_merged_reduce = True  # (the host logic of the merged requester tail runs on the stand-in too)

  def emb_reduce_local_tail(self, routed, dense, wgrads=None):
    assert not wgrads  # (never queued here)
    if dense:
      self.emb_bwd_reduce_dense([g for g, _ in dense], [d for _, d in dense])
    for g, t in routed:
      self.emb_bwd_reduce_routed(g, t)

  def emb_bwd_reduce_dense(self, groups, dense):
    for g, d in zip(groups, dense):  # the two-step form: de-duplicated rows, then the scatter
      keys, grads, n_unique = self.emb_bwd_reduce(g)
      self.scatter_unique(keys, grads, n_unique, keys.numel(), g['dim'], d)

  def emb_dense_apply(self, tables, opt_kind, hyper):
    """Restated through the sparse path: the rows with a count are the ids of one lookup whose upstream gradient is
    the dense buffer; TF-exact Adam then sweeps the others."""
    for var, m, v, dense in tables:
      n, dim = var.shape
      ids = torch.where(dense[:, dim] > 0, torch.arange(n, dtype=torch.int64), torch.full((n,), -1, dtype=torch.int64))
      bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32) if opt_kind == OPT_ADAM else None
      spec = _Spec(table=var, ids=ids, offsets=None, weights=None, out=dense, out_col=0, rows=n, key_base=0,
                   dim=dim, combiner=0, n_rows=n, max_nnz=n, name='dense_apply')
      grp = self.emb_group_create([spec], dim, n, var, m, v, bitmap)
      self.emb_bwd_update(grp, opt_kind, hyper)

  def emb_mark_touched(self, group):
    pass

  def adam_decay_sweep(self, var, m, v, bitmap, total_rows, dim, hyper):
    h = hyper.detach().cpu().numpy().reshape(-1)
    bits = None if bitmap is None else bitmap.cpu().numpy().view(np.uint32)
    rows = np.arange(total_rows)
    live = np.ones(total_rows, dtype=bool)
    if bits is not None:
      live = ((bits[rows >> 5] >> (rows & 31).astype(np.uint32)) & 1) == 0
    vn, mn, vv = var.numpy(), m.numpy(), v.numpy()
    mn[live] = mn[live] * F32(h[HYPER_BETA1])
    vv[live] = vv[live] * F32(h[HYPER_BETA2])
    vn[live] = vn[live] - (F32(h[HYPER_LR_T]) * mn[live]) / (np.sqrt(vv[live], dtype=np.float32) +
                                                          F32(h[HYPER_EPS]))

  # -- FM / wide
  def fm_fwd(self, x, F, D):
    e = x[:, :F * D].reshape(x.shape[0], F, D)
    S = e.sum(dim=1)
    q = (e * e).sum(dim=1)
    return 0.5 * (S * S - q), S

  def fm_bwd(self, x, S, g, F, D, into=None, accumulate=False):
    e = x[:, :F * D].reshape(x.shape[0], F, D)
    dx = (g[:, None, :] * (S[:, None, :] - e)).reshape(x.shape[0], F * D)
    return self._into(dx, into, accumulate)

  @staticmethod
  def _into(val, into, accumulate):
    if into is None:
      return val
    with torch.no_grad():
      if accumulate:
        into.add_(val)
      else:
        into.copy_(val)
    return into

  def grouped_auc(self, keys, preds, labels, reduction):
    """er_grouped_auc restated: per key with both classes, the Mann-Whitney AUC with tied predictions at their average
    rank; -> (sum of w * auc, sum of w, keys used)."""
    keys, preds = keys.reshape(-1).numpy(), preds.reshape(-1).numpy().astype(np.float32)
    pos = labels.reshape(-1).numpy() != 0
    total = wsum = used = 0.0
    for key in np.unique(keys):
      sel = keys == key
      p, y = preds[sel].astype(np.float64), pos[sel]
      n_pos, n_neg = int(y.sum()), int((~y).sum())
      if n_pos == 0 or n_neg == 0:
        continue
      ranks = np.array([((p < v).sum() + 1 + (p <= v).sum()) / 2.0 for v in p])
      auc = (ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * float(n_neg))
      w = (1.0, float(p.size), float(n_pos))[int(reduction)]
      total, wsum, used = total + w * auc, wsum + w, used + 1
    return total, wsum, used

  def auc_update(self, probs, labels, weights, thresholds, counts):
    """tf.metrics.auc's per-threshold `prediction > threshold` counts, as a histogram (core/metrics.py)."""
    p = probs.detach().reshape(-1).cpu().numpy().astype(np.float32)
    y = labels.detach().reshape(-1).cpu().numpy() != 0
    t = thresholds.cpu().numpy().astype(np.float32)
    keep = np.ones(len(p), dtype=bool) if weights is None else (weights.reshape(-1).cpu().numpy() > 0)
    bucket = (p[:, None] > t[None, :]).sum(axis=1)
    for b, pos, k in zip(bucket, y, keep):
      if k:
        counts[int(pos), int(b)] += 1

  @staticmethod
  def _dot_pairs(F, self_interaction):
    off = 0 if self_interaction else 1
    return [(i, j) for i in range(F) for j in range(i + off, F)]  # model/dlrm.py:51-57

  def dot_interaction_fwd(self, x, F, D, self_interaction):
    e = x[:, :F * D].reshape(x.shape[0], F, D)
    inter = torch.einsum('bne,bme->bnm', e, e)
    return torch.stack([inter[:, i, j] for i, j in self._dot_pairs(F, self_interaction)], dim=1)

  def dot_interaction_bwd(self, x, g, F, D, self_interaction):
    e = x[:, :F * D].reshape(x.shape[0], F, D)
    de = torch.zeros_like(e)
    for p, (i, j) in enumerate(self._dot_pairs(F, self_interaction)):
      de[:, i] += g[:, p:p + 1] * e[:, j]
      de[:, j] += g[:, p:p + 1] * e[:, i]
    return de.reshape(x.shape[0], F * D)

  def rowsum_fwd(self, x, n):
    return x[:, :n].sum(dim=1, keepdim=True)

  def rowsum_bwd(self, g, n, into=None, accumulate=False):
    return self._into(g.reshape(-1, 1).expand(-1, n).contiguous(), into, accumulate)

  def copy_multi(self, pairs):
    for dst, src in pairs:
      dst.copy_(src)

  def concat_cols(self, parts):
    return torch.cat([t.detach() for t in parts], dim=1)

  fused_wide_fm = True

  def wide_fm_concat(self, wide, fm_x, F, D, deep):
    fm, S = self.fm_fwd(fm_x, F, D)
    return torch.cat([self.rowsum_fwd(wide, wide.shape[1]), fm, deep.detach()], dim=1), S

  def group_grad_finish(self, groups):
    for dout, out, lam, has_base, terms in groups:
      v = dout.detach().clone() if has_base else torch.zeros_like(dout)
      o = out.detach()
      for term in terms:
        if term[0] == 'rowsum':
          _, g, col0, width = term
          v[:, col0:col0 + width] += g.detach().reshape(-1, 1)
        else:
          _, g, saved, col0, width, dim = term
          F = width // dim
          x3 = o[:, col0:col0 + width].reshape(o.shape[0], F, dim)
          v[:, col0:col0 + width] += (g.detach().reshape(-1, 1, dim) * (saved.reshape(-1, 1, dim) - x3)).reshape(o.shape[0], width)
      if lam:
        v += F32(lam) * o
      dout.copy_(v)

  def axpy2d(self, x, alpha, y, accumulate=True):
    if accumulate:
      y.add_(x, alpha=alpha)
    else:
      y.copy_(x * alpha)

  # -- cross
  def cross_v1_fwd(self, x0, w, b):
    x = x0
    dots = []
    for l in range(w.shape[0]):
      xw = (x * w[l]).sum(dim=1, keepdim=True)
      dots.append(xw)
      x = (x0 * xw + b[l]) + x
    return x, torch.cat(dots, dim=1)

  def cross_v1_bwd(self, x0, w, b, dots, dout):
    x0_ = x0.detach().clone().requires_grad_(True)
    w_ = w.detach().clone().requires_grad_(True)
    b_ = b.detach().clone().requires_grad_(True)
    with torch.enable_grad():
      out, _ = self.cross_v1_fwd(x0_, w_, b_)
      out.backward(dout)
    return x0_.grad, w_.grad, b_.grad

  def cross_v2_fwd(self, x0, x, u, bias, diag_scale):
    t = u if bias is None else u + bias
    if diag_scale != 0:
      t = t + diag_scale * x
    return x0 * t + x

  # the fused cross layer of the HIP backend (kernels.CrossLayerFn): same results from the separate steps
  fused_cross = True

  def _cross_b16(self, w, bf16, *rows):
    return None

  def cross_fwd_fused(self, x0, x, w, bias, diag, bf16):
    u = self.gemm(0, x, w, bf16=bf16)
    return self.cross_v2_fwd(x0, x, u, bias, diag), u

  def cross_bwd_top(self, x0, x, u, bias, diag, dout, dx0, acc0, bf16, w):
    g0, _, du = self.cross_v2_bwd(x0, x, u, bias, diag, dout)
    if acc0:
      dx0.add_(g0)
    else:
      dx0.copy_(g0)
    return du, du.sum(dim=0, keepdim=True)

  def cross_dgrad_fused(self, du, w, dout, diag, bf16, dst, acc, prev=None):
    v = self.gemm(1, du, w, bf16=bf16) + dout
    if diag != 0:
      v = v + diag * du
    if acc:
      dst.add_(v)
    else:
      dst.copy_(v)
    if prev is None:
      return None, None
    return self.cross_bwd_top(prev['x0'], prev['xl'], prev['u'], prev['bias'], diag, v, prev['dx0'], prev['acc0'], bf16, w)

  def queue_colsum(self, sink, partial, dst, n_cols):
    dst.add_(partial.sum(dim=0))

  def colsum_partials_multi(self, jobs, accumulate=True):
    for partial, dst, n_cols in jobs:
      if accumulate:
        dst.add_(partial.sum(dim=0))
      else:
        dst.copy_(partial.sum(dim=0))

  def cross_v2_bwd_acc(self, x0, x, u, bias, diag_scale, dout, dx0, acc0, dx, accx):
    g0, gx, du = self.cross_v2_bwd(x0, x, u, bias, diag_scale, dout)
    if dx is None:
      g0 = g0 + gx
    if acc0:
      dx0.add_(g0)
    else:
      dx0.copy_(g0)
    if dx is not None:
      if accx:
        dx.add_(gx)
      else:
        dx.copy_(gx)
    return du

  def cross_v2_bwd(self, x0, x, u, bias, diag_scale, dout):
    t = u if bias is None else u + bias
    if diag_scale != 0:
      t = t + diag_scale * x
    dx = dout + (dout * x0 * diag_scale if diag_scale != 0 else 0)
    return dout * t, dx, dout * x0

  # -- DIN
  # the first attention layer on the generated operand (kernels.DINFirstLayerFn): same results through the built block
  din_fused = True

  @staticmethod
  def din_gemm_ok(q, h, w):
    return h.dim() == 3 and q.dim() == 2 and h.shape[-1] % 16 == 0 and h.shape[1] >= 2

  def din_gemm_fwd(self, q, h, w, bias, col_stats=None):
    B, L, E = h.shape
    return self.gemm(0, self.din_concat_fwd(q, h).reshape(B * L, 4 * E), w, bias=bias)

  def din_gemm_wgrad(self, q, h, dz, out, accumulate=True):
    B, L, E = h.shape
    return self.gemm(2, self.din_concat_fwd(q, h).reshape(B * L, 4 * E), dz, out=out, accumulate=accumulate)

  def din_gemm_dgrad(self, dz, w, q, h, dh=None, acc_h=False):
    B, L, E = h.shape
    dcat = self.gemm(1, dz, w).reshape(B, L, 4 * E)
    return self.din_concat_bwd(q, h, dcat, dh=dh, acc_h=acc_h)

  def din_concat_fwd(self, q, h):
    B, L, E = h.shape
    qq = q[:, None, :].expand(B, L, E)
    return torch.cat([qq, h, qq - h, qq * h], dim=-1)

  def din_concat_bwd(self, q, h, dout, dh=None, acc_h=False):
    B, L, E = h.shape
    g0, g1, g2, g3 = dout[..., :E], dout[..., E:2 * E], dout[..., 2 * E:3 * E], dout[..., 3 * E:]
    qq = q[:, None, :]
    dq = (g0 + g2 + g3 * h).sum(dim=1)
    r = g1 - g2 + g3 * qq
    if dh is None:
      return dq, r
    if acc_h:
      dh.add_(r)
    else:
      dh.copy_(r)
    return dq, dh

  def din_pool_fwd(self, scores, hist, seq_len, scale=1.0):
    B, L, E = hist.shape
    mask = torch.arange(L)[None, :] < seq_len[:, None].to(torch.int64)
    pad = torch.full_like(scores, float(-2**32 + 1))
    s = torch.where(mask, scores * scale, pad)
    p = torch.softmax(s, dim=1)
    return torch.bmm(p[:, None, :], hist)[:, 0, :], p

  def din_pool_bwd(self, probs, hist, seq_len, dout, scale=1.0, dhist=None, acc_h=False):
    B, L, E = hist.shape
    dp = torch.bmm(hist, dout[:, :, None])[:, :, 0]
    ds = probs * (dp - (probs * dp).sum(dim=1, keepdim=True))
    mask = torch.arange(L)[None, :] < seq_len[:, None].to(torch.int64)
    ds = torch.where(mask, ds * scale, torch.zeros_like(ds))
    dh = probs[:, :, None] * dout[:, None, :]
    if dhist is None:
      return ds, dh
    if acc_h:
      dhist.add_(dh)
    else:
      dhist.copy_(dh)
    return ds, dhist

  # -- hash-table (KV) embedding tables: a python dict for the map, the same counter-based row initialiser in numpy
  @staticmethod
  def _mix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over='ignore'):
      x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
      x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
      x = x ^ (x >> np.uint64(31))
    return x

  @classmethod
  def kv_init_value(cls, seed, keys, dim, mean, stddev):
    """er_kv.hip kv_init_value, restated: [len(keys), dim] fp32."""
    keys = np.asarray(keys, dtype=np.int64).astype(np.uint64)
    with np.errstate(over='ignore'):
      base = cls._mix64(np.uint64(seed) ^ cls._mix64(keys))[:, None] + \
          np.arange(dim, dtype=np.uint64)[None, :] * np.uint64(0x9E3779B97F4A7C15)
      u = [(cls._mix64(base + np.uint64(k) * np.uint64(0xD1B54A32D192ED03)) >> np.uint64(40)).astype(np.float32) *
           np.float32(5.9604644775390625e-08) for k in range(4)]
    z = (((u[0] + u[1]) + (u[2] + u[3])) - np.float32(2.0)) * np.float32(1.7320508075688772)
    return (np.float32(mean) + np.float32(stddev) * z).astype(np.float32)

  def kv_create(self, var_rows, capacity, seed, init_mean, init_stddev, filter_freq=0, steps_to_live=0, step=None):
    """map: key -> arena row (-1: tracked by the counter filter, no row yet); freq / version: key -> count / step."""
    return {'map': {}, 'freq': {}, 'version': {}, 'n_rows': 0, 'capacity': int(capacity), 'var': var_rows,
            'dim': int(var_rows.shape[1]), 'seed': int(seed) & ((1 << 63) - 1), 'mean': float(init_mean),
            'stddev': float(init_stddev), 'overflow': torch.zeros(1, dtype=torch.int32), 'filter_freq': int(filter_freq),
            'steps_to_live': int(steps_to_live), 'step': step}

  def _kv_insert(self, kv, ids):
    """A training lookup's insert launch (er_kv.hip kv_insert_one): every occurrence counts, a key gets its row when
    its count reaches filter_freq (plain tables: on first sight)."""
    counted = kv['filter_freq'] > 1
    now = int(kv['step'].item()) if kv['step'] is not None else 0
    for k in ids.view(-1).tolist():
      if k < 0:
        continue
      known = k in kv['map']
      if not known:
        full = (len(kv['map']) >= 4 * kv['capacity']) if counted else (kv['n_rows'] >= kv['capacity'])
        if full:
          kv['overflow'][0] = 1  # (a full map claims no further key: the id reads zeros, the flag is sticky)
          continue
        kv['map'][k] = -1
      if kv['steps_to_live'] > 0:
        kv['version'][k] = now
      create = not known
      if counted:
        create = False
        if kv['freq'].get(k, 0) < kv['filter_freq']:
          kv['freq'][k] = kv['freq'].get(k, 0) + 1
          create = kv['freq'][k] == kv['filter_freq']
      if create:
        if kv['n_rows'] >= kv['capacity']:
          kv['overflow'][0] = 1
          continue
        r = kv['n_rows']
        kv['n_rows'] += 1
        kv['var'].detach()[r] = torch.from_numpy(self.kv_init_value(kv['seed'], [k], kv['dim'], kv['mean'], kv['stddev'])[0])
        kv['map'][k] = r

  @staticmethod
  def _kv_find(kv, ids, rows_out):
    out = [-1 if k < 0 else kv['map'].get(k, -1) for k in ids.view(-1).tolist()]
    rows_out.view(-1).copy_(torch.tensor(out, dtype=torch.int64))

  def kv_translate(self, kv, ids, rows_out, insert):
    if insert:
      self._kv_insert(kv, ids)
    self._kv_find(kv, ids, rows_out)

  def kv_jobs_create(self, jobs):
    return {'jobs': jobs}

  def kv_translate_multi(self, handle, insert):
    # (one insert launch over every job, then one find launch: er_kv_translate_multi)
    spans = []
    for job in handle['jobs']:
      kv, ids, rows_out = job[:3]
      n = ids.numel() if len(job) < 4 or job[3] is None else min(ids.numel(), int(job[3].item()))
      spans.append((kv, ids.view(-1)[:n], rows_out.view(-1)[:n]))
    if insert:
      for kv, ids, _ in spans:
        self._kv_insert(kv, ids)
    for kv, ids, rows_out in spans:
      self._kv_find(kv, ids, rows_out)

  def kv_route_create(self, jobs, world):
    offs = [0]
    for job in jobs:
      offs.append(offs[-1] + job[0].numel())
    C = max(offs[-1], 1)
    buf = lambda: torch.full((world, C), -1, dtype=torch.int64)
    return {'jobs': jobs, 'world': int(world), 'C': C, 'offs': offs, 'slots': [torch.full((j[0].numel(),), -1, dtype=torch.int64) for j in jobs],
            'send': buf(), 'recv': buf(), 'owner_rows': buf(), 'back': buf()}

  def kv_bucket(self, h):
    W, C = h['world'], h['C']
    h['send'].fill_(-1)
    for j, job in enumerate(h['jobs']):
      ids = job[0].view(-1)
      n = ids.numel() if len(job) < 3 or job[2] is None else min(ids.numel(), int(job[2].item()))
      counts = [0] * W
      for i, key in enumerate(ids.tolist()):
        if i >= n or key < 0:
          h['slots'][j][i] = -1
          continue
        o = key % W
        at = o * C + h['offs'][j] + counts[o]
        counts[o] += 1
        h['send'].view(-1)[at] = key
        h['slots'][j][i] = at

  def kv_unbucket(self, h):
    W, C = h['world'], h['C']
    back = h['back'].view(-1)
    for j, job in enumerate(h['jobs']):
      at = h['slots'][j]
      r = back[at.clamp(min=0)]
      job[1].view(-1).copy_(torch.where((at >= 0) & (r >= 0), r * W + at // C, torch.full_like(r, -1)))

  def kv_export(self, kv):
    items = sorted((k, r) for k, r in kv['map'].items() if r >= 0)
    return (torch.tensor([k for k, _ in items], dtype=torch.int64), torch.tensor([r for _, r in items], dtype=torch.int64))

  def kv_export_all(self, kv):
    items = sorted(kv['map'].items())
    keys = [k for k, _ in items]
    return (torch.tensor(keys, dtype=torch.int64), torch.tensor([r for _, r in items], dtype=torch.int64),
            torch.tensor([kv['freq'].get(k, 0) for k in keys], dtype=torch.int32),
            torch.tensor([kv['version'].get(k, 0) for k in keys], dtype=torch.int32))

  def kv_rebuild(self, kv, keys, rows, freq=None, version=None):
    keys, rows = [int(k) for k in keys.view(-1).tolist()], [int(r) for r in rows.view(-1).tolist()]
    assert len(set(keys)) == len(keys)
    kv['map'] = dict(zip(keys, rows))
    kv['n_rows'] = sum(1 for r in rows if r >= 0)
    kv['freq'] = dict(zip(keys, freq.view(-1).tolist())) if (freq is not None and kv['filter_freq'] > 1) else {}
    kv['version'] = dict(zip(keys, version.view(-1).tolist())) if (version is not None and kv['steps_to_live'] > 0) else {}

  # -- CIN (xDeepFM): strided views of the operands, plain torch arithmetic
  @staticmethod
  def _cin_view(x, strides, B, H, D):
    return torch.as_strided(x, (B, H, D), strides)

  def cin_outer_fwd(self, xi, strides, H, x0, z):
    B, H0, D = x0.shape
    xv = self._cin_view(xi, strides, B, H, D)
    z.copy_((xv.permute(0, 2, 1)[:, :, :, None] * x0.permute(0, 2, 1)[:, :, None, :]).reshape(B * D, H * H0))

  def cin_act_pool_fwd(self, c, bias, B, D, pooled, col0):
    N = c.shape[1]
    c.copy_(torch.relu(c + bias))
    pooled[:, col0:col0 + N] = c.reshape(B, D, N).sum(dim=1)

  def cin_act_pool_bwd(self, fm, dpooled, col0, dnext, B, D, dc):
    N = fm.shape[1]
    g = dpooled[:, col0:col0 + N][:, None, :].expand(B, D, N).reshape(B * D, N)
    if dnext is not None:
      g = g + dnext
    dc.copy_(torch.where(fm > 0, g, torch.zeros_like(g)))

  def cin_outer_bwd(self, dz, xi, strides, H, x0, dxi, add_xi, dx0):
    B, H0, D = x0.shape
    xv = self._cin_view(xi, strides, B, H, D).clone()
    dzv = dz.reshape(B, D, H, H0)
    d0 = torch.einsum('bdhm,bhd->bmd', dzv, xv)
    di = torch.einsum('bdhm,bmd->bhd', dzv, x0)
    dx0.add_(d0)
    out = self._cin_view(dxi, strides, B, H, D)
    if add_xi:
      out.add_(di)
    else:
      out.copy_(di)

  # -- MLP pieces
  def bn_act_fwd(self, x, bias, gamma, beta, use_bn, eps, momentum, moving_mean, moving_var, act):
    z = x if bias is None else x + bias
    mean = invstd = None
    if use_bn == BN_FROZEN:  # batch_normalization(training=False): the moving statistics, untouched
      mean = moving_mean.clone()
      invstd = 1.0 / torch.sqrt(moving_var + eps)
      y = (z - mean) * invstd
      y = y * (gamma if gamma is not None else 1.0) + (beta if beta is not None else 0.0)
    elif use_bn:
      mean = z.mean(dim=0)
      var = ((z - mean)**2).mean(dim=0)  # tf.nn.moments: biased
      invstd = 1.0 / torch.sqrt(var + eps)
      y = (z - mean) * invstd
      y = y * (gamma if gamma is not None else 1.0) + (beta if beta is not None else 0.0)
      if moving_mean is not None:
        moving_mean.sub_((moving_mean - mean) * (1.0 - momentum))
        moving_var.sub_((moving_var - var) * (1.0 - momentum))
    else:
      y = z
    if act == ACT_RELU:
      y = torch.relu(y)
    return y, mean, invstd

  def bn_act_bwd(self, x, bias, gamma, y, mean, invstd, dy, use_bn, act, need_bias, need_affine, into=None,
                 partial=None, beta=None):
    assert partial is None  # only the HIP backend fuses the column sums into the dgrad GEMM
    res = self._bn_act_bwd(x, bias, gamma, y, mean, invstd, dy, use_bn, act, need_bias, need_affine)
    if into is None:
      return res
    dx, dbias, dgamma, dbeta = res
    for buf, g in zip(into, (dbias, dgamma, dbeta)):
      if buf is not None and g is not None:
        buf.add_(g)
    return dx, None, None, None

  def _bn_act_bwd(self, x, bias, gamma, y, mean, invstd, dy, use_bn, act, need_bias, need_affine):
    B = x.shape[0]
    g = dy * (y > 0).to(dy.dtype) if act == ACT_RELU else dy
    dbias = dgamma = dbeta = None
    if use_bn:
      z = x if bias is None else x + bias
      xh = (z - mean) * invstd
      sg, sgx = g.sum(dim=0), (g * xh).sum(dim=0)
      ga = gamma if gamma is not None else 1.0
      if use_bn == BN_FROZEN:
        dx = ga * invstd * g
      else:
        dx = ga * invstd * (g - sg / B - xh * (sgx / B))
      if need_affine:
        dgamma, dbeta = sgx, sg
      if need_bias:
        dbias = ga * invstd * sg if use_bn == BN_FROZEN else torch.zeros_like(sg)
    else:
      dx = g
      if need_bias:
        dbias = g.sum(dim=0)
    return dx, dbias, dgamma, dbeta

  def colsum(self, x, out=None, accumulate=False):
    r = x.detach().sum(dim=0)
    if out is None:
      return r
    if accumulate:
      out += r
    else:
      out.copy_(r)
    return out

  def dice_fwd(self, x, alpha, eps, momentum, moving_mean, moving_var):
    mean = x.mean(dim=0)
    var = ((x - mean)**2).mean(dim=0)
    invstd = 1.0 / torch.sqrt(var + eps)
    p = torch.sigmoid((x - mean) * invstd)
    if moving_mean is not None:
      moving_mean.sub_((moving_mean - mean) * (1.0 - momentum))
      moving_var.sub_((moving_var - var) * (1.0 - momentum))
    return alpha * (1.0 - p) * x + p * x, mean, invstd

  def dice_bwd(self, x, alpha, mean, invstd, dy):
    B = x.shape[0]
    xh = (x - mean) * invstd
    p = torch.sigmoid(xh)
    direct = dy * (alpha + (1.0 - alpha) * p)
    q = dy * x * (1.0 - alpha) * p * (1.0 - p)
    dx = direct + invstd * (q - q.sum(dim=0) / B - xh * ((q * xh).sum(dim=0) / B))
    dalpha = (dy * (1.0 - p) * x).sum(dim=0)
    return dx, dalpha

  # -- loss
  def sigmoid_ce(self, logits, labels, weights, loss_scale=1.0):
    z, y = logits.to(torch.float32), labels.to(torch.float32)
    ce = torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-torch.abs(z)))
    w = torch.ones_like(z) if weights is None else weights
    nz = torch.clamp((w != 0).sum().to(torch.float32), min=1.0)
    loss = (w * ce).sum().reshape(1) / nz * loss_scale
    p = torch.sigmoid(z)
    return loss, loss_scale * w * (p - y) / nz, p

  def sigmoid_ce_multi(self, heads):
    return [self.sigmoid_ce(z.reshape(-1), y.reshape(-1), w, s)[:2] for z, y, w, s in heads]

  def total_loss(self, reg_emb, reg_dense, losses, reports, reg_out, total_out):
    reg_out.copy_(reg_emb + reg_dense)
    total = reg_out.clone()
    for src, dst in zip(losses, reports):
      dst.copy_(src.reshape(1))
      total = total + src.reshape(1)
    total_out.copy_(total)

  fused_head = True  # layers/dnn.py dense(head=True) + builders/loss_builder.py: the host logic runs on the stand-in too

  def head_sigmoid_ce(self, x, w, b, labels, loss_scale, src=None, logits=None):
    """easyrec_amd.kernels.HipBackend.head_sigmoid_ce restated with torch ops (per-64-row-tile partial sums like the kernel)."""
    B, K = x.shape
    xd, wd = x.detach().to(torch.float32), w.detach().reshape(K, 1)
    z = xd @ wd
    if b is not None:
      z = z + b.detach().reshape(1, 1)
    if logits is None:
      logits = torch.empty(B, 1, dtype=torch.float32)
    logits.copy_(z)
    zf, y = z.reshape(-1), labels.to(torch.float32).reshape(-1)
    ce = torch.clamp(zf, min=0) - zf * y + torch.log1p(torch.exp(-torch.abs(zf)))
    p = torch.sigmoid(zf)
    dz = loss_scale * (p - y) / torch.tensor(float(B))  # (sigmoid_ce's own expression, nz = B)
    dx = dz.reshape(B, 1) * wd.reshape(1, K)
    T = (B + 63) // 64
    pad = T * 64 - B

    def tiles(t):  # [B, C] -> per-tile column sums [T, C]
      t2 = torch.cat([t, torch.zeros(pad, t.shape[1])], 0) if pad else t
      return t2.reshape(T, 64, t.shape[1]).sum(dim=1)

    out = {'logits': logits, 'probs': p, 'dlogits': dz, 'dx': dx, 'loss_partials': tiles(ce.reshape(B, 1)).reshape(T),
           'wb_partials': torch.cat([tiles(xd * dz.reshape(B, 1)), tiles(dz.reshape(B, 1))], 1).contiguous(), 'bn_partials': None}
    if src is not None:
      g = dx.clone()
      if src.act == ACT_RELU:
        g = torch.where(xd > 0, g, torch.zeros_like(g))
      gx = torch.zeros_like(g)
      if src.mean is not None:
        gx = g * ((src.z.detach() - src.mean.reshape(1, K)) * src.invstd.reshape(1, K))
      out['bn_partials'] = torch.stack([tiles(g), tiles(gx)], dim=2).contiguous()
    return out

  def loss_tail(self, emb_partials, emb_scale, dense_partials, losses, reports, reg_out, total_out, jobs=(), defer=False):
    for partial, dst, n_cols in jobs:
      dst.add_(partial[:, :n_cols].sum(dim=0))
    scalars = []
    for t in losses:
      pl = getattr(t, '_er_partials', None)
      if pl is not None:
        t[0] = float(F32(pl[1]) * F32(pl[0].detach().numpy().astype(np.float64).sum()) / F32(pl[2]))
      scalars.append(t.reshape(1))
    self.reg_total_loss(emb_partials, emb_scale, dense_partials, scalars, reports, reg_out, total_out)

  def l2_partials(self, w, coef, partials):
    c, ww = coef.numpy().astype(np.float64), w.detach().numpy().astype(np.float64)
    t = np.zeros(partials.numel() * 256)
    t[:ww.size] = 0.5 * c * ww * ww
    partials.copy_(torch.from_numpy(t.reshape(-1, 256).sum(axis=1).astype(np.float32)))

  def reg_total_loss(self, emb_partials, emb_scale, dense_partials, losses, reports, reg_out, total_out):
    reg = F32(0)
    if emb_partials is not None and emb_partials.numel():
      reg = F32(emb_scale) * F32(emb_partials.detach().cpu().numpy().astype(np.float64).sum())
    if dense_partials is not None and dense_partials.numel():
      reg = F32(reg + F32(dense_partials.detach().numpy().astype(np.float64).sum()))
    reg_out[0] = float(reg)
    total = F32(reg)
    for src, dst in zip(losses, reports):
      dst[0] = float(src.reshape(-1)[0])
      total = F32(total + F32(src.reshape(-1)[0].item()))
    total_out[0] = float(total)

  def step_prologue(self, table, counter, out, history=None, zero=None, history_index=HYPER_LR_T, decay_tables=None,
                    hash_job=None):
    self.hyper_select(table, counter, out, history=history, history_index=history_index)
    if zero is not None:
      zero.zero_()
    if hash_job is not None:
      hb, ho, hpc, hk, hdrop, hout = hash_job
      self.hash_bucket_fast(hb, ho, hpc, hk, hdrop, out=hout)

  def reduce_sum(self, partials, scale, out, accumulate=False):
    s = partials.to(torch.float64).sum().to(torch.float32) * scale
    if accumulate:
      out.add_(s)
    else:
      out.fill_(float(s))

  def l2_loss(self, w, coef, out, accumulate=False):
    s = (coef.to(torch.float64) * 0.5 * w.detach().to(torch.float64)**2).sum().to(torch.float32)
    if accumulate:
      out.add_(s)
    else:
      out.fill_(float(s))

  # -- MMoE
  def mmoe_mix_fwd(self, experts, gate_logits):
    gates = torch.softmax(gate_logits, dim=-1)  # [T,B,E]
    out = torch.einsum('tbe,ebh->tbh', gates, experts)
    return out, gates

  def mmoe_mix_bwd(self, experts, gates, dout):
    dexperts = torch.einsum('tbe,tbh->ebh', gates, dout)
    dg = torch.einsum('tbh,ebh->tbe', dout, experts)
    dlogits = gates * (dg - (gates * dg).sum(dim=-1, keepdim=True))
    return dexperts, dlogits

  def hyper_select(self, table, counter, out, history=None, history_index=HYPER_LR_T):  # history: [values | maxima]
    c = int(counter.item())
    slot = table[c % table.shape[0]]
    out.copy_(slot)
    if history is not None and c < history.numel() // 2:
      cap = history.numel() // 2
      val = slot.reshape(-1)[history_index]
      history[c] = val
      history[cap + c] = max(float(val), float(history[cap + c - 1]) if c > 0 else 0.0)
    counter += 1

  # -- the closed-form replay of the HIP library (csrc/er_decay.h): the stand-in accepts the calls (so that the host
  # logic around them runs on CPU) and keeps replaying step by step - the recurrence the closed form is held to
  def decay_tables_create(self, lr_hist, step_counter, beta1, beta2):
    return {'handle': None, 'lr_hist': lr_hist, 'step_counter': step_counter, 'betas': (float(beta1), float(beta2))}

  def decay_tables_destroy(self, tabs):
    pass

  def emb_group_set_decay_tables(self, group, tabs):
    group['decay_tables'] = tabs

  # -- TF-exact Adam without the sweep: decay-only steps replayed when a row is next touched
  def emb_group_enable_lazy_decay(self, group, last_step, lr_hist, step_counter):
    group['last_step'], group['lr_hist'], group['step_counter'] = last_step, lr_hist, step_counter

  @staticmethod
  def _replay(group, rows, s_end, h):
    var, m, v = group['var'].detach().numpy(), group['m'].numpy(), group['v'].numpy()
    last, hist = group['last_step'].numpy(), group['lr_hist'].numpy()
    b1, b2, eps = F32(h[HYPER_BETA1]), F32(h[HYPER_BETA2]), F32(h[HYPER_EPS])
    for r in rows:
      for sidx in range(int(last[r]) + 1, s_end):
        if not (m[r].any() or v[r].any()):
          break
        lr_t = F32(hist[sidx])
        m[r] = m[r] * b1
        v[r] = v[r] * b2
        var[r] = var[r] - (lr_t * m[r]) / (np.sqrt(v[r], dtype=np.float32) + eps)

  def emb_catch_up(self, group, unique_keys, n_unique, hyper):
    h = hyper.detach().cpu().numpy().reshape(-1)
    t = int(group['step_counter'].item()) - 1
    n = int(n_unique.item())
    rows = [int(k) for k in unique_keys[:n].tolist()]
    self._replay(group, rows, t, h)
    for r in rows:
      group['last_step'][r] = max(int(group['last_step'][r]), t - 1)

  def emb_flush_window(self, groups, n_windows, hyper, lag=0, max_blocks=0):
    h = hyper.detach().cpu().numpy().reshape(-1)
    for group in groups:
      done = int(group['step_counter'].item()) - int(lag)
      chunk = -(-group['total_rows'] // n_windows)
      w = done % n_windows
      rows = range(w * chunk, min((w + 1) * chunk, group['total_rows']))
      self._replay(group, rows, done, h)
      for r in rows:
        group['last_step'][r] = max(int(group['last_step'][r]), done - 1)

  def emb_flush_decay(self, group, hyper):
    h = hyper.detach().cpu().numpy().reshape(-1)
    done = int(group['step_counter'].item())
    self._replay(group, range(group['total_rows']), done, h)
    group['last_step'].fill_(done - 1)

  # -- gradient clipping by global norm (compat/optimizers.py:365-376, 453-481)
  def gradsq_rows(self, x, cols, weight, acc, accumulate, counts=None, seg_stride=None):
    rows = x.shape[0]
    valid = np.ones(rows, dtype=bool)
    if counts is not None:
      stride = rows if seg_stride is None else int(seg_stride)
      r = np.arange(rows)
      seg = r // max(stride, 1)
      cnt = counts.cpu().numpy()
      valid = (seg < len(cnt)) & ((r - seg * stride) < cnt[np.minimum(seg, len(cnt) - 1)])
    xs = x.detach().cpu().numpy()[valid, :cols].astype(np.float32)
    s = F32(weight) * F32((xs * xs).sum(dtype=np.float32))
    acc[0] = float(F32(acc[0].item()) + s) if accumulate else float(s)

  def gradsq_dense(self, w, grad, l2coef, hyper, acc, accumulate=False):
    h = hyper.detach().cpu().numpy().reshape(-1)
    g = grad.detach().numpy().astype(np.float32) * F32(h[HYPER_GSCALE])
    if l2coef is not None:
      c = l2coef.numpy()
      g = np.where(c != 0, g + c * w.detach().numpy(), g).astype(np.float32)
    s = F32((g * g).sum(dtype=np.float32))
    acc[0] = float(F32(acc[0].item()) + s) if accumulate else float(s)

  def clip_scale(self, normsq, clip_norm, records, norm_out=None):
    norm = np.sqrt(F32(normsq[0].item()), dtype=np.float32)
    with np.errstate(divide='ignore'):
      scale = F32(clip_norm) * min(F32(1.0) / norm, F32(1.0) / F32(clip_norm))
    if not np.isfinite(norm):  # (TensorFlow propagates a non-finite norm into every clipped gradient)
      scale = F32(np.nan)
    records[:, HYPER_CLIP] = float(scale)
    if norm_out is not None:
      norm_out[0] = float(norm)

  def emb_apply_unique(self, group, keys, grads, n_unique, opt_kind, hyper):
    h = hyper.detach().cpu().numpy().reshape(-1)
    n = int(n_unique[0].item())
    ks, gs = keys.cpu().numpy(), grads.detach().cpu().numpy()
    sums = {int(ks[i]): gs[i, :group['dim']].astype(np.float32) for i in range(n) if ks[i] >= 0}
    var = group['var'].detach().numpy()
    m = None if group['m'] is None else group['m'].numpy()
    v = None if group['v'] is None else group['v'].numpy()
    if opt_kind == OPT_ADAM and 'last_step' in group:
      apply_sparse(var, m, v, sums, OPT_LAZY_ADAM, h)
      t = int(group['step_counter'].item()) - 1
      for key in sums:
        group['last_step'][key] = t
      return
    apply_sparse(var, m, v, sums, opt_kind, h)

  def dense_opt_step(self, w, m, v, grad, l2coef, opt_kind, hyper, l2_partials=None):
    h = hyper.detach().cpu().numpy().reshape(-1)
    dense_opt(w.detach().numpy(), None if m is None else m.numpy(), None if v is None else v.numpy(),
              grad.detach().numpy(), None if l2coef is None else l2coef.numpy(), opt_kind, h)
    if l2_partials is not None and l2coef is not None:
      self.l2_partials(w, l2coef, l2_partials)

"""-m gpu: the embedding-parallel path through the real HIP kernels.  gpurun exposes ONE MI355X, so W
ranks run as W threads on it with an in-process stand-in for the RCCL all-to-alls (tests/_sim_comm.py);
everything else - er_emb_route, er_gather_rows, the fused lookup over received rows,
er_emb_bwd_reduce_routed, owner-side er_emb_bwd_update, replicated small tables - is the product path."""
import logging
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from easyrec_amd import kernels  # noqa: E402
from easyrec_amd.input.criteo_synthetic import SyntheticCriteo  # noqa: E402
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator  # noqa: E402
from easyrec_amd.model.embedding_parallel import EmbeddingParallelEstimator  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402
from _sim_comm import SimWorld  # noqa: E402  (tests/ is on sys.path under pytest)

logging.disable(logging.WARNING)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'


def _cfg(name, lazy=False):
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', name))
  if lazy:
    oc = cfg.train_config.optimizer_config[0]
    oc.lazy_adam_optimizer.learning_rate.CopyFrom(oc.adam_optimizer.learning_rate)
  return cfg


def _worst(a, b):
  worst = ('', 0.0)
  for k in b:
    if k.endswith('/bias') or k.endswith('/bias/m') or k.endswith('/bias/v'):
      continue  # zero-gradient biases under BatchNorm: rounding noise through Adam
    if float(np.max(np.abs(b[k]))) < 1e-6:
      continue  # a shift in front of a BatchNorm (MultiTowerDIN's feature-BN beta) has a zero gradient too: noise only
    d = float(np.max(np.abs(a[k] - b[k]))) / (float(np.max(np.abs(b[k]))) + 1e-12)
    if d > worst[1]:
      worst = (k, d)
  return worst


@pytest.mark.parametrize('W,B,rows', [(4, 500, [1001, 37]), (4, 500, [20000003, 37]), (8, 4096, [100000, 3, 1]),
                                      (3, 9000, [1001, 37])])
def test_routing_kernels_against_numpy(W, B, rows):
  """er_emb_route / er_gather_rows / er_emb_bwd_reduce_routed on a group of a few tables: the per-lookup sort with
  32-bit composites, with 64-bit ones (a 20 M-row table), at the headline batch, and (9000 rows per lookup: above
  the per-lookup sort's limit) the device-wide radix sort."""
  hip = kernels.hip()
  rng = np.random.default_rng(3)
  dim = 16
  shard_rows = [(r + W - 1) // W for r in rows]
  T = len(rows)
  local_base = [int(x) for x in np.concatenate([[0], np.cumsum(shard_rows)[:-1]])]
  stride = sum(shard_rows)
  ids = [rng.integers(-1, r, size=B).astype(np.int64) for r in rows]
  ids[0][:100] = 5  # a hot id
  dout = torch.from_numpy((rng.standard_normal((B, T * dim)) * 0.1).astype(np.float32)).to(DEV)
  dummy = torch.zeros(T * B, dim, device=DEV)
  specs = [kernels.LookupSpec(table=dummy, ids=torch.from_numpy(ids[t]).to(DEV), offsets=None, weights=None, out=dout,
                              out_col=t * dim, rows=rows[t], key_base=0, dim=dim, combiner=0, n_rows=B, max_nnz=B)
           for t in range(T)]
  g = hip.emb_group_create(specs, dim, max(rows), dummy, None, None, None)
  hip.emb_group_set_routing(g, W, stride, local_base)
  n_ent = T * B
  ukeys = torch.zeros(n_ent, dtype=torch.int32, device=DEV)
  nu = torch.zeros(1, dtype=torch.int32, device=DEV)
  uidx = torch.zeros(n_ent, dtype=torch.int64, device=DEV)
  counts = torch.zeros(W, dtype=torch.int32, device=DEV)
  hip.emb_route(g, ukeys, nu, uidx, counts)
  ugrads = torch.zeros(n_ent, dim, device=DEV)
  hip.emb_bwd_reduce_routed(g, ugrads)
  torch.cuda.synchronize()
  # numpy restatement
  keys = np.full(n_ent, -1, dtype=np.int64)
  for t in range(T):
    ok = ids[t] >= 0
    keys[t * B:(t + 1) * B][ok] = (ids[t][ok] % W) * stride + local_base[t] + ids[t][ok] // W
  uniq = np.unique(keys[keys >= 0])
  n = int(nu.item())
  assert n == len(uniq)
  assert np.array_equal(ukeys[:n].cpu().numpy().astype(np.int64), uniq)
  exp_idx = np.where(keys >= 0, np.searchsorted(uniq, np.maximum(keys, 0)), -1)
  assert np.array_equal(uidx.cpu().numpy(), exp_idx)
  assert np.array_equal(counts.cpu().numpy(), [int(((uniq // stride) == w).sum()) for w in range(W)])
  exp = np.zeros((n, dim), dtype=np.float64)
  d = dout.cpu().numpy().astype(np.float64)
  for t in range(T):
    ok = keys[t * B:(t + 1) * B] >= 0
    np.add.at(exp, exp_idx[t * B:(t + 1) * B][ok], d[ok, t * dim:(t + 1) * dim])
  assert np.allclose(ugrads[:n].cpu().numpy(), exp, rtol=1e-5, atol=2e-5)  # fp32 sums of up to ~3000 terms of 0.1
  # owner side: rank 2 gathers the rows of the keys it owns
  table = torch.from_numpy(rng.standard_normal((stride, dim)).astype(np.float32)).to(DEV)
  mine = uniq[(uniq // stride) == 2]
  out = torch.zeros(len(mine) + 1, dim, device=DEV)
  hip.gather_rows(table, torch.from_numpy(mine.astype(np.int32)).to(DEV), len(mine), 2 * stride, out)
  torch.cuda.synchronize()
  assert torch.equal(out[:len(mine)], table[torch.from_numpy(mine - 2 * stride).to(DEV)])
  hip.emb_group_destroy(g)


@pytest.mark.parametrize('world,lazy,padded,config', [
    (1, False, True, 'deepfm_criteo_small.config'), (2, False, True, 'deepfm_criteo_small.config'),
    (4, True, True, 'deepfm_criteo_small.config'), (2, True, False, 'deepfm_criteo_small.config'),
    (3, False, False, 'deepfm_criteo_small.config'),
    # ONE table behind all 26 categorical features (the reference's own embedding-parallel Criteo config): the
    # device-wide sort + the fixed-capacity layout pass instead of the per-lookup sort
    (2, True, True, 'deepfm_shared_criteo_small.config'), (4, False, True, 'deepfm_shared_criteo_small.config'),
    # sequence + key lookups of MultiTowerDIN, ragged TagFeature lookups of MMoE through the same exchange
    (2, True, True, 'din_taobao_small.config'), (2, False, True, 'mmoe_taobao_small.config')])
def test_sharded_ranks_with_the_same_batch_equal_single_gpu(world, lazy, padded, config, monkeypatch):
  """Every rank sees the SAME batch: each embedding row gets world * g / world and each dense gradient
  the average of identical gradients, so the W-rank run must follow the single-GPU run (to the fp32
  noise of the GEMM library between runs)."""
  # padded: the fixed-capacity exchange (no host sync); else the compact one with host-side split sizes
  monkeypatch.setenv('EASYREC_AMD_PADDED_EXCHANGE', '1' if padded else '0')
  cfg = _cfg(config, lazy)
  B, steps = 128, 2
  ref = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=4).build()
  if 'criteo' in config:
    gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=9)
  else:
    from easyrec_amd.input.synthetic import SyntheticBatches
    gen = SyntheticBatches(cfg.data_config, ref.feature_configs, batch_size=B, seed=9)
  batches = [gen.next_batch() for _ in range(steps)]
  ref_losses = []
  for b in batches:
    ref.train_step(b)
    ref_losses.append(ref.loss_values())
  ref_state = ref.state_dict(slots=True)
  sim = SimWorld(world)

  def rank_fn(rank, comm):
    torch.cuda.set_device(0)
    est = EmbeddingParallelEstimator(cfg, device=DEV, batch_size=B, seed=4, rank=rank, world=world, comm=comm,
                                     replicate_bytes=1024).build()
    losses = []
    for b in batches:
      est.train_step(b)
      losses.append(est.loss_values())
    assert est.engine.padded == padded
    return est.state_dict(slots=True), losses, dict(est.engine.placement)

  results = sim.run(rank_fn)
  for state, losses, placement in results:
    assert any(p[0] == 'shard' for p in placement.values())
    assert 'criteo' not in config or any(p[0] == 'rep' for p in placement.values())
    for got, exp in zip(losses, ref_losses):
      assert abs(got['total_loss'] - exp['total_loss']) <= 2e-4 * abs(exp['total_loss']), (got, exp)
    first = {k: v for k, v in state.items() if k.endswith('/m')}
    k, d = _worst(first, {k: ref_state[k] for k in first})
    assert d < 5e-3, (k, d)
  # all ranks hold the same gathered tables
  for k in results[0][0]:
    if 'embedding_weights' in k:
      assert np.array_equal(results[0][0][k], results[-1][0][k]), k


@pytest.mark.parametrize('whole', [False, True])
def test_graph_segments_equal_eager_embedding_parallel_step(whole):
  """The EP step replayed as hipGraph segments (+ eager exchanges), or as ONE hipGraph with the exchanges inside
  (capture(whole=True)), must give the same bits as the all-eager EP step (every kernel on the path is deterministic)."""
  from easyrec_amd.core.comm import LocalComm
  cfg = _cfg('deepfm_criteo_small.config')
  B = 256
  gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=13)
  batches = [gen.next_batch() for _ in range(5)]
  ests = [EmbeddingParallelEstimator(cfg, device=DEV, batch_size=B, seed=6, rank=0, world=1, comm=LocalComm(),
                                     replicate_bytes=1024).build() for _ in range(2)]
  for e in ests:
    e.features.load(batches[0])
  for _ in range(3):
    ests[0].train_step()
  graphs = ests[1].capture(warmup=3, whole=whole)
  assert (ests[1]._whole_graph is not None) == whole and len(graphs) == (1 if whole else len(ests[1]._phases()))
  for b in batches[1:]:
    for e in ests:
      e.train_step(b)
    la, lb = ests[0].loss_values(), ests[1].loss_values()
    assert la == lb, (la, lb)
  sa, sb = ests[0].state_dict(slots=True), ests[1].state_dict(slots=True)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


@pytest.mark.parametrize('counts', [[700], [300, 0, 450, 1, 0], [2000, 1500, 1800, 1700, 1600, 1900, 1400, 1300]])
def test_owner_merge_and_serve_against_the_sorted_form(counts):
  """er_emb_owner_merge + er_emb_owner_serve + er_emb_bwd_update on runs of ascending duplicate-free keys (one per
  requester, some empty) against the sorted form (er_gather_rows, er_emb_bwd_update's own radix sort) on a copy -
  bit for bit: the merge yields the stable by-key order the sort yields.  Two groups (dim 16 and 1) share the merge."""
  hip = kernels.hip()
  rng = np.random.default_rng(len(counts))
  rows, cap = 5000, sum(counts) + 100
  runs = [np.sort(rng.choice(rows, size=c, replace=False)) for c in counts]
  ids = np.full(cap, -1, dtype=np.int64)
  m = sum(counts)
  ids[:m] = np.concatenate(runs)
  ids_dev = torch.from_numpy(ids).to(DEV)
  hyper = torch.zeros(16, dtype=torch.float32)
  hyper[kernels.HYPER_LR], hyper[kernels.HYPER_GSCALE] = 0.1, 0.5
  hyper = hyper.to(DEV)
  made = []
  for dim in (16, 1):
    table = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32)).to(DEV)
    grads = torch.from_numpy(rng.standard_normal((cap, dim)).astype(np.float32)).to(DEV)
    if dim == 16:  # numpy: a row asked for by several requesters moves by the sum of their gradients
      want = table.cpu().numpy().astype(np.float64)
      np.subtract.at(want, ids[:m], 0.1 * 0.5 * grads[:m].cpu().numpy().astype(np.float64))
    pair = []
    for _ in range(2):  # [merged form, sorted form]
      var = table.clone()
      spec = kernels.LookupSpec(table=var, ids=ids_dev, offsets=None, weights=None, out=grads, out_col=0, rows=rows,
                                key_base=0, dim=dim, combiner=0, n_rows=cap, max_nnz=cap)
      g = hip.emb_group_create([spec], dim, rows, var, None, None, None)
      hip.emb_group_set_active(g, m)
      pair.append((g, var))
    made.append(pair)
  (g16, v16), (s16, sv16) = made[0]
  (g1, v1), (s1, sv1) = made[1]
  assert hip.emb_group_share_sort(g1, g16)
  out16, out1 = torch.zeros(cap, 16, device=DEV), torch.zeros(cap, 1, device=DEV)
  hip.emb_owner_merge(g16, counts)
  hip.emb_owner_serve([g16, g1], [out16, out1], None)
  hip.emb_bwd_update_multi([g16, g1], kernels.OPT_SGD, hyper)
  exp16, exp1 = torch.zeros(cap, 16, device=DEV), torch.zeros(cap, 1, device=DEV)
  keys32 = ids_dev[:m].to(torch.int32)
  hip.gather_rows(sv16, keys32, m, 0, exp16)
  hip.gather_rows(sv1, keys32, m, 0, exp1)
  hip.emb_bwd_update(s16, kernels.OPT_SGD, hyper)
  hip.emb_bwd_update(s1, kernels.OPT_SGD, hyper)
  torch.cuda.synchronize()
  assert torch.equal(out16[:m], exp16[:m]) and torch.equal(out1[:m], exp1[:m])
  assert torch.equal(v16, sv16) and torch.equal(v1, sv1)
  assert np.allclose(v16.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
  for g, _ in made[0] + made[1]:
    hip.emb_group_destroy(g)


def test_owner_serve_catches_rows_up_in_closed_form():
  """er_emb_owner_serve on an owner whose rows carry 0 .. 500 pending decay-only steps (the keys of three requesters,
  some rows asked for by several): the closed-form replay (csrc/er_decay.h: per-launch table for backlogs <= K, the
  per-step table behind it) against the exact step-by-step replay on a copy - reply rows and var within 1e-6 of the
  table's scale, m within 1e-4 and v within 2e-4 relative, last_step identical."""
  hip = kernels.hip()
  rng = np.random.default_rng(5)
  rows, dim, T = 3000, 16, 520
  counts = [700, 300, 450]
  runs = [np.sort(rng.choice(rows, size=c, replace=False)) for c in counts]
  m_keys = sum(counts)
  cap = m_keys + 50
  ids = np.full(cap, -1, dtype=np.int64)
  ids[:m_keys] = np.concatenate(runs)
  ids_dev = torch.from_numpy(ids).to(DEV)
  hist_cap = T + 8
  hist = torch.zeros(2 * hist_cap, device=DEV)
  counter = torch.zeros(1, dtype=torch.int64, device=DEV)
  hyper = torch.zeros(kernels.HYPER_FLOATS, device=DEV)
  f = np.float32
  rows_h = np.zeros((T, kernels.HYPER_FLOATS), dtype=np.float32)
  for s_ in range(T):
    lr = f(1e-2 * 0.5 ** (s_ // 200))
    rows_h[s_, :8] = [lr, lr * np.sqrt(f(1) - f(0.999) ** (s_ + 1)) / (f(1) - f(0.9) ** (s_ + 1)), 0.9, 0.999, f(1) - f(0.9),
                      f(1) - f(0.999), 1e-8, 1.0]
  rows_h = torch.from_numpy(rows_h).to(DEV)
  tabs = hip.decay_tables_create(hist, counter, 0.9, 0.999)
  for s_ in range(T):  # steps 0 .. T-1 have "run": the history and the per-step table are complete up to here
    hip.step_prologue(rows_h, counter, hyper, history=hist, decay_tables=tabs)
  var0 = (rng.standard_normal((rows, dim)) * 0.05).astype(np.float32)
  g0 = (rng.standard_normal((rows, dim)) * 10.0 ** rng.integers(-7, -1, size=(rows, 1))).astype(np.float32)
  m0, v0 = (0.1 * g0).astype(np.float32), (1e-3 * g0 * g0).astype(np.float32)
  # the row's last update: T-2 (nothing pending when brought to step T-2... one step), down to T-2-500, a few never updated
  last0 = (T - 2 - rng.choice([0, 1, 2, 5, 30, 150, 191, 192, 193, 300, 500], size=rows)).astype(np.int32)
  never = rng.random(rows) < 0.02
  last0[never] = -1
  m0[never], v0[never] = 0, 0
  grads = torch.zeros(cap, dim, device=DEV)
  res = {}
  for mode in ('exact', 'closed'):
    var, m, v = (torch.from_numpy(x.copy()).to(DEV) for x in (var0, m0, v0))
    last = torch.from_numpy(last0.copy()).to(DEV)
    spec = kernels.LookupSpec(table=var, ids=ids_dev, offsets=None, weights=None, out=grads, out_col=0, rows=rows,
                              key_base=0, dim=dim, combiner=0, n_rows=cap, max_nnz=cap)
    g = hip.emb_group_create([spec], dim, rows, var, m, v, None)
    hip.emb_group_set_active(g, m_keys)
    hip.emb_group_enable_lazy_decay(g, last, hist, counter)
    if mode == 'closed':
      hip.emb_group_set_decay_tables(g, tabs)
    out = torch.zeros(cap, dim, device=DEV)
    hip.emb_owner_merge(g, counts)
    hip.emb_owner_serve([g], [out], rows_h[T - 1])  # counter == T: the rows are brought to step T - 2
    torch.cuda.synchronize()
    res[mode] = [t.cpu().numpy().astype(np.float64) for t in (out[:m_keys], var, m, v)] + [last.cpu().numpy()]
    hip.emb_group_destroy(g)
  hip.decay_tables_destroy(tabs)
  (oe, ve, me, se, le), (oc, vc, mc, sc, lc) = res['exact'], res['closed']
  assert np.array_equal(le, lc)
  asked = np.zeros(rows, dtype=bool)
  asked[ids[:m_keys]] = True
  assert np.array_equal(le[asked & (last0 >= 0) & (last0 < T - 2)], np.full(int((asked & (last0 >= 0) & (last0 < T - 2)).sum()), T - 2))
  scale = float(np.abs(ve).max())
  assert float(np.abs(oe - oc).max()) <= 1e-6 * scale and float(np.abs(ve - vc).max()) <= 1e-6 * scale
  assert np.array_equal(ve[~asked], var0[~asked].astype(np.float64)) and np.array_equal(vc[~asked], var0[~asked].astype(np.float64))
  big = np.abs(me) > 1e-30
  assert float((np.abs(me - mc)[big] / np.abs(me)[big]).max()) <= 1e-4
  big = se > 1e-35
  assert float((np.abs(se - sc)[big] / se[big]).max()) <= 2e-4
  moved = asked & (last0 >= 0) & (last0 < T - 2)
  assert float(np.abs(ve[moved] - var0[moved]).max()) > 1e-4, 'the case is meant to replay something'


@pytest.mark.parametrize('header', [True, False])
def test_fixed_capacity_route_and_owner_side_against_numpy(header):
  """er_emb_group_set_peer_capacity: owner w's keys at [w * C, ...) (behind their count with a header), the entry
  index and the reduced gradients in the same padded row layout; er_emb_owner_ids + er_emb_owner_merge_padded +
  er_emb_owner_serve on such a buffer (as if every peer had sent this rank's buffer) against numpy; an owner with more
  keys than C raises the overflow flag."""
  hip = kernels.hip()
  rng = np.random.default_rng(17)
  W, B, dim, rows = 4, 600, 16, [3001, 37, 1]
  T = len(rows)
  shard_rows = [(r + W - 1) // W for r in rows]
  local_base = [int(x) for x in np.concatenate([[0], np.cumsum(shard_rows)[:-1]])]
  stride = sum(shard_rows)
  ids = [rng.integers(-1, r, size=B).astype(np.int64) for r in rows]
  dout = torch.from_numpy((rng.standard_normal((B, T * dim)) * 0.1).astype(np.float32)).to(DEV)
  dummy = torch.zeros(T * B, dim, device=DEV)
  specs = [kernels.LookupSpec(table=dummy, ids=torch.from_numpy(ids[t]).to(DEV), offsets=None, weights=None, out=dout,
                              out_col=t * dim, rows=rows[t], key_base=0, dim=dim, combiner=0, n_rows=B, max_nnz=B)
           for t in range(T)]
  keys = np.full(T * B, -1, dtype=np.int64)
  for t in range(T):
    ok = ids[t] >= 0
    keys[t * B:(t + 1) * B][ok] = (ids[t][ok] % W) * stride + local_base[t] + ids[t][ok] // W
  uniq = np.unique(keys[keys >= 0])
  per_owner = [uniq[(uniq // stride) == w] for w in range(W)]
  C = max(len(p) for p in per_owner) + 5
  hdr = 1 if header else 0
  g = hip.emb_group_create(specs, dim, max(rows + [W * C]), dummy, None, None, None)
  hip.emb_group_set_routing(g, W, stride, local_base)
  hip.emb_group_set_peer_capacity(g, C, count_header=header)
  ukeys = torch.full((W * (C + hdr),), -7, dtype=torch.int32, device=DEV)
  nu = torch.zeros(1, dtype=torch.int32, device=DEV)
  uidx = torch.zeros(T * B, dtype=torch.int64, device=DEV)
  counts = torch.zeros(W, dtype=torch.int32, device=DEV)
  hip.emb_route(g, ukeys, nu, uidx, counts)
  ugrads = torch.zeros(W * C, dim, device=DEV)
  hip.emb_bwd_reduce_routed(g, ugrads)
  torch.cuda.synchronize()
  assert not hip.emb_route_overflow(g)
  uk = ukeys.cpu().numpy().astype(np.int64)
  slot = {}
  for w in range(W):
    seg = uk[w * (C + hdr):(w + 1) * (C + hdr)]
    if header:
      assert seg[0] == len(per_owner[w])
    assert np.array_equal(seg[hdr:hdr + len(per_owner[w])], per_owner[w])
    for r, k in enumerate(per_owner[w]):
      slot[int(k)] = w * C + r
  assert np.array_equal(counts.cpu().numpy(), [len(p) for p in per_owner]) and int(nu.item()) == len(uniq)
  exp_idx = np.array([slot[int(k)] if k >= 0 else -1 for k in keys])
  assert np.array_equal(uidx.cpu().numpy(), exp_idx)
  exp = np.zeros((W * C, dim), dtype=np.float64)
  d = dout.cpu().numpy().astype(np.float64)
  for t in range(T):
    ok = keys[t * B:(t + 1) * B] >= 0
    np.add.at(exp, exp_idx[t * B:(t + 1) * B][ok], d[ok, t * dim:(t + 1) * dim])
  got = ugrads.cpu().numpy()
  used = sorted(slot.values())
  assert np.allclose(got[used], exp[used], rtol=1e-5, atol=1e-6)
  # owner side: pretend every peer sent this buffer's segment of owner 2 (so every key arrives W times)
  me = 2
  seg = ukeys[me * (C + hdr):(me + 1) * (C + hdr)]
  recv = seg.repeat(W).contiguous()
  cnt_in = None if header else torch.full((W,), len(per_owner[me]), dtype=torch.int32, device=DEV)
  rids = torch.zeros(W * C, dtype=torch.int64, device=DEV)
  rcnt = torch.zeros(W, dtype=torch.int32, device=DEV)
  hip.emb_owner_ids(recv, cnt_in, W, C, me * stride, rids, rcnt)
  table = torch.from_numpy(rng.standard_normal((stride, dim)).astype(np.float32)).to(DEV)
  rgrads = torch.from_numpy(rng.standard_normal((W * C, dim)).astype(np.float32)).to(DEV)
  var = table.clone()
  ospec = kernels.LookupSpec(table=var, ids=rids, offsets=None, weights=None, out=rgrads, out_col=0, rows=stride, key_base=0,
                             dim=dim, combiner=0, n_rows=W * C, max_nnz=W * C)
  og = hip.emb_group_create([ospec], dim, stride, var, None, None, None)
  hip.emb_owner_merge_padded(og, rcnt, W, C)
  rows_out = torch.full((W * C, dim), 7.0, device=DEV)
  hip.emb_owner_serve([og], [rows_out], None)
  hyper = torch.zeros(16, dtype=torch.float32)
  hyper[kernels.HYPER_LR], hyper[kernels.HYPER_GSCALE] = 0.1, 1.0
  hip.emb_bwd_update(og, kernels.OPT_SGD, hyper.to(DEV))
  torch.cuda.synchronize()
  n_me = len(per_owner[me])
  assert np.array_equal(rcnt.cpu().numpy(), [n_me] * W)
  local = per_owner[me] - me * stride
  exp_ids = np.full((W, C), -1, dtype=np.int64)
  exp_ids[:, :n_me] = local
  assert np.array_equal(rids.cpu().numpy().reshape(W, C), exp_ids)
  ro = rows_out.cpu().numpy().reshape(W, C, dim)
  assert np.array_equal(ro[:, :n_me], np.broadcast_to(table.cpu().numpy()[local], (W, n_me, dim)))
  assert np.all(ro[:, n_me:] == 7.0)  # padding slots are not written
  want = table.cpu().numpy().astype(np.float64)
  gsum = rgrads.cpu().numpy().astype(np.float64).reshape(W, C, dim)[:, :n_me].sum(axis=0)
  want[local] -= 0.1 * gsum
  assert np.allclose(var.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
  # overflow: a capacity below the largest owner's count
  hip.emb_group_set_peer_capacity(g, max(len(p) for p in per_owner) - 1, count_header=header)
  hip.emb_route(g, ukeys, nu, uidx, counts)
  torch.cuda.synchronize()
  assert hip.emb_route_overflow(g)
  hip.emb_group_destroy(g)
  hip.emb_group_destroy(og)


@pytest.mark.parametrize('overlap', ['0', '1'])
def test_lazy_decay_equals_sweep_through_two_sharded_ranks(monkeypatch, overlap):
  """overlap '1': the owners' rolling flush on a second stream next to the compute phase (lag 1).  The EXACT replay
  (EASYREC_AMD_EXACT_DECAY=1); the default closed-form replay of the owners is held to it by
  test_owner_serve_catches_rows_up_in_closed_form (two whole-model runs that differ in rounding cannot be: with two ranks'
  averaged gradients this config's first Adam steps are sign-like on many elements and the runs are 1e-3 of a table's
  scale apart after three steps, tools/dbg_ep_closed.py - while two closed-form runs are bit-identical to each other).
  The model-level lazy-dense-decay == sweep check of tests/test_deepfm_gpu.py through EmbeddingParallelEstimator,
  W = 2 ranks as threads with their own batches: the owner side's er_emb_owner_serve catches rows up, the owner's
  er_emb_bwd_update_multi stamps them, er_emb_flush_decay finishes - against the same two ranks streaming every row."""
  from test_deepfm_gpu import _assert_lazy_equals_sweep, _idle_schedule
  monkeypatch.setenv('EASYREC_AMD_EXACT_DECAY', '1')
  monkeypatch.setenv('EASYREC_AMD_OVERLAP_FLUSH', overlap)
  cfg = _cfg('deepfm_criteo_small.config')
  B, world = 64, 2
  feats = list(cfg.feature_config.features)
  scheds = [_idle_schedule(cfg, feats, B, 1000)]
  gen = SyntheticCriteo(cfg.data_config, feats, batch_size=B, seed=77)
  scheds.append([gen.next_batch() for _ in range(len(scheds[0]))])  # rank 1: fresh batches throughout
  states = {}
  for sweep in (False, True):
    sim = SimWorld(world)

    def rank_fn(rank, comm):
      torch.cuda.set_device(0)
      est = EmbeddingParallelEstimator(cfg, device=DEV, batch_size=B, seed=4, rank=rank, world=world, comm=comm,
                                       replicate_bytes=1024, dense_sweep=sweep).build()
      for b in scheds[rank]:
        est.train_step(b)
      return est.state_dict(slots=True)

    states[sweep] = sim.run(rank_fn)[0]
  _assert_lazy_equals_sweep(states[False], states[True])


# ---- hash-table (ev_params) tables under embedding parallelism --------------------------------------------------------
@pytest.mark.parametrize('W', [1, 2, 8])
def test_kv_bucket_unbucket_kernels(W):
  """er_kv_bucket / er_kv_unbucket: every id lands once in its owner's block (id % W) inside its job's region, padding
  and the stale tail of a ragged list get no slot, and the way back writes arena_row * W + owner (or -1)."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(W)
  jobs, limit = [], torch.tensor([700], dtype=torch.int32, device=DEV)
  for j, n in enumerate((1000, 4096, 900)):
    ids = torch.randint(0, 1 << 40, (n,), generator=g, dtype=torch.int64)
    ids[::13] = -1
    jobs.append((ids.to(DEV), torch.full((n,), -7, dtype=torch.int64, device=DEV)) + ((limit,) if j == 2 else ()))
  h = hip.kv_route_create(jobs, W)
  hip.kv_bucket(h)
  torch.cuda.synchronize()
  send, C = h['send'].cpu(), h['C']
  for j, job in enumerate(jobs):
    ids, slot = job[0].cpu(), h['slots'][j].cpu().to(torch.int64)
    valid = ids >= 0
    if j == 2:
      valid &= torch.arange(ids.numel()) < 700
    assert bool((slot[~valid] == -1).all()) and bool((slot[valid] >= 0).all())
    at = slot[valid]
    assert torch.equal(send.view(-1)[at], ids[valid]) and at.unique().numel() == at.numel()
    assert torch.equal(at // C, ids[valid] % W)
    off = at % C
    assert bool((off >= h['offs'][j]).all()) and bool((off < h['offs'][j + 1]).all())
  assert int((send >= 0).sum()) == sum(int((h['slots'][j] >= 0).sum()) for j in range(3))
  # the owners' answer: row = (id >> 3) for even ids, none for odd ones
  back = torch.where((send >= 0) & (send % 2 == 0), send >> 3, torch.full_like(send, -1))
  h['back'].copy_(back.to(DEV))
  hip.kv_unbucket(h)
  torch.cuda.synchronize()
  for j, job in enumerate(jobs):
    ids, out = job[0].cpu(), job[1].cpu()
    valid = (ids >= 0) & (ids % 2 == 0)
    if j == 2:
      valid &= torch.arange(ids.numel()) < 700
    assert torch.equal(out[valid], (ids[valid] >> 3) * W + ids[valid] % W) and bool((out[~valid] == -1).all())


@pytest.mark.parametrize('world', [1, 2])
def test_sharded_kv_tables_with_the_same_batch_equal_single_gpu(world):
  """Hash-table tables sharded by id % world (ids to their owners and back around the route): with the SAME batch on
  every rank each row gets world * g / world, so the run follows the single-GPU engine - losses, the ids that have a
  row, the rows."""
  cfg = _cfg('deepfm_kv_criteo_small.config')
  B, steps = 128, 2
  ref = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=4).build()
  gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=9)
  batches = [gen.next_batch() for _ in range(steps)]
  ref_losses = []
  for b in batches:
    ref.train_step(b)
    ref_losses.append(ref.loss_values())
  ref_state = ref.state_dict(slots=True)
  names = sorted(ref.engine.kv_tables)
  sim = SimWorld(world)

  def rank_fn(rank, comm):
    torch.cuda.set_device(0)
    est = EmbeddingParallelEstimator(cfg, device=DEV, batch_size=B, seed=4, rank=rank, world=world, comm=comm,
                                     replicate_bytes=1024).build()
    losses = []
    for b in batches:
      est.train_step(b)
      losses.append(est.loss_values())
    owned = {n: int(est.engine.kv_tables[n]['next_row'].item()) for n in names}
    return est.state_dict(slots=True), losses, owned

  results = sim.run(rank_fn)
  for state, losses, owned in results:
    for got, exp in zip(losses, ref_losses):
      assert abs(got['total_loss'] - exp['total_loss']) <= 2e-4 * abs(exp['total_loss']), (got, exp)
    for n in names:
      assert np.array_equal(state[n + '/keys'], ref_state[n + '/keys']), n
      assert world == 1 or 0 < owned[n] < state[n + '/keys'].size, (n, owned[n])
      m_scale = float(np.abs(ref_state[n + '/m']).max())
      assert float(np.abs(state[n + '/m'] - ref_state[n + '/m']).max()) <= 5e-3 * m_scale, n
      assert float(np.abs(state[n] - ref_state[n]).max()) <= 4e-3, n


@pytest.mark.parametrize('world', [1, 2, 4])
def test_merged_requester_tail_changes_no_bit(world, monkeypatch):
  """The embedding-parallel step with the requester's tail merged (er_emb_reduce_local_tail: weight gradients + every local
  reduction + loss tail in two launches, round 6) against the launches apart (one per reduction kind and dim group, the
  grouped weight-gradient launch and the loss tail on their own), W = 1, 2, 4 ranks as threads on different batches: with
  the contraction's k-splits those of the stand-alone grouped launch (tail_wgrad_blocks = 0) every loss, table, slot and
  dense variable is bit-identical over three steps - same bodies, same order."""
  cfg = _cfg('deepfm_criteo_small.config')
  B, steps = 128, 3
  gens = [SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=20 + r) for r in range(world)]
  batches = [[g.next_batch() for _ in range(steps)] for g in gens]
  monkeypatch.setattr(kernels.HipBackend, 'tail_wgrad_blocks', 0)

  def run(merged):
    monkeypatch.setattr(kernels.HipBackend, 'ep_merged_reduce', merged)
    sim = SimWorld(world)

    def rank_fn(rank, comm):
      torch.cuda.set_device(0)
      est = EmbeddingParallelEstimator(cfg, device=DEV, batch_size=B, seed=4, rank=rank, world=world, comm=comm,
                                       replicate_bytes=1024).build()
      assert est.merged_reduce == merged
      losses = []
      for b in batches[rank]:
        est.train_step(b)
        losses.append(est.loss_values())
      return est.state_dict(slots=True), losses

    return sim.run(rank_fn)

  a, b = run(True), run(False)
  for (sa, la), (sb, lb) in zip(a, b):
    assert la == lb, (la, lb)
    for k in sb:
      assert np.array_equal(sa[k], sb[k]), k


@pytest.mark.parametrize('switch', ['ep_update_tail', 'ep_owner_fused'])
@pytest.mark.parametrize('world,config', [(1, 'deepfm_criteo_small.config'), (2, 'deepfm_criteo_small.config'),
                                          (4, 'deepfm_criteo_small.config'), (2, 'mmoe_taobao_small.config')])
def test_merged_owner_launches_change_no_bit(world, config, switch, monkeypatch):
  """ep_update_tail: the end of the embedding-parallel step as two launches (er_emb_owner_update_tail: the owners' row update
  whose cross-tile fix launch also carries the replicated tables' apply and the dense optimizer) against the four launches
  apart (er_emb_bwd_update_multi, er_emb_dense_apply, er_dense_opt_step_l2).  ep_owner_fused: owner ids + entry build + merge
  + the serve launch's lag-1 replay table as one launch (er_emb_owner_ids_merge) against the four apart.  Ranks as threads
  on different batches, every loss, table, slot and dense variable bit-identical over three steps."""
  cfg = _cfg(config)
  B, steps = 128, 3
  if 'criteo' in config:
    gens = [SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=30 + r) for r in range(world)]
  else:
    from easyrec_amd.input.synthetic import SyntheticBatches
    from easyrec_amd.utils import config_util
    fcs = config_util.get_compatible_feature_configs(cfg)
    gens = [SyntheticBatches(cfg.data_config, fcs, batch_size=B, seed=30 + r) for r in range(world)]
  batches = [[g.next_batch() for _ in range(steps)] for g in gens]

  def run(merged):
    monkeypatch.setattr(kernels.HipBackend, switch, merged)
    sim = SimWorld(world)

    def rank_fn(rank, comm):
      torch.cuda.set_device(0)
      est = EmbeddingParallelEstimator(cfg, device=DEV, batch_size=B, seed=4, rank=rank, world=world, comm=comm,
                                       replicate_bytes=1024).build()
      losses = []
      for b in batches[rank]:
        est.train_step(b)
        losses.append(est.loss_values())
      return est.state_dict(slots=True), losses

    return sim.run(rank_fn)

  a, b = run(True), run(False)
  for (sa, la), (sb, lb) in zip(a, b):
    assert la == lb, (la, lb)
    for k in sb:
      assert np.array_equal(sa[k], sb[k]), k


@pytest.mark.parametrize('counts', [[700], [300, 0, 450, 1, 0], [64, 64, 64, 64], [2000, 1500, 1800, 1700, 1600, 1900, 1400, 1300]])
def test_owner_ids_merge_equals_the_launches_apart(counts):
  """er_emb_owner_ids_merge (owner ids + entry build + merge in one launch, the merge searching the RECEIVED runs) against
  er_emb_owner_ids + er_emb_owner_merge_padded on twin owner groups: runs of different lengths (empty, one key, filled to
  the capacity), keys shared between runs, one id outside the table - the ids, the counts, the rows the serve launch
  replies and the table after an SGD row update are bit-identical."""
  hip = kernels.hip()
  rng = np.random.default_rng(sum(counts))
  W, dim, stride = len(counts), 16, 5000
  C = max(counts)
  me = W // 2
  recv = np.zeros((W, C + 1), dtype=np.int64)
  for q, c in enumerate(counts):
    ks = np.sort(rng.choice(stride, size=c, replace=False)) + me * stride
    recv[q, 0] = c
    recv[q, 1:1 + c] = ks
    recv[q, 1 + c:] = rng.integers(0, 1 << 20, size=C - c)  # (stale padding)
  if counts[0] > 2:
    recv[0, counts[0]] = me * stride + stride + 3  # an id past the table: its entry carries the invalid key
  recv_t = torch.from_numpy(recv.astype(np.int32).reshape(-1)).to(DEV)
  table = torch.from_numpy(rng.standard_normal((stride, dim)).astype(np.float32)).to(DEV)
  rgrads = torch.from_numpy(rng.standard_normal((W * C, dim)).astype(np.float32)).to(DEV)
  hyper = torch.zeros(16, dtype=torch.float32)
  hyper[kernels.HYPER_LR], hyper[kernels.HYPER_GSCALE] = 0.1, 1.0
  hyper = hyper.to(DEV)
  res = []
  for fused in (False, True):
    rids = torch.zeros(W * C, dtype=torch.int64, device=DEV)
    rcnt = torch.zeros(W, dtype=torch.int32, device=DEV)
    var = table.clone()
    ospec = kernels.LookupSpec(table=var, ids=rids, offsets=None, weights=None, out=rgrads, out_col=0, rows=stride, key_base=0,
                               dim=dim, combiner=0, n_rows=W * C, max_nnz=W * C)
    og = hip.emb_group_create([ospec], dim, stride, var, None, None, None)
    if fused:
      hip.emb_owner_ids_merge(og, recv_t, W, C, me * stride, rids, rcnt, False)
    else:
      hip.emb_owner_ids(recv_t, None, W, C, me * stride, rids, rcnt)
      hip.emb_owner_merge_padded(og, rcnt, W, C)
    rows_out = torch.full((W * C, dim), 7.0, device=DEV)
    hip.emb_owner_serve([og], [rows_out], None)
    hip.emb_bwd_update(og, kernels.OPT_SGD, hyper)
    torch.cuda.synchronize()
    res.append((rids.cpu().numpy(), rcnt.cpu().numpy(), rows_out.cpu().numpy(), var.cpu().numpy()))
    hip.emb_group_destroy(og)
  assert np.array_equal(res[0][1], counts)
  for a, b, what in zip(res[0], res[1], ('ids', 'counts', 'served rows', 'table after the update')):
    assert np.array_equal(a, b), what
  assert not np.array_equal(res[0][3], table.cpu().numpy())

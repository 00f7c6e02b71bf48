"""-m gpu: every HIP kernel, called through the C ABI (easyrec_amd.kernels.HipBackend -> ctypes ->
libeasyrec_hip.so), against the CPU oracle (oracle/kernel_ref.py) on the same seeded inputs.

Bars: bit-exact for integer/id work (hashing) and for the embedding forward (pure copies and fp32
sums in a defined order); fp32 tolerances (stated per test) where the summation order differs.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from easyrec_amd import kernels  # noqa: E402
from oracle import hashing, kernel_ref  # noqa: E402

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
  assert torch.cuda.is_available(), 'gpu tests need an MI355X'
  return kernels.hip()


@pytest.fixture(scope='module')
def ref():
  return kernel_ref.RefBackend()


def _hyper(lr=1e-3, t=1, b1=0.9, b2=0.999, gscale=1.0):
  f = np.float32
  row = np.zeros(kernels.HYPER_FLOATS, dtype=np.float32)
  row[0] = f(lr)
  row[1] = f(lr) * np.sqrt(f(1) - f(b2)**t) / (f(1) - f(b1)**t)
  row[2], row[3], row[4], row[5], row[6], row[7] = f(b1), f(b2), f(1) - f(b1), f(1) - f(b2), f(1e-8), f(gscale)
  return torch.from_numpy(row)


# ------------------------------------------------------------------------------------------- K1
def _random_strings(rng, n, maxlen):
  return [bytes(rng.integers(0, 256, size=int(rng.integers(0, maxlen + 1)), dtype=np.uint8)) for _ in range(n)]


def test_hash_device_bit_exact(hip):
  from easyrec_amd.input.input import pack_strings
  rng = np.random.default_rng(0)
  strs = _random_strings(rng, 4000, 16) + _random_strings(rng, 2000, 300) + [b'', b'a', b'Hello', b'x' * 64,
                                                                             b'y' * 65, b'z' * 129]
  n = len(strs) // 3 * 3
  strs = strs[:n]
  data, offs = pack_strings(strs)
  nb = np.array([1000000, 3, 2**40 + 7], dtype=np.uint64)
  for drop in (0, 1):
    exp = hashing.hash_bucket_fast(data, offs, n // 3, nb, drop)
    got = hip.hash_bucket_fast(torch.from_numpy(data).to(DEV), torch.from_numpy(offs).to(DEV), n // 3,
                               torch.from_numpy(nb.astype(np.int64)).to(DEV), drop).cpu().numpy()
    assert np.array_equal(got, exp)


def test_hash_tf_vectors_on_device(hip):
  from easyrec_amd.input.input import pack_strings
  data, offs = pack_strings([b'a', b'b', b'c', b'd'])
  nb = torch.tensor([10], dtype=torch.int64, device=DEV)
  got = hip.hash_bucket_fast(torch.from_numpy(data).to(DEV), torch.from_numpy(offs).to(DEV), 4, nb, 0)
  assert got.cpu().tolist() == [9, 2, 2, 5]


def test_hash_int64_as_string(hip):
  vals = np.array([0, 7, -1, 123456789, -987654321, 2**62, -2**63 + 1, 42] * 4, dtype=np.int64)
  nb = np.array([1000003], dtype=np.uint64)
  got = hip.hash_bucket_fast_int64(torch.from_numpy(vals).to(DEV), len(vals),
                                   torch.from_numpy(nb.astype(np.int64)).to(DEV)).cpu().numpy()
  exp = np.array([hashing.fingerprint64(str(int(v))) % 1000003 for v in vals])
  assert np.array_equal(got, exp)


# ------------------------------------------------------------------------------------------- K2-K4
def _make_lookup_problem(rng, B, dims, device, ragged=True):
  """A model-like mix: dense single-id lookups (with missing ids), weighted dense, ragged
  sum/mean/sqrtn with and without weights, different dims, a shared table."""
  specs_cpu, specs_dev, tables = [], [], {}
  outs = {}

  def table(dim, rows):
    key = (dim, len([k for k in tables if k[0] == dim]))
    t = (rng.standard_normal((rows, dim)) * 0.1).astype(np.float32)
    tables[key] = t
    return key

  layout = []
  for dim in dims:
    layout.append((dim, 'dense', None))
    layout.append((dim, 'dense_w', None))
    if ragged:
      layout.append((dim, 'ragged', 'sum'))
      layout.append((dim, 'ragged_w', 'mean'))
      layout.append((dim, 'ragged_w', 'sqrtn'))
      layout.append((dim, 'ragged', 'mean'))
  width = sum(d for d, _, _ in layout)
  out_cpu = torch.zeros(B, width)
  out_dev = torch.zeros(B, width, device=device)
  # tables of one dim back to back
  group_rows = {}
  metas = []
  col = 0
  for (dim, kind, comb) in layout:
    rows = int(rng.integers(5, 60))
    base = group_rows.get(dim, 0)
    group_rows[dim] = base + rows
    metas.append((dim, kind, comb, rows, base, col))
    col += dim
  storage_cpu = {d: torch.from_numpy((rng.standard_normal((n, d)) * 0.1).astype(np.float32))
                 for d, n in group_rows.items()}
  storage_dev = {d: t.to(device) for d, t in storage_cpu.items()}
  for (dim, kind, comb, rows, base, col) in metas:
    if kind.startswith('dense'):
      ids = rng.integers(-1, rows, size=B).astype(np.int64)  # -1 = missing
      ids[rng.random(B) < 0.1] = rows + 3  # out of range -> pruned
      offsets, w = None, None
      if kind == 'dense_w':
        w = rng.standard_normal(B).astype(np.float32)
      n_rows, max_nnz, combiner = B, B, 0
    else:
      lens = rng.integers(0, 5, size=B)
      lens[rng.random(B) < 0.2] = 0
      offsets = np.zeros(B + 1, dtype=np.int32)
      offsets[1:] = np.cumsum(lens)
      nnz = int(offsets[-1])
      cap = nnz + 7
      ids = np.full(cap, -1, dtype=np.int64)
      ids[:nnz] = rng.integers(-1, rows, size=nnz)
      w = None
      if kind == 'ragged_w':
        w = np.zeros(cap, dtype=np.float32)
        w[:nnz] = rng.standard_normal(nnz).astype(np.float32)  # negatives get pruned for mean/sqrtn
      n_rows, max_nnz, combiner = B, cap, kernels.COMBINERS[comb]
    for store, out, lst, dv in ((storage_cpu, out_cpu, specs_cpu, 'cpu'), (storage_dev, out_dev, specs_dev, device)):
      lst.append(
          kernels.LookupSpec(
              table=store[dim][base:base + rows], ids=torch.from_numpy(ids).to(dv),
              offsets=None if offsets is None else torch.from_numpy(offsets).to(dv),
              weights=None if w is None else torch.from_numpy(w).to(dv), out=out, out_col=col, rows=rows,
              key_base=base, dim=dim, combiner=combiner, n_rows=n_rows, max_nnz=max_nnz))
  return specs_cpu, specs_dev, storage_cpu, storage_dev, out_cpu, out_dev, group_rows


@pytest.mark.parametrize('B,dims', [(257, (16, 1)), (64, (4, 8, 64, 3)), (1, (16,)), (1000, (32, 2))])
def test_emb_fwd_bit_exact(hip, ref, B, dims):
  rng = np.random.default_rng(B)
  sc, sd, _, _, oc, od, _ = _make_lookup_problem(rng, B, dims, DEV)
  pc, pd = ref.emb_plan_create(sc), hip.emb_plan_create(sd)
  assert pc['num_blocks'] == pd['num_blocks']
  ssq_c = torch.zeros(pc['num_blocks'])
  ssq_d = torch.zeros(pd['num_blocks'], device=DEV)
  # the oracle's lookup_rows (sequential, op by op) is the bit-exact reference
  for s in sc:
    res = kernel_ref.lookup_rows(s.table.numpy(), s.ids.numpy(), None if s.offsets is None else s.offsets.numpy(),
                                 None if s.weights is None else s.weights.numpy(), s.combiner, s.n_rows)
    oc[:, s.out_col:s.out_col + s.dim] = torch.from_numpy(res)
  hip.emb_fwd(pd, ssq_d)
  torch.cuda.synchronize()
  assert torch.equal(od.cpu(), oc), 'embedding forward must be bit-exact'
  ref.emb_fwd(pc, ssq_c)
  assert torch.allclose(ssq_d.cpu().sum(), ssq_c.sum(), rtol=1e-5)
  hip.emb_plan_destroy(pd)


@pytest.mark.parametrize('opt', [kernels.OPT_SGD, kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM, kernels.OPT_ADAGRAD])
@pytest.mark.parametrize('B,dims', [(300, (16, 1)), (50, (8, 3))])
def test_emb_bwd_update(hip, ref, opt, B, dims):
  """sort + segmented reduce + row-wise optimizer (+ dense-decay sweep for ER_OPT_ADAM).
  Runs that fit one 32-entry chunk are summed in source order (bit-exact vs the oracle); longer
  runs are combined piece-wise -> 1e-6 relative."""
  rng = np.random.default_rng(B + opt)
  sc, sd, stc, std, oc, od, group_rows = _make_lookup_problem(rng, B, dims, DEV)
  dout = (rng.standard_normal(tuple(oc.shape)) * 0.01).astype(np.float32)
  dc, dd = torch.from_numpy(dout), torch.from_numpy(dout).to(DEV)
  hyper = _hyper(lr=0.05, t=3, gscale=0.5)
  for dim, total in group_rows.items():
    mc = torch.from_numpy((rng.random((total, dim)) * 0.01).astype(np.float32))
    vc = torch.from_numpy((rng.random((total, dim)) * 0.01 + 1e-4).astype(np.float32))
    md, vd = mc.to(DEV), vc.to(DEV)
    var_c, var_d = stc[dim], std[dim]
    bitmap = torch.zeros((total + 31) // 32, dtype=torch.int32, device=DEV)
    gc = ref.emb_group_create([s.with_out(dc) for s in sc if s.dim == dim], dim, total, var_c, mc, vc, None)
    gd = hip.emb_group_create([s.with_out(dd) for s in sd if s.dim == dim], dim, total, var_d, md, vd, bitmap)
    assert gd['num_entries'] == gc['num_entries']
    for _ in range(2):  # twice: the bitmap must come back clean
      ref.emb_bwd_update(gc, opt, hyper)
      hip.emb_bwd_update(gd, opt, hyper.to(DEV))
    torch.cuda.synchronize()
    assert int(bitmap.abs().sum()) == 0
    for a, b, what in ((var_d, var_c, 'var'), (md, mc, 'm'), (vd, vc, 'v')):
      assert torch.allclose(a.cpu(), b, rtol=1e-5, atol=1e-8), (what, dim, opt, float((a.cpu() - b).abs().max()))
    hip.emb_group_destroy(gd)


def test_emb_bwd_long_runs_and_reduce(hip, ref):
  """Hot ids: runs of thousands of equal keys cross many 32-entry chunks."""
  rng = np.random.default_rng(5)
  B, dim, rows = 5000, 16, 7
  ids = rng.integers(0, rows, size=B).astype(np.int64)
  ids[:3000] = 2  # one very hot row
  table = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32))
  dout = torch.from_numpy((rng.standard_normal((B, dim)) * 0.01).astype(np.float32))

  def spec(dev):
    return kernels.LookupSpec(table=table.to(dev), ids=torch.from_numpy(ids).to(dev), offsets=None, weights=None,
                              out=dout.to(dev), out_col=0, rows=rows, key_base=0, dim=dim, combiner=0, n_rows=B,
                              max_nnz=B)

  gd = hip.emb_group_create([spec(DEV)], dim, rows, table.to(DEV), None, None, None)
  keys, grads, n = hip.emb_bwd_reduce(gd)
  torch.cuda.synchronize()
  n = int(n.item())
  assert n == len(np.unique(ids))
  exp = np.zeros((rows, dim), dtype=np.float64)
  np.add.at(exp, ids, dout.numpy().astype(np.float64))
  got_keys = keys[:n].cpu().numpy()
  assert np.array_equal(got_keys, np.unique(ids))
  assert np.allclose(grads[:n].cpu().numpy(), exp[got_keys], rtol=1e-5, atol=1e-7)
  # determinism: same bits on a second run
  keys2, grads2, _ = hip.emb_bwd_reduce(gd)
  torch.cuda.synchronize()
  assert torch.equal(grads[:n], grads2[:n])
  hip.emb_group_destroy(gd)


def test_overlapped_sweep_equals_sequential_adam(hip):
  """TF-exact Adam split over two streams (mark touched rows -> side stream sweeps the others while
  the main stream updates the touched rows) must give the same bits as the sequential ER_OPT_ADAM."""
  rng = np.random.default_rng(77)
  B, dims = 300, (16, 1)
  _, sd, _, std, _, od, group_rows = _make_lookup_problem(rng, B, dims, DEV)
  dd = torch.from_numpy((rng.standard_normal(tuple(od.shape)) * 0.01).astype(np.float32)).to(DEV)
  hyper = _hyper(lr=0.05, t=3, gscale=0.5).to(DEV)
  side = torch.cuda.Stream()
  for dim, total in group_rows.items():
    m0 = torch.from_numpy((rng.random((total, dim)) * 0.01).astype(np.float32)).to(DEV)
    v0 = torch.from_numpy((rng.random((total, dim)) * 0.01 + 1e-4).astype(np.float32)).to(DEV)
    res = []
    for mode in ('sequential', 'overlapped'):
      var, m, v = std[dim].clone(), m0.clone(), v0.clone()
      bitmap = torch.zeros((total + 31) // 32, dtype=torch.int32, device=DEV)
      specs = [kernels.LookupSpec(table=var[s.key_base:s.key_base + s.rows], ids=s.ids, offsets=s.offsets,
                                  weights=s.weights, out=dd, out_col=s.out_col, rows=s.rows, key_base=s.key_base,
                                  dim=s.dim, combiner=s.combiner, n_rows=s.n_rows, max_nnz=s.max_nnz)
               for s in sd if s.dim == dim]
      g = hip.emb_group_create(specs, dim, total, var, m, v, bitmap)
      for _ in range(3):
        if mode == 'sequential':
          hip.emb_bwd_update(g, kernels.OPT_ADAM, hyper)
        else:
          hip.emb_mark_touched(g)
          side.wait_stream(torch.cuda.current_stream())
          with torch.cuda.stream(side):
            hip.emb_sweep_untouched(g, hyper)
          hip.emb_bwd_update(g, kernels.OPT_LAZY_ADAM, hyper)
          torch.cuda.current_stream().wait_stream(side)
      torch.cuda.synchronize()
      assert int(bitmap.abs().sum()) == 0
      res.append((var, m, v))
      hip.emb_group_destroy(g)
    for a, b in zip(res[0], res[1]):
      assert torch.equal(a, b)


def test_stream_copy(hip):
  src = torch.randn(1 << 20, device=DEV)
  dst = torch.zeros_like(src)
  hip.stream_copy(src, dst)
  torch.cuda.synchronize()
  assert torch.equal(src, dst)


@pytest.mark.parametrize('dim,rows', [(16, 100003), (1, 50001), (3, 1001), (64, 4097)])
def test_adam_decay_sweep_bit_exact(hip, ref, dim, rows):
  rng = np.random.default_rng(dim)
  var = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32))
  m = torch.from_numpy((rng.standard_normal((rows, dim)) * 0.01).astype(np.float32))
  v = torch.from_numpy((rng.random((rows, dim)) * 0.01).astype(np.float32))
  bits = rng.integers(0, 2**31, size=(rows + 31) // 32).astype(np.int32)
  hyper = _hyper(t=5)
  vd, md, sd = var.to(DEV), m.to(DEV), v.to(DEV)
  hip.adam_decay_sweep(vd, md, sd, torch.from_numpy(bits).to(DEV), rows, dim, hyper.to(DEV))
  ref.adam_decay_sweep(var, m, v, torch.from_numpy(bits), rows, dim, hyper)
  torch.cuda.synchronize()
  assert torch.equal(md.cpu(), m) and torch.equal(sd.cpu(), v)
  assert torch.equal(vd.cpu(), var)


# ------------------------------------------------------------------------------------------- K5
@pytest.mark.parametrize('B,F,D', [(513, 39, 16), (7, 3, 5), (64, 8, 64)])
def test_fm_and_rowsum(hip, ref, B, F, D):
  rng = np.random.default_rng(F)
  x = torch.from_numpy(rng.standard_normal((B, F * D)).astype(np.float32))
  g = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
  fm_d, S_d = hip.fm_fwd(x.to(DEV), F, D)
  fm_c, S_c = ref.fm_fwd(x, F, D)
  # 0.5*(S^2 - sum e^2) cancels: tolerance relative to the magnitude of S^2
  assert float((fm_d.cpu() - fm_c).abs().max()) <= 2e-6 * float((S_c * S_c).abs().max()) + 1e-6
  assert torch.allclose(S_d.cpu(), S_c, rtol=1e-5, atol=1e-5)
  dx_d = hip.fm_bwd(x.to(DEV), S_d, g.to(DEV), F, D)
  dx_c = ref.fm_bwd(x, S_c, g, F, D)
  assert torch.allclose(dx_d.cpu(), dx_c, rtol=1e-4, atol=1e-4)
  rs = hip.rowsum_fwd(x.to(DEV), F * D)
  assert torch.allclose(rs.cpu(), x.sum(dim=1, keepdim=True), rtol=1e-4, atol=1e-4)
  gb = hip.rowsum_bwd(g[:, :1].contiguous().to(DEV), 9)
  assert torch.equal(gb.cpu(), g[:, :1].expand(-1, 9))


def test_axpy2d_strided(hip):
  x = torch.randn(33, 20)
  y = torch.randn(33, 50)
  yd = y.to(DEV)
  hip.axpy2d(x.to(DEV), 0.5, yd[:, 7:27], accumulate=True)
  y[:, 7:27] += 0.5 * x
  assert torch.allclose(yd.cpu(), y, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------- K6/K7
@pytest.mark.parametrize('B,d,L', [(130, 624, 3), (5, 17, 1), (64, 1000, 5)])
def test_cross_v1(hip, ref, B, d, L):
  rng = np.random.default_rng(d)
  x0 = torch.from_numpy((rng.standard_normal((B, d)) * 0.3).astype(np.float32))
  w = torch.from_numpy((rng.standard_normal((L, d)) * 0.05).astype(np.float32))
  b = torch.from_numpy((rng.standard_normal((L, d)) * 0.05).astype(np.float32))
  dout = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32))
  out_d, dots_d = hip.cross_v1_fwd(x0.to(DEV), w.to(DEV), b.to(DEV))
  out_c, dots_c = ref.cross_v1_fwd(x0, w, b)
  assert torch.allclose(out_d.cpu(), out_c, rtol=1e-4, atol=1e-5)
  dx0_d, dw_d, db_d = hip.cross_v1_bwd(x0.to(DEV), w.to(DEV), b.to(DEV), dots_d, dout.to(DEV))
  dx0_c, dw_c, db_c = ref.cross_v1_bwd(x0, w, b, dots_c, dout)
  assert torch.allclose(dx0_d.cpu(), dx0_c, rtol=1e-3, atol=1e-4)
  assert torch.allclose(dw_d.cpu(), dw_c, rtol=1e-3, atol=1e-3)
  assert torch.allclose(db_d.cpu(), db_c, rtol=1e-3, atol=1e-3)


def test_cross_v2_epilogue(hip, ref):
  B, d = 77, 130
  x0, x, u = torch.randn(B, d), torch.randn(B, d), torch.randn(B, d)
  bias, dout = torch.randn(d), torch.randn(B, d)
  for diag in (0.0, 0.3):
    o_d = hip.cross_v2_fwd(x0.to(DEV), x.to(DEV), u.to(DEV), bias.to(DEV), diag)
    assert torch.allclose(o_d.cpu(), ref.cross_v2_fwd(x0, x, u, bias, diag), rtol=1e-5, atol=1e-5)
    got = hip.cross_v2_bwd(x0.to(DEV), x.to(DEV), u.to(DEV), bias.to(DEV), diag, dout.to(DEV))
    exp = ref.cross_v2_bwd(x0, x, u, bias, diag, dout)
    for a, e in zip(got, exp):
      assert torch.allclose(a.cpu(), e, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------- K8
# (L <= 64 with E / 4 a power of two: the one-pass 16-byte-lane kernels; (4, 70, 8) and (5, 13, 12): the general ones)
@pytest.mark.parametrize('B,L,E', [(33, 50, 32), (4, 70, 8), (9, 1, 16), (7, 64, 64), (5, 13, 12), (130, 50, 4)])
def test_din(hip, ref, B, L, E):
  rng = np.random.default_rng(L)
  q = torch.from_numpy(rng.standard_normal((B, E)).astype(np.float32))
  h = torch.from_numpy(rng.standard_normal((B, L, E)).astype(np.float32))
  lens = torch.from_numpy(rng.integers(1, L + 1, size=B).astype(np.int32))
  c_d = hip.din_concat_fwd(q.to(DEV), h.to(DEV))
  assert torch.equal(c_d.cpu(), ref.din_concat_fwd(q, h))
  dout = torch.from_numpy(rng.standard_normal((B, L, 4 * E)).astype(np.float32))
  dq_d, dh_d = hip.din_concat_bwd(q.to(DEV), h.to(DEV), dout.to(DEV))
  dq_c, dh_c = ref.din_concat_bwd(q, h, dout)
  assert torch.allclose(dq_d.cpu(), dq_c, rtol=1e-4, atol=1e-4) and torch.allclose(dh_d.cpu(), dh_c, rtol=1e-5, atol=1e-5)
  scores = torch.from_numpy(rng.standard_normal((B, L)).astype(np.float32))
  o_d, p_d = hip.din_pool_fwd(scores.to(DEV), h.to(DEV), lens.to(DEV))
  o_c, p_c = ref.din_pool_fwd(scores, h, lens)
  assert torch.allclose(p_d.cpu(), p_c, rtol=1e-5, atol=1e-6) and torch.allclose(o_d.cpu(), o_c, rtol=1e-4, atol=1e-5)
  go = torch.from_numpy(rng.standard_normal((B, E)).astype(np.float32))
  ds_d, dhh_d = hip.din_pool_bwd(p_d, h.to(DEV), lens.to(DEV), go.to(DEV))
  ds_c, dhh_c = ref.din_pool_bwd(p_c, h, lens, go)
  assert torch.allclose(ds_d.cpu(), ds_c, rtol=1e-4, atol=1e-5) and torch.allclose(dhh_d.cpu(), dhh_c, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------- K9
# (40000 x 80: a tall activation like DIN's attention MLP - > 256 partial chunks are merged by bn_*_merge_kernel first and
# every workgroup of the apply kernels walks 16 row tiles)
@pytest.mark.parametrize('B,N', [(4096, 256), (37, 5), (300, 64), (2, 1), (40000, 80)])
@pytest.mark.parametrize('use_bn,act', [(1, 1), (1, 0), (0, 1), (2, 1), (2, 0)])  # 2 = ER_BN_FROZEN (moving statistics)
def test_bn_act(hip, ref, B, N, use_bn, act):
  rng = np.random.default_rng(B + N)
  x = torch.from_numpy((rng.standard_normal((B, N)) * 2 + 0.5).astype(np.float32))
  bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
  gamma = torch.from_numpy((rng.random(N) + 0.5).astype(np.float32))
  beta = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
  mm_c, mv_c = torch.zeros(N), torch.ones(N)
  if use_bn == 2:
    mm_c = torch.from_numpy((rng.standard_normal(N) * 0.5).astype(np.float32))
    mv_c = torch.from_numpy((rng.random(N) * 2 + 0.25).astype(np.float32))
  mm_d, mv_d = mm_c.to(DEV), mv_c.to(DEV)
  dy = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32))
  y_d, mean_d, inv_d = hip.bn_act_fwd(x.to(DEV), bias.to(DEV), gamma.to(DEV), beta.to(DEV), use_bn, 1e-3, 0.99, mm_d,
                                      mv_d, act)
  y_c, mean_c, inv_c = ref.bn_act_fwd(x, bias, gamma, beta, use_bn, 1e-3, 0.99, mm_c, mv_c, act)
  assert torch.allclose(y_d.cpu(), y_c, rtol=1e-4, atol=2e-5)
  if use_bn:
    assert torch.allclose(mean_d.cpu(), mean_c, rtol=1e-5, atol=1e-6)
    assert torch.allclose(mm_d.cpu(), mm_c, rtol=1e-5, atol=1e-7) and torch.allclose(mv_d.cpu(), mv_c, rtol=1e-5)
  # (the backward masks by y > 0: both sides get the SAME y - with 3.2 M elements some normalised values lie within
  # rounding of 0 and the two forward passes disagree on their sign)
  got = hip.bn_act_bwd(x.to(DEV), bias.to(DEV), gamma.to(DEV), y_c.to(DEV), mean_d, inv_d, dy.to(DEV), use_bn, act, True,
                       bool(use_bn))
  exp = ref.bn_act_bwd(x, bias, gamma, y_c, mean_c, inv_c, dy, use_bn, act, True, bool(use_bn))
  for a, e, tol in zip(got, exp, (1e-3, 1e-3, 1e-3, 1e-3)):
    if e is None:
      continue
    scale = float(e.abs().max()) + 1e-6
    assert float((a.cpu() - e).abs().max()) <= tol * scale + 1e-5, (B, N, use_bn, act)


def test_frozen_bn_matches_autograd_of_the_formula(hip):
  """use_bn = ER_BN_FROZEN (tf.layers.batch_normalization(training=False) inside the training graph: the experts of the
  reference's MMoE / DBMTL): forward and every gradient - the bias's too - against torch autograd in fp64."""
  B, N = 700, 70
  g = torch.Generator().manual_seed(5)
  x = torch.randn(B, N, dtype=torch.float64, generator=g, requires_grad=True)
  bias = torch.randn(N, dtype=torch.float64, generator=g, requires_grad=True)
  gamma = (torch.rand(N, dtype=torch.float64, generator=g) + 0.5).requires_grad_(True)
  beta = torch.randn(N, dtype=torch.float64, generator=g, requires_grad=True)
  mm = torch.randn(N, dtype=torch.float64, generator=g) * 0.4
  mv = torch.rand(N, dtype=torch.float64, generator=g) + 0.3
  y = torch.relu((x + bias - mm) / torch.sqrt(mv + 1e-3) * gamma + beta)
  dy = torch.randn(B, N, dtype=torch.float64, generator=g)
  y.backward(dy)
  f = lambda t: t.detach().float().to(DEV)
  mm_d, mv_d = f(mm), f(mv)
  y_d, mean_d, inv_d = hip.bn_act_fwd(f(x), f(bias), f(gamma), f(beta), 2, 1e-3, 0.99, mm_d, mv_d, 1)
  assert torch.equal(mm_d.cpu(), mm.float()) and torch.equal(mv_d.cpu(), mv.float()), 'the moving statistics must stay'
  assert torch.allclose(y_d.cpu().double(), y.detach(), rtol=1e-5, atol=1e-5)
  dx, dbias, dgamma, dbeta = hip.bn_act_bwd(f(x), f(bias), f(gamma), y_d, mean_d, inv_d, f(dy), 2, 1, True, True)
  for got, want in ((dx, x.grad), (dbias, bias.grad), (dgamma, gamma.grad), (dbeta, beta.grad)):
    assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6
  # accumulating into existing buffers (slices of the flat gradient buffer)
  into = tuple(torch.full((N,), 0.5, device=DEV) for _ in range(3))
  hip.bn_act_bwd(f(x), f(bias), f(gamma), y_d, mean_d, inv_d, f(dy), 2, 1, True, True, into=into)
  for got, want in zip(into, (bias.grad, gamma.grad, beta.grad)):
    assert float((got.cpu().double() - 0.5 - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-5


def test_bn_matches_autograd_of_the_formula(hip):
  """independent check of the hand-written BN backward: torch autograd through the textbook formula."""
  B, N = 512, 48
  x = torch.randn(B, N, dtype=torch.float64, requires_grad=True)
  gamma = (torch.rand(N, dtype=torch.float64) + 0.5).requires_grad_(True)
  beta = torch.randn(N, dtype=torch.float64, requires_grad=True)
  mean = x.mean(0)
  var = ((x - mean)**2).mean(0)
  y = torch.relu((x - mean) / torch.sqrt(var + 1e-3) * gamma + beta)
  dy = torch.randn(B, N, dtype=torch.float64)
  y.backward(dy)
  xf, gf, bf = x.detach().float().to(DEV), gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
  y_d, mean_d, inv_d = hip.bn_act_fwd(xf, None, gf, bf, 1, 1e-3, 0.99, None, None, 1)
  dx, _, dg, db = hip.bn_act_bwd(xf, None, gf, y_d, mean_d, inv_d, dy.float().to(DEV), 1, 1, False, True)
  assert torch.allclose(dx.cpu().double(), x.grad, rtol=1e-3, atol=1e-4)
  assert torch.allclose(dg.cpu().double(), gamma.grad, rtol=1e-3, atol=1e-3)
  assert torch.allclose(db.cpu().double(), beta.grad, rtol=1e-3, atol=1e-3)


def test_dice_matches_autograd(hip):
  B, N = 600, 20
  x = torch.randn(B, N, dtype=torch.float64, requires_grad=True)
  alpha = (torch.randn(N, dtype=torch.float64) * 0.3).requires_grad_(True)
  mean = x.mean(0)
  var = ((x - mean)**2).mean(0)
  p = torch.sigmoid((x - mean) / torch.sqrt(var + 1e-9))
  y = alpha * (1 - p) * x + p * x
  dy = torch.randn(B, N, dtype=torch.float64)
  y.backward(dy)
  xf, af = x.detach().float().to(DEV), alpha.detach().float().to(DEV)
  y_d, mean_d, inv_d = hip.dice_fwd(xf, af, 1e-9, 0.99, None, None)
  assert torch.allclose(y_d.cpu().double(), y.detach(), rtol=1e-4, atol=1e-5)
  dx, da = hip.dice_bwd(xf, af, mean_d, inv_d, dy.float().to(DEV))
  assert torch.allclose(dx.cpu().double(), x.grad, rtol=1e-3, atol=1e-4)
  assert torch.allclose(da.cpu().double(), alpha.grad, rtol=1e-3, atol=1e-3)


def test_colsum(hip):
  x = torch.randn(1000, 70)
  assert torch.allclose(hip.colsum(x.to(DEV)).cpu(), x.sum(0), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------- K10
@pytest.mark.parametrize('B', [4096, 1, 300])
def test_sigmoid_ce(hip, ref, B):
  rng = np.random.default_rng(B)
  z = torch.from_numpy((rng.standard_normal(B) * 3).astype(np.float32))
  y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32))
  w = torch.from_numpy((rng.random(B) < 0.8).astype(np.float32) * 2.0)
  for weights in (None, w):
    l_d, g_d, p_d = hip.sigmoid_ce(z.to(DEV), y.to(DEV), None if weights is None else weights.to(DEV), 0.7)
    l_c, g_c, p_c = ref.sigmoid_ce(z, y, weights, 0.7)
    assert torch.allclose(l_d.cpu(), l_c, rtol=1e-5)
    assert torch.allclose(g_d.cpu(), g_c, rtol=1e-4, atol=1e-9) and torch.allclose(p_d.cpu(), p_c, rtol=1e-5)


def test_reductions(hip):
  p = torch.randn(1234)
  out = torch.zeros(1, device=DEV)
  hip.reduce_sum(p.to(DEV), 0.5, out)
  assert abs(float(out.item()) - 0.5 * float(p.double().sum())) < 1e-3
  w, c = torch.randn(5000), (torch.rand(5000) < 0.5).float() * 1e-2
  hip.l2_loss(w.to(DEV), c.to(DEV), out)
  assert abs(float(out.item()) - float((c.double() * 0.5 * w.double()**2).sum())) < 1e-4


# ------------------------------------------------------------------------------------------- K11
def test_mmoe_mix(hip, ref):
  T, E, B, H = 4, 5, 130, 64
  experts, logits, dout = torch.randn(E, B, H), torch.randn(T, B, E), torch.randn(T, B, H)
  o_d, g_d = hip.mmoe_mix_fwd(experts.to(DEV), logits.to(DEV))
  o_c, g_c = ref.mmoe_mix_fwd(experts, logits)
  assert torch.allclose(o_d.cpu(), o_c, rtol=1e-4, atol=1e-5) and torch.allclose(g_d.cpu(), g_c, rtol=1e-5, atol=1e-6)
  de_d, dl_d = hip.mmoe_mix_bwd(experts.to(DEV), g_d, dout.to(DEV))
  de_c, dl_c = ref.mmoe_mix_bwd(experts, g_c, dout)
  assert torch.allclose(de_d.cpu(), de_c, rtol=1e-4, atol=1e-5) and torch.allclose(dl_d.cpu(), dl_c, rtol=1e-3, atol=1e-4)


# ------------------------------------------------------------------------------------------- dense optimizer
@pytest.mark.parametrize('opt', [kernels.OPT_SGD, kernels.OPT_ADAM, kernels.OPT_ADAGRAD])
def test_dense_opt_bit_exact(hip, ref, opt):
  n = 100003
  rng = np.random.default_rng(opt)
  w = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
  m = torch.from_numpy((rng.standard_normal(n) * 0.01).astype(np.float32))
  v = torch.from_numpy((rng.random(n) * 0.01 + 1e-4).astype(np.float32))
  g = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
  c = torch.from_numpy(((rng.random(n) < 0.5) * 1e-3).astype(np.float32))
  hyper = _hyper(t=7)
  wd, md, vd = w.to(DEV), m.to(DEV), v.to(DEV)
  hip.dense_opt_step(wd, md, vd, g.to(DEV), c.to(DEV), opt, hyper.to(DEV))
  ref.dense_opt_step(w, m, v, g, c, opt, hyper)
  torch.cuda.synchronize()
  assert torch.equal(md.cpu(), m) and torch.equal(vd.cpu(), v)
  assert torch.equal(wd.cpu(), w)


# ------------------------------------------------------------------------------------------- K13 GEMM
_GEMM_SHAPES = [  # (M, N, K): DeepFM layers, odd sizes (81, 1), tiny, split-K (small MxN, long K)
    (4096, 256, 624), (4096, 64, 128), (4096, 256, 81), (4096, 1, 64), (624, 256, 4096), (81, 256, 4096),
    (64, 1, 4096), (33, 65, 97), (1, 1, 1), (130, 70, 1000)]


@pytest.mark.parametrize('layout', [kernels.GEMM_NN, kernels.GEMM_NT, kernels.GEMM_TN])
@pytest.mark.parametrize('M,N,K', _GEMM_SHAPES)
def test_gemm_f32(hip, layout, M, N, K):
  """fp32 MFMA GEMM against an fp64 matmul: every element within 1e-6 of sum |a||b| (an f32 fma chain)."""
  g = torch.Generator().manual_seed(M * 131 + N * 17 + K + layout)
  a_shape = (K, M) if layout == kernels.GEMM_TN else (M, K)
  b_shape = (N, K) if layout == kernels.GEMM_NT else (K, N)
  a = torch.randn(a_shape, generator=g)
  b = torch.randn(b_shape, generator=g)
  bias = torch.randn(N, generator=g)
  A = a.t() if layout == kernels.GEMM_TN else a
  Bm = b.t() if layout == kernels.GEMM_NT else b
  ref = A.double() @ Bm.double()
  bound = (A.double().abs() @ Bm.double().abs()) * 1e-6 + 1e-6
  got = hip.gemm(layout, a.to(DEV), b.to(DEV))
  assert ((got.cpu().double() - ref).abs() <= bound).all()
  # bias + accumulate into a strided output view
  base = torch.randn(M, N + 3, generator=g)
  out = base.to(DEV)
  hip.gemm(layout, a.to(DEV), b.to(DEV), out=out[:, 1:N + 1], bias=bias.to(DEV), accumulate=True)
  exp = base.double()
  exp[:, 1:N + 1] += ref + bias.double()
  torch.cuda.synchronize()
  assert ((out.cpu().double() - exp).abs() <= torch.nn.functional.pad(bound, (1, 2), value=0.0) + 1e-5).all()
  # deterministic: same bits on a second launch
  again = hip.gemm(layout, a.to(DEV), b.to(DEV))
  assert torch.equal(got, again)


@pytest.mark.parametrize('layout', [kernels.GEMM_NN, kernels.GEMM_NT, kernels.GEMM_TN])
@pytest.mark.parametrize('M,N,K', [(4096, 256, 624), (624, 256, 4096), (33, 65, 97), (4096, 1, 64)])
def test_gemm_bf16(hip, layout, M, N, K):
  """bf16 MFMA GEMM: operands rounded to bfloat16 (RNE, checked against torch's cast), fp32 accumulate."""
  g = torch.Generator().manual_seed(M + N + K + layout)
  a_shape = (K, M) if layout == kernels.GEMM_TN else (M, K)
  b_shape = (N, K) if layout == kernels.GEMM_NT else (K, N)
  a, b = torch.randn(a_shape, generator=g), torch.randn(b_shape, generator=g)
  ar, br = a.to(torch.bfloat16).double(), b.to(torch.bfloat16).double()
  A = ar.t() if layout == kernels.GEMM_TN else ar
  Bm = br.t() if layout == kernels.GEMM_NT else br
  ref = A @ Bm
  bound = (A.abs() @ Bm.abs()) * 2e-6 + 1e-6
  got = hip.gemm(layout, a.to(DEV), b.to(DEV), bf16=True)
  assert ((got.cpu().double() - ref).abs() <= bound).all()


def test_linear_fn_matches_torch_autograd(hip):
  x = torch.randn(300, 81, device=DEV, requires_grad=True)
  w = torch.randn(81, 40, device=DEV, requires_grad=True)
  b = torch.randn(40, device=DEV, requires_grad=True)
  wg, bg = torch.zeros_like(w), torch.zeros_like(b)
  y = kernels.LinearFn.apply(x, w, b, wg, bg, False)
  dy = torch.randn_like(y)
  y.backward(dy)
  xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
  yr = xr @ wr + br
  yr.backward(dy.double())
  assert torch.allclose(y.double(), yr, rtol=1e-5, atol=1e-5)
  assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-5, atol=1e-5)
  assert torch.allclose(wg.double(), wr.grad, rtol=1e-5, atol=1e-4)
  assert torch.allclose(bg.double(), br.grad, rtol=1e-5, atol=1e-4)
  assert w.grad is None and b.grad is None  # accumulated into the given buffers instead


@pytest.mark.parametrize('M,N,K', [(4096, 256, 624), (300, 40, 81), (130, 70, 33), (64, 1, 16), (20000, 40, 24)])
@pytest.mark.parametrize('bf16', [False, True])
def test_gemm_epilogue_statistics_feed_batchnorm(hip, M, N, K, bf16):
  """er_gemm's column statistics + er_bn_apply_from_stats == BatchNorm(train)+ReLU of the GEMM output."""
  g = torch.Generator().manual_seed(M + N + K)
  x, w = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) * 0.1
  bias, gamma, beta = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
  mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
  chunks = hip.gemm_row_tiles(M)
  stats = torch.zeros(chunks * N * 3, device=DEV)
  z = hip.gemm(kernels.GEMM_NN, x.to(DEV), w.to(DEV), bias=bias.to(DEV), bf16=bf16, col_stats=stats)
  y, mean, invstd = hip.bn_apply_from_stats(z, None, stats, chunks, gamma.to(DEV), beta.to(DEV), 1e-3, 0.99, mm, mv,
                                            kernels.ACT_RELU)
  torch.cuda.synchronize()
  zr = z.cpu().double()
  mu, var = zr.mean(dim=0), zr.var(dim=0, unbiased=False)
  assert torch.allclose(mean.cpu().double(), mu, rtol=1e-5, atol=1e-5)
  assert torch.allclose(invstd.cpu().double(), 1.0 / torch.sqrt(var + 1e-3), rtol=1e-5, atol=1e-6)
  yr = torch.relu((zr - mu) / torch.sqrt(var + 1e-3) * gamma.double() + beta.double())
  assert torch.allclose(y.cpu().double(), yr, rtol=1e-4, atol=1e-5)
  assert torch.allclose(mm.cpu().double(), 0.01 * mu, rtol=1e-4, atol=1e-6)
  assert torch.allclose(mv.cpu().double(), 0.99 + 0.01 * var, rtol=1e-4, atol=1e-6)


def test_lazy_dense_decay_equals_the_sweep(hip):
  """TF-exact Adam two ways over 1300 steps on one table: (A) the streaming sweep of every row every step,
  (B) lazy dense decay (er_emb_route -> er_emb_catch_up -> touched-row update, er_emb_flush_decay at the end).
  Rows touched early and never again sit idle for > 1000 steps (m settles on its denormal fixed point: the
  closed-form tail of v is exercised).  var and m must agree bit for bit wherever no closed-form step happened, v to 1e-5."""
  rng = np.random.default_rng(11)
  rows, dim, B, T = 97, 16, 6, 1300
  table0 = torch.from_numpy((rng.standard_normal((rows, dim)) * 0.05).astype(np.float32))
  ids_all = rng.integers(0, rows, size=(T, B)).astype(np.int64)
  ids_all[60:, :] = ids_all[60:, :] % 11            # after step 60 only rows 0..10 are ever touched again
  dout_all = (rng.standard_normal((T, B, dim)) * 0.01).astype(np.float32)
  state = {}
  # lazy_roll: + the rolling flush (er_emb_flush_window, 16 windows) after the row update; lazy_roll_side: the flush
  # with lag 1 on a second stream, concurrent with the row update of the same step
  side = torch.cuda.Stream()
  for mode in ('sweep', 'lazy', 'lazy_roll', 'lazy_roll_side'):
    var, m, v = table0.clone().to(DEV), torch.zeros(rows, dim, device=DEV), torch.zeros(rows, dim, device=DEV)
    ids = torch.zeros(B, dtype=torch.int64, device=DEV)
    dout = torch.zeros(B, dim, device=DEV)
    bitmap = torch.zeros((rows + 31) // 32, dtype=torch.int32, device=DEV) if mode == 'sweep' else None
    spec = kernels.LookupSpec(table=var, ids=ids, offsets=None, weights=None, out=dout, out_col=0, rows=rows,
                              key_base=0, dim=dim, combiner=0, n_rows=B, max_nnz=B)
    g = hip.emb_group_create([spec], dim, rows, var, m, v, bitmap)
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    cap = T + 8
    hist = torch.zeros(2 * cap, device=DEV)  # [lr_t per step | running maximum] as er_hyper_select lays it out
    hyper = torch.zeros(kernels.HYPER_FLOATS, device=DEV)
    if mode != 'sweep':
      last = torch.full((rows,), -1, dtype=torch.int32, device=DEV)
      hip.emb_group_enable_lazy_decay(g, last, hist, counter)
      ukeys = torch.zeros(B, dtype=torch.int32, device=DEV)
      nu = torch.zeros(1, dtype=torch.int32, device=DEV)
      uidx = torch.zeros(B, dtype=torch.int64, device=DEV)
      cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    rows_h = torch.stack([_hyper(lr=1e-2 * (0.5**(s // 400)), t=s + 1) for s in range(T)]).to(DEV)
    hist[:T] = rows_h[:, kernels.HYPER_LR_T]
    hist[cap:cap + T] = torch.cummax(rows_h[:, kernels.HYPER_LR_T], 0).values
    for s in range(T):
      hyper.copy_(rows_h[s])
      counter.fill_(s + 1)
      ids.copy_(torch.from_numpy(ids_all[s]))
      dout.copy_(torch.from_numpy(dout_all[s]))
      if mode != 'sweep':
        hip.emb_route(g, ukeys, nu, uidx, cnt)
        hip.emb_catch_up(g, ukeys, nu, hyper)
      if mode == 'lazy_roll_side':
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          hip.emb_flush_window([g], 16, hyper, lag=1, max_blocks=1)
      hip.emb_bwd_update(g, kernels.OPT_ADAM, hyper)
      if mode == 'lazy_roll':
        hip.emb_flush_window([g], 16, hyper)
      if mode == 'lazy_roll_side':
        torch.cuda.current_stream().wait_stream(side)
    if mode != 'sweep':
      hip.emb_flush_decay(g, hyper)
      torch.cuda.synchronize()
      assert int(last.min()) == T - 1
    torch.cuda.synchronize()
    state[mode] = (var.cpu(), m.cpu(), v.cpu())
    hip.emb_group_destroy(g)
  (va, ma, sa) = state['sweep']
  hot = torch.arange(rows) < 11
  # fp32 never reaches 0 by repeated * 0.9: it settles on a denormal fixed point (|m| <= 4 * 2^-149)
  assert (ma[~hot].abs() <= 6e-45).all(), 'idle rows: m has settled (the closed-form tail of v was taken)'
  for mode in ('lazy', 'lazy_roll', 'lazy_roll_side'):
    vb, mb, sb = state[mode]
    assert torch.equal(ma, mb), (mode, 'first moments must be bit-identical')
    assert torch.equal(va[hot], vb[hot]) and torch.equal(sa[hot], sb[hot]), (mode, 'recently touched rows: every bit')
    assert torch.equal(va, vb), (mode, 'var: every bit, through the full, the absorbed and the settled regime')
    if mode != 'lazy':  # never more than 16 steps behind: v decays step by step too
      assert torch.equal(sa, sb), mode
    else:  # one catch-up of > 1200 steps at the flush: the closed-form tail of v (backlog > 2048: no; <= 2048: exact)
      assert torch.equal(sa, sb), mode


def test_closed_form_decay_tracks_the_sweep(hip):
  """The closed-form replay (csrc/er_decay.h: er_decay_tables_create, the per-step table appended by
  er_step_prologue_decay, the per-launch table in front of er_emb_catch_up / er_emb_flush_decay) against the streaming
  sweep of every row every step, 1300 steps, rows idle for 1 .. 1240 steps, a halving learning rate: var within 1e-6 of
  the table's scale, m within 1e-5 and v within 2e-4 relative (step-by-step fp32 rounding of 1240 multiplications
  drifts by that much from beta^k; the closed form is the more accurate of the two)."""
  rng = np.random.default_rng(11)
  rows, dim, B, T = 97, 16, 6, 1300
  table0 = torch.from_numpy((rng.standard_normal((rows, dim)) * 0.05).astype(np.float32))
  ids_all = rng.integers(0, rows, size=(T, B)).astype(np.int64)
  ids_all[60:, :] = ids_all[60:, :] % 11            # after step 60 only rows 0..10 are ever touched again
  ids_all[700, 0] = 50                               # ... except one touch of a row idle for > 600 steps
  dout_all = (rng.standard_normal((T, B, dim)) * 0.01).astype(np.float32)
  dout_all[:, :, :4] *= 1e-4                         # columns whose sqrt(v) is comparable to eps
  dout_all[:, :, 4:6] *= 1e-7                        # ... and far below it
  state = {}
  rows_h = torch.stack([_hyper(lr=1e-2 * (0.5**(s // 400)), t=s + 1) for s in range(T)]).to(DEV)
  for mode in ('sweep', 'closed'):
    var, m, v = table0.clone().to(DEV), torch.zeros(rows, dim, device=DEV), torch.zeros(rows, dim, device=DEV)
    ids = torch.zeros(B, dtype=torch.int64, device=DEV)
    dout = torch.zeros(B, dim, device=DEV)
    bitmap = torch.zeros((rows + 31) // 32, dtype=torch.int32, device=DEV) if mode == 'sweep' else None
    spec = kernels.LookupSpec(table=var, ids=ids, offsets=None, weights=None, out=dout, out_col=0, rows=rows,
                              key_base=0, dim=dim, combiner=0, n_rows=B, max_nnz=B)
    g = hip.emb_group_create([spec], dim, rows, var, m, v, bitmap)
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    cap = T + 8
    hist = torch.zeros(2 * cap, device=DEV)
    hyper = torch.zeros(kernels.HYPER_FLOATS, device=DEV)
    tabs = None
    if mode == 'closed':
      last = torch.full((rows,), -1, dtype=torch.int32, device=DEV)
      hip.emb_group_enable_lazy_decay(g, last, hist, counter)
      tabs = hip.decay_tables_create(hist, counter, 0.9, 0.999)
      assert tabs is not None
      hip.emb_group_set_decay_tables(g, tabs)
      ukeys = torch.zeros(B, dtype=torch.int32, device=DEV)
      nu = torch.zeros(1, dtype=torch.int32, device=DEV)
      uidx = torch.zeros(B, dtype=torch.int64, device=DEV)
      cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    for s in range(T):
      hip.step_prologue(rows_h, counter, hyper, history=hist, decay_tables=tabs)  # slot s of the table; counter = s + 1
      ids.copy_(torch.from_numpy(ids_all[s]))
      dout.copy_(torch.from_numpy(dout_all[s]))
      if mode == 'closed':
        hip.emb_route(g, ukeys, nu, uidx, cnt)
        hip.emb_catch_up(g, ukeys, nu, hyper)
      hip.emb_bwd_update(g, kernels.OPT_ADAM, hyper)
    if mode == 'closed':
      hip.emb_flush_decay(g, hyper)
      torch.cuda.synchronize()
      assert int(last.min()) == T - 1
    torch.cuda.synchronize()
    state[mode] = (var.cpu().double(), m.cpu().double(), v.cpu().double())
    hip.emb_group_destroy(g)
    hip.decay_tables_destroy(tabs)
  (va, ma, sa), (vb, mb, sb) = state['sweep'], state['closed']
  scale = float(va.abs().max())
  dv = float((va - vb).abs().max()) / scale
  print('closed form vs sweep: var %.3g of the table scale' % dv)
  assert dv <= 1e-6, dv
  big_m = ma.abs() > 1e-30
  assert float(((ma - mb).abs()[big_m] / ma.abs()[big_m]).max()) <= 1e-4
  assert float((ma - mb).abs()[~big_m].max()) <= 1e-30
  big_v = sa > 1e-35
  assert float(((sa - sb).abs()[big_v] / sa[big_v]).max()) <= 2e-4


@pytest.mark.parametrize('sizes', [(100, 4096, 5000, 8192, 257), (1, 2, 3), (4097,), (8192, 8192), (9000, 50),
                                   (-3000000, 4096), (-600000, 8192, 77), (8193,), (16384, 1), (20000, 8192, 3),
                                   (204800,), (70000, 9001, 130000)])
def test_segmented_sort_sizes(hip, sizes):
  """One table per lookup -> one workgroup sorts one lookup's entries (emb_segment_sort_kernel); a lookup above 8192
  entries takes the device-wide sort - hand-written since round 5: 8192-entry chunks sorted in LDS, then pairwise
  merge-path passes (1 to 5 of them here, a ragged last chunk, DIN's 4096 x 50 sequence lookup) and the two-level head
  scan.  The de-duplicated keys must come out ascending and every row's gradient must be the source-order sum of its
  entries."""
  rng = np.random.default_rng(abs(sum(sizes)))
  dim = 4
  specs, base, exp_keys, exp = [], 0, [], []
  big_rows = -sizes[0] if sizes[0] < 0 else None  # a table too large for 32-bit sort composites -> 64-bit path
  sizes = sizes[1:] if big_rows else sizes
  for li, n in enumerate(sizes):
    rows = big_rows if (big_rows and li == 0) else int(rng.integers(1, 3 * n + 2))
    ids = rng.integers(-1, rows, size=n).astype(np.int64)
    if big_rows and li == 0:
      ids[-1] = rows - 1
    if n > 64:
      ids[:n // 3] = ids[0]  # one long run
    dout = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
    specs.append(kernels.LookupSpec(table=torch.zeros(rows, dim, device=DEV), ids=torch.from_numpy(ids).to(DEV),
                                    offsets=None, weights=None, out=torch.from_numpy(dout).to(DEV), out_col=0,
                                    rows=rows, key_base=base, dim=dim, combiner=0, n_rows=n, max_nnz=n))
    acc = np.zeros((rows, dim), dtype=np.float64)
    ok = ids >= 0
    np.add.at(acc, ids[ok], dout[ok].astype(np.float64))
    u = np.unique(ids[ok])
    exp_keys.append(u + base)
    exp.append(acc[u])
    base += rows
  var = torch.zeros(base, dim, device=DEV)
  g = hip.emb_group_create(specs, dim, base, var, None, None, None)
  keys, grads, n = hip.emb_bwd_reduce(g)
  torch.cuda.synchronize()
  exp_keys, exp = np.concatenate(exp_keys), np.concatenate(exp)
  assert int(n.item()) == len(exp_keys)
  assert np.array_equal(keys[:len(exp_keys)].cpu().numpy(), exp_keys)
  assert np.allclose(grads[:len(exp_keys)].cpu().numpy(), exp, rtol=1e-5, atol=1e-7)
  keys2, grads2, _ = hip.emb_bwd_reduce(g)
  torch.cuda.synchronize()
  assert torch.equal(grads[:len(exp_keys)], grads2[:len(exp_keys)])
  hip.emb_group_destroy(g)


@pytest.mark.parametrize('n_lookups,n,rows', [(26, 4096, 5000), (3, 300, 40), (7, 9000, 100000), (2, 5, 3)])
def test_lookups_that_share_one_table_take_the_device_wide_sort(hip, n_lookups, n, rows):
  """Several features on ONE table (the reference's own embedding-parallel Criteo config,
  samples/model_config/dlrm_on_criteo_parquet_ep_v2.config; `embedding_name` sharing): the lookups' key ranges coincide, so
  the group's order is not the concatenation of per-lookup sorts - chunk sort + merge passes over all entries.  A key's
  gradient is the sum over every lookup that read it, in entry order."""
  rng = np.random.default_rng(n_lookups * 1000 + n)
  dim = 8
  table = torch.zeros(rows, dim, device=DEV)
  specs = []
  acc = np.zeros((rows, dim), dtype=np.float64)
  for li in range(n_lookups):
    ids = rng.integers(-1, rows, size=n).astype(np.int64)
    ids[: n // 4] = ids[0]
    dout = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
    specs.append(kernels.LookupSpec(table=table, ids=torch.from_numpy(ids).to(DEV), offsets=None, weights=None,
                                    out=torch.from_numpy(dout).to(DEV), out_col=0, rows=rows, key_base=0, dim=dim,
                                    combiner=0, n_rows=n, max_nnz=n))
    ok = ids >= 0
    np.add.at(acc, ids[ok], dout[ok].astype(np.float64))
  touched = np.zeros(rows, dtype=bool)
  for sp in specs:
    i = sp.ids.cpu().numpy()
    touched[i[i >= 0]] = True
  exp_keys = np.nonzero(touched)[0]
  g = hip.emb_group_create(specs, dim, rows, table, None, None, None)
  keys, grads, cnt = hip.emb_bwd_reduce(g)
  torch.cuda.synchronize()
  assert int(cnt.item()) == len(exp_keys)
  assert np.array_equal(keys[:len(exp_keys)].cpu().numpy(), exp_keys)
  assert np.allclose(grads[:len(exp_keys)].cpu().numpy(), acc[exp_keys], rtol=1e-5, atol=1e-7)
  keys2, grads2, _ = hip.emb_bwd_reduce(g)  # deterministic: the same bits again
  torch.cuda.synchronize()
  assert torch.equal(grads[:len(exp_keys)], grads2[:len(exp_keys)]) and torch.equal(keys[:len(exp_keys)], keys2[:len(exp_keys)])
  hip.emb_group_destroy(g)


@pytest.mark.parametrize('lazy', [False, True])
def test_shared_sort_between_wide_and_deep_groups(hip, lazy):
  """A dim-1 (wide) and a dim-16 (deep) table group reading the same id columns: the follower reuses the leader's
  sort (er_emb_group_share_sort).  Tables and Adam slots must equal, bit for bit, those of two independent groups."""
  rng = np.random.default_rng(3)
  B, n_feat, T = 700, 5, 4
  rows = [int(rng.integers(3, 400)) for _ in range(n_feat)]
  ids_steps = [[rng.integers(-1, r, size=B).astype(np.int64) for r in rows] for _ in range(T)]
  results = {}
  for share in (False, True, 'multi'):  # 'multi': shared sort + the groups' kernels fused into one launch each
    ids = [torch.zeros(B, dtype=torch.int64, device=DEV) for _ in rows]
    groups, state = {}, {}
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    hist = torch.zeros(2 * (T + 8), device=DEV)  # [lr_t per step | running maximum]
    hyper = torch.zeros(kernels.HYPER_FLOATS, device=DEV)
    rows_h = torch.stack([_hyper(lr=1e-2, t=s + 1) for s in range(T)]).to(DEV)
    hist[:T] = rows_h[:, kernels.HYPER_LR_T]
    hist[T + 8:2 * T + 8] = torch.cummax(rows_h[:, kernels.HYPER_LR_T], 0).values
    total = sum(rows)
    lz = {}
    for dim in (1, 16):
      g0 = torch.Generator().manual_seed(dim)
      var = (torch.randn(total, dim, generator=g0) * 0.05).to(DEV)
      m, v = torch.zeros(total, dim, device=DEV), torch.zeros(total, dim, device=DEV)
      dout = torch.zeros(B, n_feat * dim, device=DEV)
      specs, base = [], 0
      for f, r in enumerate(rows):
        specs.append(kernels.LookupSpec(table=var[base:base + r], ids=ids[f], offsets=None, weights=None, out=dout,
                                        out_col=f * dim, rows=r, key_base=base, dim=dim, combiner=0, n_rows=B,
                                        max_nnz=B))
        base += r
      bitmap = None if lazy else torch.zeros((total + 31) // 32, dtype=torch.int32, device=DEV)
      groups[dim] = hip.emb_group_create(specs, dim, total, var, m, v, bitmap)
      state[dim] = (var, m, v, dout)
      if lazy:
        lz[dim] = dict(last=torch.full((total,), -1, dtype=torch.int32, device=DEV),
                       ukeys=torch.zeros(B * n_feat, dtype=torch.int32, device=DEV),
                       nu=torch.zeros(1, dtype=torch.int32, device=DEV))
        hip.emb_group_enable_lazy_decay(groups[dim], lz[dim]['last'], hist, counter)
    if share:
      assert hip.emb_group_share_sort(groups[16], groups[1])
    for s in range(T):
      hyper.copy_(rows_h[s])
      counter.fill_(s + 1)
      for f in range(n_feat):
        ids[f].copy_(torch.from_numpy(ids_steps[s][f]))
      for dim in (1, 16):
        gd = torch.Generator().manual_seed(100 * s + dim)
        state[dim][3].copy_((torch.randn(B, n_feat * dim, generator=gd) * 0.01).to(DEV))
      if lazy and share == 'multi':
        hip.emb_route(groups[1], lz[1]['ukeys'], lz[1]['nu'], None, None)
        hip.emb_route(groups[16], None, None, None, None)
        hip.emb_catch_up_multi([groups[1], groups[16]], [lz[1]['ukeys']] * 2, [lz[1]['nu']] * 2, hyper)
      elif lazy:
        for dim in (1, 16):
          if share and dim == 16:
            hip.emb_route(groups[16], None, None, None, None)
            hip.emb_catch_up(groups[16], lz[1]['ukeys'], lz[1]['nu'], hyper)
          else:
            hip.emb_route(groups[dim], lz[dim]['ukeys'], lz[dim]['nu'], None, None)
            hip.emb_catch_up(groups[dim], lz[dim]['ukeys'], lz[dim]['nu'], hyper)
      if share == 'multi':
        hip.emb_bwd_update_multi([groups[1], groups[16]], kernels.OPT_ADAM, hyper)
      else:
        for dim in (1, 16):
          hip.emb_bwd_update(groups[dim], kernels.OPT_ADAM, hyper)
    if lazy:
      for dim in (1, 16):
        hip.emb_flush_decay(groups[dim], hyper)
    torch.cuda.synchronize()
    results[share] = {dim: tuple(t.cpu() for t in state[dim][:3]) for dim in (1, 16)}
    if share:  # the follower refuses a stale sort
      with pytest.raises(RuntimeError):
        hip.emb_bwd_update(groups[16], kernels.OPT_ADAM, hyper)
    for dim in (16, 1):
      hip.emb_group_destroy(groups[dim])
  for dim in (1, 16):
    for variant in (True, 'multi'):
      for a, b, what in zip(results[False][dim], results[variant][dim], ('var', 'm', 'v')):
        assert torch.equal(a, b), (dim, what, variant)
  assert float(results[True][16][1].abs().max()) > 0


def test_gemm_grouped_matches_single_launches(hip):
  """The weight gradients of a step in ONE grouped launch (er_gemm_grouped_f32): every problem within the f32 bound of
  an fp64 matmul, accumulate honoured, bit-identical across launches; > 16 problems are chunked."""
  g = torch.Generator().manual_seed(7)
  hip.gemm_reserve(1 << 22)
  shapes = [(624, 256, 4096), (256, 128, 4096), (128, 64, 4096), (81, 256, 4096), (64, 1, 4096), (33, 65, 97),
            (1, 1, 1), (130, 72, 1000), (320, 128, 20000), (129, 129, 131)] * 3  # 30 problems
  probs, refs, bases = [], [], []
  for (M, N, K) in shapes:
    a, b = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    base = torch.randn(M, N, generator=g)
    out = base.to(DEV)
    probs.append((a.to(DEV), b.to(DEV), out, None, True))
    refs.append(((a.double().t() @ b.double()), (a.double().abs().t() @ b.double().abs()) * 1e-6 + 1e-5))
    bases.append(base)
  hip.gemm_grouped(kernels.GEMM_TN, probs)
  torch.cuda.synchronize()
  first = [p[2].clone() for p in probs]
  for (ref, bound), base, got in zip(refs, bases, first):
    assert ((got.cpu().double() - (base.double() + ref)).abs() <= bound).all()
  for p, base in zip(probs, bases):
    p[2].copy_(base.to(DEV))
  hip.gemm_grouped(kernels.GEMM_TN, probs)
  torch.cuda.synchronize()
  for p, f in zip(probs, first):
    assert torch.equal(p[2], f)


@pytest.mark.parametrize('shapes', [
    [(624, 256, 4096), (256, 128, 4096), (128, 64, 4096), (81, 256, 4096), (256, 128, 4096), (128, 64, 4096)],  # 8 splits each
    [(130, 72, 30000), (64, 64, 30000)],   # 63 splits: 56 dealt to the XCDs whole + 7 in the legacy region
    [(200, 40, 2048), (64, 200, 2304)],    # 16 / 18 splits
])
def test_gemm_grouped_splits_by_xcd(hip, shapes):
  """A grouped weight-gradient launch whose problems get >= 8 k-splits: the XCD region of the grid (whole splits per XCD,
  er_gemm_grouped_layout) computes every tile of every split - within the f32 bound of an fp64 matmul, accumulate
  honoured - and the sum over splits is the fixed-order reduce's: bit-identical from launch to launch."""
  g = torch.Generator().manual_seed(11)
  hip.gemm_reserve(1 << 23)
  probs, refs, bases = [], [], []
  for (M, N, K) in shapes:
    a, b = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    base = torch.randn(M, N, generator=g)
    probs.append((a.to(DEV), b.to(DEV), base.to(DEV), None, True))
    refs.append(((a.double().t() @ b.double()), (a.double().abs().t() @ b.double().abs()) * 1e-6 + 1e-5))
    bases.append(base)
  hip.gemm_grouped(kernels.GEMM_TN, probs)
  torch.cuda.synchronize()
  first = [p[2].clone() for p in probs]
  for (ref, bound), base, got in zip(refs, bases, first):
    assert ((got.cpu().double() - (base.double() + ref)).abs() <= bound).all()
  for p, base in zip(probs, bases):
    p[2].copy_(base.to(DEV))
  hip.gemm_grouped(kernels.GEMM_TN, probs)
  torch.cuda.synchronize()
  for p, f in zip(probs, first):
    assert torch.equal(p[2], f)


def test_sigmoid_ce_multi_matches_single_launches(hip):
  """er_sigmoid_ce_multi (the towers' losses of a multi-task model in one launch): head by head the bits of
  er_sigmoid_ce_fwd_bwd - with and without example weights, different batch sizes and scales, more heads than a launch."""
  g = torch.Generator().manual_seed(31)
  heads = []
  for i in range(11):
    B = [8192, 4096, 1, 777][i % 4]
    z = (torch.randn(B, generator=g) * 3).to(DEV)
    y = (torch.rand(B, generator=g) > 0.7).float().to(DEV)
    w = None if i % 3 == 0 else (torch.rand(B, generator=g) * (torch.rand(B, generator=g) > 0.2).float()).to(DEV)
    heads.append((z, y, w, 0.5 + 0.25 * i))
  single = [hip.sigmoid_ce(z, y, w, s)[:2] for z, y, w, s in heads]
  multi = hip.sigmoid_ce_multi(heads)
  torch.cuda.synchronize()
  for i, ((l1, d1), (l2, d2)) in enumerate(zip(single, multi)):
    assert torch.equal(l1, l2) and torch.equal(d1, d2), i


def test_copy_multi(hip):
  """er_copy_multi: any mix of sizes / alignments / dtypes, more items than one launch holds."""
  g = torch.Generator().manual_seed(5)
  pairs, refs = [], []
  for i in range(37):
    n = [1, 3, 4, 16, 1000, 4096 * 50, 12345, 7][i % 8]
    dt = [torch.float32, torch.int64, torch.int32, torch.uint8][i % 4]
    src = torch.randint(0, 100, (n + 3,), generator=g).to(dt).to(DEV)
    off = i % 3  # odd element offsets: unaligned byte addresses for the narrow types
    s_ = src[off:off + n]
    dst = torch.zeros(n + 5, dtype=dt, device=DEV)
    d = dst[1:1 + n] if i % 2 else dst[:n]
    pairs.append((d, s_))
    refs.append((dst, d, s_.clone()))
  hip.copy_multi(pairs)
  torch.cuda.synchronize()
  for dst, d, want in refs:
    assert torch.equal(d, want)
    assert int(dst.sum()) == int(want.sum())  # nothing written outside the destination


def test_gemm_grouped_bf16_matches_rounded_operands(hip):
  """er_gemm_grouped_bf16 (the weight gradients of a bf16 step in one launch): every problem = the fp32-accumulated
  product of the operands rounded to bf16 (fp64 reference of the ROUNDED operands within the fp32 summation bound),
  accumulate honoured, bit-identical across launches."""
  g = torch.Generator().manual_seed(23)
  hip.gemm_reserve(1 << 22)
  shapes = [(429, 256, 4096), (256, 128, 4096), (128, 64, 4096), (64, 1, 4096), (33, 65, 97), (130, 72, 1000)] * 3
  probs, refs, bases = [], [], []
  for (M, N, K) in shapes:
    a, b = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    base = torch.randn(M, N, generator=g)
    probs.append((a.to(DEV), b.to(DEV), base.to(DEV), None, True))
    ar, br = a.to(torch.bfloat16).double(), b.to(torch.bfloat16).double()
    refs.append((ar.t() @ br, (ar.abs().t() @ br.abs()) * 1e-6 + 1e-5))
    bases.append(base)
  hip.gemm_grouped(kernels.GEMM_TN, probs, bf16=True)
  torch.cuda.synchronize()
  first = [p[2].clone() for p in probs]
  for (ref, bound), base, got in zip(refs, bases, first):
    assert ((got.cpu().double() - (base.double() + ref)).abs() <= bound).all()
  for p, base in zip(probs, bases):
    p[2].copy_(base.to(DEV))
  hip.gemm_grouped(kernels.GEMM_TN, probs, bf16=True)
  torch.cuda.synchronize()
  for p, f in zip(probs, first):
    assert torch.equal(p[2], f)


def test_grouped_launch_epilogues_match_single_launches(hip):
  """What kernels.GroupedLinearFn relies on: problems of ONE grouped launch with their own epilogues - bias + per-row-tile
  column statistics forward (er_gemm_problem.col_stats), the BatchNorm-backward column sums of the producing layer on the
  input-gradient side (er_gemm_problem.bn_*) - give the bits of the single launches er_gemm_f32(col_stats) /
  er_gemm_f32_bn_bwd, next to problems without any epilogue."""
  g = torch.Generator().manual_seed(3)
  shapes = [(8192, 256, 64), (8192, 192, 256), (300, 70, 33), (8192, 4, 64)]
  xs = [torch.randn(M, K, generator=g).to(DEV) for (M, N, K) in shapes]
  ws = [(torch.randn(K, N, generator=g) * 0.1).to(DEV) for (M, N, K) in shapes]
  bs = [torch.randn(N, generator=g).to(DEV) for (M, N, K) in shapes]
  want_stats = [True, True, True, False]
  single_z, single_st = [], []
  for x, w, b, ws_ in zip(xs, ws, bs, want_stats):
    st = torch.zeros(hip.gemm_row_tiles(x.shape[0]) * w.shape[1] * 3, device=DEV) if ws_ else None
    single_z.append(hip.gemm(kernels.GEMM_NN, x, w, bias=b, col_stats=st))
    single_st.append(st)
  zs = [torch.empty_like(z) for z in single_z]
  sts = [torch.zeros_like(st) if st is not None else None for st in single_st]
  hip.gemm_grouped(kernels.GEMM_NN, [(x, w, z, b, False, st) for x, w, z, b, st in zip(xs, ws, zs, bs, sts)])
  torch.cuda.synchronize()
  for i in range(len(shapes)):
    assert torch.equal(zs[i], single_z[i]), i
    if sts[i] is not None:
      assert torch.equal(sts[i], single_st[i]), i
  # input gradients: dx_e = dz_e . W_e^T, two of the four with the BatchNorm-backward sums of the layer that produced x_e
  srcs, dzs = [], []
  for i, (M, N, K) in enumerate(shapes):
    dzs.append((torch.randn(M, N, generator=g) * 0.01).to(DEV))
    if i in (0, 2):  # x_i as the output of a dense + BatchNorm + ReLU layer
      zprev = torch.randn(M, K, generator=g).to(DEV)
      gamma, beta = (torch.rand(K, generator=g) + 0.5).to(DEV), (torch.randn(K, generator=g) * 0.1).to(DEV)
      y, mean, invstd = hip.bn_act_fwd(zprev, None, gamma, beta, 1, 1e-3, 0.99, torch.zeros(K, device=DEV),
                                       torch.ones(K, device=DEV), kernels.ACT_RELU)
      srcs.append(kernels.BnSource(zprev, None, y, mean, invstd, kernels.ACT_RELU, gamma, None, beta=beta))
    else:
      srcs.append(None)
  single_dx, single_part = [], []
  for i, (M, N, K) in enumerate(shapes):
    if srcs[i] is not None:
      part = torch.zeros(hip.gemm_row_tiles(M) * K * 2, device=DEV)
      single_dx.append(hip.gemm_bn_bwd(kernels.GEMM_NT, dzs[i], ws[i], srcs[i], part))
      single_part.append(part)
    else:
      single_dx.append(hip.gemm(kernels.GEMM_NT, dzs[i], ws[i]))
      single_part.append(None)
  dxs = [torch.empty_like(d) for d in single_dx]
  parts = [torch.zeros_like(p) if p is not None else None for p in single_part]
  hip.gemm_grouped(kernels.GEMM_NT, [(dzs[i], ws[i], dxs[i], None, False) + ((None, (srcs[i], parts[i])) if srcs[i] is not None else ())
                                     for i in range(len(shapes))])
  torch.cuda.synchronize()
  for i in range(len(shapes)):
    assert torch.equal(dxs[i], single_dx[i]), i
    if parts[i] is not None:
      assert torch.equal(parts[i], single_part[i]), i


def test_multi_layer_batchnorm_launches_match_single_launches(hip):
  """er_bn_fwd_multi / er_bn_bwd_multi (kernels.GroupedBNActFn): the bias / BatchNorm / activation kernels of several
  layers in one launch run the bodies of the single-layer kernels - every output bit for bit what er_bn_apply_from_stats /
  er_bn_act_fwd and er_bn_act_bwd(_from_partials, _ld) give: batch statistics from a GEMM epilogue, the moving statistics
  (ER_BN_FROZEN), bias + ReLU alone, bias alone; gradients from a column block of a wider tensor, from ready-made
  partial sums, and accumulated into buffers.  More than 8 layers: several launches."""
  g = torch.Generator().manual_seed(11)
  eps, mom = 1e-3, 0.99
  # (rows, cols, mode, act, bias?)
  spec = [(8192, 256, kernels.BN_BATCH, kernels.ACT_RELU, False), (8192, 192, kernels.BN_FROZEN, kernels.ACT_RELU, True),
          (300, 70, kernels.BN_BATCH, kernels.ACT_NONE, False), (8192, 64, kernels.BN_NONE, kernels.ACT_RELU, True),
          (130, 1, kernels.BN_NONE, kernels.ACT_NONE, True), (4096, 33, kernels.BN_FROZEN, kernels.ACT_NONE, False),
          (8192, 128, kernels.BN_BATCH, kernels.ACT_RELU, False), (64, 5, kernels.BN_NONE, kernels.ACT_RELU, False),
          (1000, 130, kernels.BN_BATCH, kernels.ACT_RELU, False), (8192, 256, kernels.BN_FROZEN, kernels.ACT_RELU, True)]
  layers, single = [], []
  for (B, N, mode, act, has_bias) in spec:
    K = 48
    x, w = torch.randn(B, K, generator=g).to(DEV), (torch.randn(K, N, generator=g) * 0.2).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) if has_bias else None
    gamma = (torch.rand(N, generator=g) + 0.5).to(DEV) if mode else None
    beta = (torch.randn(N, generator=g) * 0.1).to(DEV) if mode else None
    mm = (torch.randn(N, generator=g) * 0.1).to(DEV) if mode else None
    mv = (torch.rand(N, generator=g) + 0.5).to(DEV) if mode else None
    stats = torch.zeros(hip.gemm_row_tiles(B) * N * 3, device=DEV) if mode == kernels.BN_BATCH else None
    z = hip.gemm(kernels.GEMM_NN, x, w, col_stats=stats)
    mm1, mv1 = (mm.clone(), mv.clone()) if mode else (None, None)
    if mode == kernels.BN_BATCH:
      ref = hip.bn_apply_from_stats(z, bias, stats, hip.gemm_row_tiles(B), gamma, beta, eps, mom, mm1, mv1, act)
    else:
      ref = hip.bn_act_fwd(z, bias, gamma, beta, mode, eps, mom, mm1, mv1, act)
    single.append(ref + (mm1, mv1))
    layers.append(dict(x=z, bias=bias, gamma=gamma, beta=beta, moving_mean=mm, moving_var=mv, col_stats=stats, use_bn=mode,
                       act=act, eps=eps, momentum=mom))
  outs = hip.bn_fwd_multi(layers)
  torch.cuda.synchronize()
  for i, (l, o, r) in enumerate(zip(layers, outs, single)):
    assert torch.equal(o[0], r[0]), i
    if l['use_bn']:
      assert torch.equal(o[1], r[1]) and torch.equal(o[2], r[2]), i
      assert torch.equal(l['moving_mean'], r[3]) and torch.equal(l['moving_var'], r[4]), i
  # backward
  blayers, bsingle = [], []
  for i, (l, o) in enumerate(zip(layers, outs)):
    B, N = l['x'].shape
    y, mean, invstd = o
    kind = i % 4  # 0: plain dy; 1: a column block of a wider gradient; 2: ready-made partial sums; 3: accumulate into buffers
    mode, act = l['use_bn'], l['act']
    if kind == 1:
      wide = (torch.randn(B, N + 7, generator=g) * 0.1).to(DEV)
      dy = wide[:, 3:3 + N]
    else:
      dy = (torch.randn(B, N, generator=g) * 0.1).to(DEV)
    partial = None
    if kind == 2 and mode == kernels.BN_BATCH:
      K2 = 40
      dzn, wn = (torch.randn(B, K2, generator=g) * 0.1).to(DEV), torch.randn(N, K2, generator=g).to(DEV)
      partial = torch.empty(hip.gemm_row_tiles(B) * N * 2, device=DEV)
      dy = hip.gemm_bn_bwd(kernels.GEMM_NT, dzn, wn, kernels.BnSource(l['x'], None, y, mean, invstd, act), partial)
    into_a = into_b = None
    if kind == 3:
      base = [None if t is None else torch.randn(N, generator=g).to(DEV) for t in (l['bias'], l['gamma'], l['gamma'])]
      into_a = tuple(None if t is None else t.clone() for t in base)
      into_b = tuple(None if t is None else t.clone() for t in base)
    ref = hip.bn_act_bwd(l['x'], l['bias'], l['gamma'], y, mean, invstd, dy, mode, act, l['bias'] is not None,
                         l['gamma'] is not None, into=into_a, partial=partial)
    bsingle.append((ref, into_a))
    blayers.append(dict(x=l['x'], bias=l['bias'], gamma=l['gamma'], beta=l['beta'], y=y, mean=mean, invstd=invstd, dy=dy,
                        use_bn=mode, act=act, partial=partial, into=into_b))
  bouts = hip.bn_bwd_multi(blayers)
  torch.cuda.synchronize()
  for i, (l, o, (r, into_a)) in enumerate(zip(blayers, bouts, bsingle)):
    for a, b_ in zip(o, r):
      assert (a is None) == (b_ is None), i
      if a is not None:
        assert torch.equal(a, b_), i
    if into_a is not None:
      for a, b_ in zip(l['into'], into_a):
        if a is not None:
          assert torch.equal(a, b_), i


@pytest.mark.parametrize('B,N,K', [(4096, 256, 128), (300, 70, 33), (64, 1, 16), (130, 128, 64)])
@pytest.mark.parametrize('act', [kernels.ACT_RELU, kernels.ACT_NONE])
def test_dgrad_gemm_emits_batchnorm_backward_sums(hip, B, N, K, act):
  """er_gemm_f32_bn_bwd: the dgrad GEMM dy = dz_next . W^T also leaves the (sum g, sum g xhat) column partials of the
  layer below; er_bn_act_bwd_from_partials must then give what the two-pass er_bn_act_bwd gives from the same dy."""
  g = torch.Generator().manual_seed(B + N + K + act)
  z = torch.randn(B, N, generator=g).to(DEV)
  gamma = (torch.rand(N, generator=g) + 0.5).to(DEV)
  beta = (torch.randn(N, generator=g) * 0.1).to(DEV)
  mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
  y, mean, invstd = hip.bn_act_fwd(z, None, gamma, beta, 1, 1e-3, 0.99, mm, mv, act)
  dz_next = (torch.randn(B, K, generator=g) * 0.1).to(DEV)
  w = torch.randn(N, K, generator=g).to(DEV)
  src = kernels.BnSource(z, None, y, mean, invstd, act)
  partial = torch.empty(hip.gemm_row_tiles(B) * N * 2, device=DEV)
  dy = hip.gemm_bn_bwd(kernels.GEMM_NT, dz_next, w, src, partial)
  assert torch.equal(dy, hip.gemm(kernels.GEMM_NT, dz_next, w))
  ref = hip.bn_act_bwd(z, None, gamma, y, mean, invstd, dy, 1, act, False, True)
  got = hip.bn_act_bwd(z, None, gamma, y, mean, invstd, dy, 1, act, False, True, partial=partial)
  torch.cuda.synchronize()
  for a, b, what in zip(got, ref, ('dz', 'dbias', 'dgamma', 'dbeta')):
    if a is None:
      assert b is None
      continue
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= 2e-5 * scale, (what, float((a - b).abs().max()), scale)


@pytest.mark.parametrize('B,N,K,col0,n_src', [(4096, 81, 256, 17, 64), (300, 150, 33, 70, 80), (130, 64, 64, 0, 64),
                                              (1000, 81, 256, 80, 1), (64, 200, 16, 3, 130)])
@pytest.mark.parametrize('act', [kernels.ACT_RELU, kernels.ACT_NONE])
def test_dgrad_gemm_emits_the_sums_of_a_column_block(hip, B, N, K, col0, n_src, act):
  """er_gemm_f32_bn_bwd_cols: the layer below produced columns [col0, col0 + n_src) of the GEMM's output position only
  (DeepFM: the deep tower inside [sum(wide) | FM | deep]).  The output is the plain GEMM's, bit for bit; the partials are,
  bit for bit, those of er_gemm_f32_bn_bwd over that block alone (the same accumulators, rows and order)."""
  g = torch.Generator().manual_seed(B + N + K + col0 + act)
  z = torch.randn(B, n_src, generator=g).to(DEV)
  gamma = (torch.rand(n_src, generator=g) + 0.5).to(DEV)
  beta = (torch.randn(n_src, generator=g) * 0.1).to(DEV)
  mm, mv = torch.zeros(n_src, device=DEV), torch.ones(n_src, device=DEV)
  y, mean, invstd = hip.bn_act_fwd(z, None, gamma, beta, 1, 1e-3, 0.99, mm, mv, act)
  dz_next = (torch.randn(B, K, generator=g) * 0.1).to(DEV)
  w = torch.randn(N, K, generator=g).to(DEV)
  src = kernels.BnSource(z, None, y, mean, invstd, act)
  tiles = hip.gemm_row_tiles(B)
  part_c = torch.full((tiles * n_src * 2,), float('nan'), device=DEV)
  dy = hip.gemm_bn_bwd(kernels.GEMM_NT, dz_next, w, src, part_c, col0=col0)
  assert torch.equal(dy, hip.gemm(kernels.GEMM_NT, dz_next, w))
  part_b = torch.full((tiles * n_src * 2,), float('nan'), device=DEV)
  dy_b = hip.gemm_bn_bwd(kernels.GEMM_NT, dz_next, w[col0:col0 + n_src].contiguous(), src, part_b)
  torch.cuda.synchronize()
  assert torch.equal(dy_b, dy[:, col0:col0 + n_src])
  assert not torch.isnan(part_c).any() and torch.equal(part_c, part_b)
  # ... and they finish the block's BatchNorm backward from the strided view of dy, as the model does
  ref = hip.bn_act_bwd(z, None, gamma, y, mean, invstd, dy[:, col0:col0 + n_src], 1, act, False, True)
  got = hip.bn_act_bwd(z, None, gamma, y, mean, invstd, dy[:, col0:col0 + n_src], 1, act, False, True, partial=part_c)
  torch.cuda.synchronize()
  for a, b, what in zip(got, ref, ('dz', 'dbias', 'dgamma', 'dbeta')):
    if a is None:
      assert b is None
      continue
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= 2e-5 * scale, (what, float((a - b).abs().max()), scale)


@pytest.mark.parametrize('B,dims', [(300, (16, 1)), (5000, (4,)), (7, (8,))])
def test_route_outputs_of_the_segmented_path(hip, ref, B, dims):
  """er_emb_route without routing tables (one table per lookup -> fused sort + heads + route launches): the
  de-duplicated key list, its length and every entry's index into it must equal the oracle's."""
  rng = np.random.default_rng(B)
  sc, sd, stc, std, oc, od, group_rows = _make_lookup_problem(rng, B, dims, DEV)
  for dim, total in group_rows.items():
    gc = ref.emb_group_create([s for s in sc if s.dim == dim], dim, total, stc[dim], None, None, None)
    gd = hip.emb_group_create([s for s in sd if s.dim == dim], dim, total, std[dim], None, None, None)
    n = gd['num_entries']
    uk_c, nu_c, ui_c = torch.zeros(n, dtype=torch.int32), torch.zeros(1, dtype=torch.int32), torch.zeros(n, dtype=torch.int64)
    uk_d, nu_d, ui_d = uk_c.to(DEV), nu_c.to(DEV), ui_c.to(DEV)
    ref.emb_route(gc, uk_c, nu_c, ui_c, None)
    for _ in range(2):
      hip.emb_route(gd, uk_d, nu_d, ui_d, None)
    torch.cuda.synchronize()
    k = int(nu_c.item())
    assert int(nu_d.item()) == k
    assert torch.equal(uk_d[:k].cpu(), uk_c[:k])
    assert torch.equal(ui_d.cpu(), ui_c)
    hip.emb_group_destroy(gd)


@pytest.mark.parametrize('B,F,D', [(300, 27, 16), (5, 2, 3), (64, 9, 64), (1, 40, 8)])
@pytest.mark.parametrize('itself', [False, True])
def test_dot_interaction(hip, ref, B, F, D, itself):
  """er_dot_interaction_fwd / _bwd against the einsum + upper-triangle restatement of model/dlrm.py:44-57 and
  against autograd of that formula."""
  g = torch.Generator().manual_seed(B * 7 + F + D)
  x = torch.randn(B, F * D, generator=g)
  out = hip.dot_interaction_fwd(x.to(DEV), F, D, itself)
  exp = ref.dot_interaction_fwd(x, F, D, itself)
  assert out.shape == exp.shape == (B, F * (F - 1) // 2 + (F if itself else 0))
  assert torch.allclose(out.cpu(), exp, rtol=1e-5, atol=1e-5)
  go = torch.randn(exp.shape, generator=g)
  dx = hip.dot_interaction_bwd(x.to(DEV), go.to(DEV), F, D, itself)
  xr = x.clone().requires_grad_(True)
  e = xr.reshape(B, F, D)
  inter = torch.einsum('bne,bme->bnm', e, e)
  off = 0 if itself else 1
  torch.cat([inter[:, i, i + off:F] for i in range(F)], dim=1).backward(go)
  assert torch.allclose(dx.cpu(), xr.grad, rtol=1e-5, atol=1e-5)
  assert torch.allclose(ref.dot_interaction_bwd(x, go, F, D, itself), xr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('n,nt', [(4096, 200), (100000, 200), (5, 3)])
def test_auc_update_counts_bit_exact(hip, ref, n, nt):
  """er_auc_update: the (label, #thresholds below the prediction) histogram, integer-exact and order-independent."""
  from easyrec_amd.core.metrics import auc_thresholds, auc_from_counts
  rng = np.random.default_rng(n + nt)
  y = torch.from_numpy((rng.random(n) < 0.25).astype(np.float32))
  p = torch.from_numpy(np.clip(rng.normal(0.5, 0.3, size=n), 0, 1).astype(np.float32))
  p[::5] = torch.from_numpy(auc_thresholds(nt)[rng.integers(0, nt, size=len(p[::5]))])  # exactly on thresholds
  w = torch.from_numpy((rng.random(n) < 0.9).astype(np.float32))
  t = torch.from_numpy(auc_thresholds(nt))
  cc = torch.zeros(2, nt + 1, dtype=torch.int64)
  cd = torch.zeros(2, nt + 1, dtype=torch.int64, device=DEV)
  ref.auc_update(p[:min(n, 3000)], y[:min(n, 3000)], w[:min(n, 3000)], t, cc)
  hip.auc_update(p[:min(n, 3000)].to(DEV), y[:min(n, 3000)].to(DEV), w[:min(n, 3000)].to(DEV), t.to(DEV), cd)
  torch.cuda.synchronize()
  assert torch.equal(cd.cpu(), cc)
  # streaming the whole set in two calls == one numpy pass
  cd.zero_()
  h = n // 2
  hip.auc_update(p[:h].to(DEV), y[:h].to(DEV), None, t.to(DEV), cd)
  hip.auc_update(p[h:].to(DEV), y[h:].to(DEV), None, t.to(DEV), cd)
  bucket = (p.numpy()[:, None] > t.numpy()[None, :]).sum(axis=1)
  exp = np.zeros((2, nt + 1), dtype=np.int64)
  np.add.at(exp, ((y.numpy() != 0).astype(np.int64), bucket), 1)
  assert np.array_equal(cd.cpu().numpy(), exp)
  assert 0.0 <= auc_from_counts(exp) <= 1.0


@pytest.mark.parametrize('opt', [kernels.OPT_SGD, kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM, kernels.OPT_ADAGRAD])
def test_replicated_tables_dense_reduce_and_apply(hip, ref, opt):
  """er_emb_bwd_reduce_dense (two groups in one launch, the second sharing the first's sort) must write what
  er_emb_bwd_reduce + er_scatter_unique write, bit for bit; er_emb_dense_apply must equal er_emb_bwd_update over the
  rows with a count (+ the dense-decay sweep of TF-exact Adam), which is how the oracle restates it."""
  rng = np.random.default_rng(11 + opt)
  B, n_feat = 700, 5
  rows = [int(rng.integers(3, 90)) for _ in range(n_feat)]
  rows[2] = 1
  ids = [rng.integers(-1, r, size=B).astype(np.int64) for r in rows]
  ids[0][:300] = 1  # a hot row: runs across tiles
  base = np.concatenate([[0], np.cumsum(rows)]).astype(int)
  total = int(base[-1])
  groups_d, groups_c, dense_d, dense_c, tabs = [], [], [], [], []
  ids_on = {dev: [torch.from_numpy(i).to(dev) for i in ids] for dev in (DEV, 'cpu')}  # one ids tensor per feature
  for dim in (1, 16):
    dout = torch.from_numpy((rng.standard_normal((B, n_feat * dim)) * 0.01).astype(np.float32))
    var = torch.from_numpy((rng.standard_normal((total, dim)) * 0.1).astype(np.float32))
    m = torch.from_numpy((rng.random((total, dim)) * 0.01).astype(np.float32))
    v = torch.from_numpy((rng.random((total, dim)) * 0.01 + 1e-4).astype(np.float32))
    tabs.append((var, m, v))
    for dev, groups, be in ((DEV, groups_d, hip), ('cpu', groups_c, ref)):
      vd, od = var.to(dev), dout.to(dev)
      specs = [kernels.LookupSpec(table=vd[base[f]:base[f + 1]], ids=ids_on[dev][f], offsets=None, weights=None, out=od,
                                  out_col=f * dim, rows=rows[f], key_base=int(base[f]), dim=dim, combiner=0, n_rows=B,
                                  max_nnz=B) for f in range(n_feat)]
      groups.append(be.emb_group_create(specs, dim, total, var.to(dev), None, None, None))
    dense_d.append(torch.zeros(total, dim + 1, device=DEV))
    dense_c.append(torch.zeros(total, dim + 1))
  assert hip.emb_group_share_sort(groups_d[1], groups_d[0])
  hip.emb_bwd_reduce_dense(groups_d, dense_d)
  # the two-step form on the device, and the oracle
  for g, dd, dc, gc in zip(groups_d, dense_d, dense_c, groups_c):
    two = torch.zeros_like(dd)
    keys, grads, n = hip.emb_bwd_reduce(g)
    hip.scatter_unique(keys, grads, n, keys.numel(), g['dim'], two)
    torch.cuda.synchronize()
    assert torch.equal(dd, two), g['dim']
    ref.emb_bwd_reduce_dense([gc], [dc])
    assert torch.allclose(dd.cpu(), dc, rtol=1e-5, atol=1e-7)  # runs of 300+ terms are summed piece-wise
    assert torch.equal(dd.cpu()[:, -1], dc[:, -1])
  # apply: both tables in one launch, twice (untouched rows keep decaying under TF-exact Adam)
  hyper = _hyper(lr=0.05, t=3, gscale=0.5)
  dev_t = [(var.to(DEV), m.to(DEV), v.to(DEV), dd) for (var, m, v), dd in zip(tabs, dense_d)]
  cpu_t = [(var.clone(), m.clone(), v.clone(), dd.cpu()) for (var, m, v), dd in zip(tabs, dense_d)]
  for _ in range(2):
    hip.emb_dense_apply(dev_t, opt, hyper.to(DEV))
    ref.emb_dense_apply(cpu_t, opt, hyper)
  torch.cuda.synchronize()
  for a, b in zip(dev_t, cpu_t):
    for x, y, what in zip(a[:3], b[:3], ('var', 'm', 'v')):
      assert torch.allclose(x.cpu(), y, rtol=1e-6, atol=1e-9), (what, float((x.cpu() - y).abs().max()))
  for g in groups_d:
    hip.emb_group_destroy(g)


def _bf16_round(x):
  return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize('rows,cols', [(4096, 624), (130, 70), (7, 3), (624, 624)])
def test_cast_bf16_plain_and_transposed(hip, rows, cols):
  """er_cast_bf16 against torch's own round-to-nearest-even cast; the padding up to the leading dimension is zero."""
  g = torch.Generator().manual_seed(rows * 7 + cols)
  x = (torch.randn(rows, cols, generator=g) * 3).to(DEV)
  x[0, 0] = float('inf')
  x[rows - 1, cols - 1] = 1.00390625  # a tie: rounds to even
  pad = kernels.Bf16Shadows.pad8
  plain = torch.full((rows, pad(cols)), 7.0, dtype=torch.bfloat16, device=DEV)
  tr = torch.full((cols, pad(rows)), 7.0, dtype=torch.bfloat16, device=DEV)
  hip.cast_bf16([(x, plain, False), (x, tr, True)])
  torch.cuda.synchronize()
  want = x.to(torch.bfloat16)
  assert torch.equal(plain[:, :cols], want)
  assert torch.equal(tr[:, :rows], want.t())
  assert (plain[:, cols:] == 0).all() and (tr[:, rows:] == 0).all()


@pytest.mark.parametrize('M,N,K', [(4096, 624, 624), (4096, 256, 624), (130, 70, 40), (128, 128, 64), (257, 129, 520),
                                   (64, 8, 8)])
@pytest.mark.parametrize('mode', ['plain', 'bias_accumulate', 'bf16_out'])
def test_gemm_bf16_nt(hip, M, N, K, mode):
  """er_gemm_bf16_nt: bf16 operands in HBM, fp32 accumulate.  The products of bf16 values are exact in fp32, so the
  reference (fp64 sum of the same bf16-rounded operands) differs only by fp32 summation order: 2e-5 of sum |a||b|."""
  hip._ck(hip.lib.er_gemm_bf16_nt_prepare(), 'prepare')
  g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
  a = torch.randn(M, K, generator=g).to(DEV)
  b = torch.randn(N, K, generator=g).to(DEV)  # Bt: asymmetric by construction
  pad = kernels.Bf16Shadows.pad8
  a16 = torch.empty(M, pad(K), dtype=torch.bfloat16, device=DEV)
  b16 = torch.empty(N, pad(K), dtype=torch.bfloat16, device=DEV)
  hip.cast_bf16([(a, a16, False), (b, b16, False)])
  ref = _bf16_round(a).double() @ _bf16_round(b).double().t()
  scale = _bf16_round(a).abs().double() @ _bf16_round(b).abs().double().t()
  if mode == 'plain':
    out = torch.full((M, N), 3.0, device=DEV)
    hip.gemm_bf16_nt(a16, b16, M, N, K, out=out)
  elif mode == 'bias_accumulate':
    bias = torch.randn(N, generator=g).to(DEV)
    base = torch.randn(M, N, generator=g).to(DEV)
    out = base.clone()
    hip.gemm_bf16_nt(a16, b16, M, N, K, out=out, bias=bias, accumulate=True)
    ref = ref + bias.double()[None, :] + base.double()
  else:
    out16 = torch.zeros(M, pad(N), dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(M, N, device=DEV)
    hip.gemm_bf16_nt(a16, b16, M, N, K, out=out, out_bf16=out16)
    torch.cuda.synchronize()
    assert torch.equal(out16[:, :N], out.to(torch.bfloat16)), 'the bf16 copy is the RNE rounding of the fp32 output'
  torch.cuda.synchronize()
  err = (out.double() - ref).abs()
  assert bool((err <= 2e-5 * scale + 1e-6).all()), float((err / (scale + 1e-9)).max())


@pytest.mark.parametrize('B,H0,D,sizes', [(37, 5, 8, (6, 4)), (128, 17, 16, (64, 64, 64)), (3, 2, 4, (1,))])
def test_cin_layers_against_einsum(hip, B, H0, D, sizes):
  """kernels.CINFn (er_cin_outer_fwd + er_gemm_f32 + er_cin_act_pool_fwd per layer, and their backward) against the
  formula of layers/keras/interaction.py:385-409 written with einsum in fp64."""
  g = torch.Generator().manual_seed(B * 131 + H0)
  x0 = torch.randn(B, H0, D, generator=g)
  hs = [H0] + list(sizes)
  ws = [torch.randn(hs[k + 1], hs[k], H0, generator=g) * 0.3 for k in range(len(sizes))]
  bs = [torch.randn(hs[k + 1], generator=g) * 0.1 for k in range(len(sizes))]
  dout = torch.randn(B, sum(sizes), generator=g)
  # reference
  xr = x0.double().requires_grad_(True)
  wr = [w.double().requires_grad_(True) for w in ws]
  br = [b.double().requires_grad_(True) for b in bs]
  xi, pooled = xr, []
  for w, b in zip(wr, br):
    fm = torch.relu(torch.einsum('bhd,bmd,nhm->bnd', xi, xr, w) + b[None, :, None])
    pooled.append(fm.sum(-1))
    xi = fm
  ref = torch.cat(pooled, dim=-1)
  ref.backward(dout.double())
  # device
  xd = x0.to(DEV).requires_grad_(True)
  wd = [w.to(DEV).requires_grad_(True) for w in ws]
  bd = [b.to(DEV).requires_grad_(True) for b in bs]
  wg = [torch.zeros_like(w) for w in wd]
  bg = [torch.zeros_like(b) for b in bd]
  out = kernels.CINFn.apply(xd, len(sizes), *wd, *bd, *wg, *bg)
  out.backward(dout.to(DEV))
  torch.cuda.synchronize()
  def close(got, want):  # fp32 sums of up to B * D * H * H0 terms: 2e-5 of the tensor's scale
    want = want.detach()
    return float((got.detach().cpu().double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6

  assert close(out, ref)
  assert close(xd.grad, xr.grad)
  for k in range(len(sizes)):
    assert close(wg[k], wr[k].grad), k
    assert close(bg[k], br[k].grad), k


# ------------------------------------------------------------------------------------------- fused binary head
@pytest.mark.parametrize('B,K,with_src,with_bn,bias', [(4096, 64, True, True, True), (300, 64, True, False, True),
                                                      (130, 16, False, False, False), (130, 8, True, True, True), (1000, 256, True, True, True),
                                                      (64, 128, False, False, True)])
def test_head_sigmoid_ce_and_loss_tail(hip, ref, B, K, with_src, with_bn, bias):
  """er_head_sigmoid_ce (dense(K -> 1) + sigmoid cross entropy + the projection's backward + the producing layer's
  BatchNorm-backward column sums, one launch) and er_loss_tail (partial-sum losses + column-sum jobs) against the stand-in's
  torch restatement: logits / probs / dlogits / dx 1e-5 relative, partial sums 1e-5 of their scale; ragged last tile."""
  rng = np.random.default_rng(B + K)
  ld = K + 8  # (a row stride wider than K: the head reads a column block in place)
  xbuf = torch.from_numpy(rng.standard_normal((B, ld)).astype(np.float32))
  z = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32))
  mean = torch.from_numpy(rng.standard_normal(K).astype(np.float32) * 0.1)
  invstd = torch.from_numpy((0.5 + rng.random(K)).astype(np.float32))
  w = torch.from_numpy((rng.standard_normal((K, 1)) * 0.3).astype(np.float32))
  b = torch.from_numpy(rng.standard_normal(1).astype(np.float32)) if bias else None
  y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32))
  scale = 0.7

  def run(be, dev):
    xd = xbuf.to(dev)
    x = xd[:, :K]
    src = None
    if with_src:
      src = kernels.BnSource(z.to(dev), None, x, mean.to(dev) if with_bn else None, invstd.to(dev) if with_bn else None,
                             kernels.ACT_RELU)
    out = be.head_sigmoid_ce(x, w.to(dev), None if b is None else b.to(dev), y.to(dev), scale, src=src)
    loss = torch.zeros(1, device=dev)
    loss._er_partials = (out['loss_partials'], scale, float(B))
    wg = torch.full((K,), 0.25, device=dev)
    bg = torch.full((1,), -0.5, device=dev)
    reg, total, report = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    other = torch.full((1,), 0.125, device=dev)
    embp = torch.arange(7, dtype=torch.float32, device=dev)
    be.loss_tail(embp, 0.5, None, [loss, other], [report, torch.zeros(1, device=dev)], reg, total,
                 jobs=[(out['wb_partials'], wg, K), (out['wb_partials'][:, K:], bg, 1)])
    if dev != 'cpu':
      torch.cuda.synchronize()
    res = {k: (v.cpu() if v is not None else None) for k, v in out.items()}
    res.update(loss=loss.cpu(), wg=wg.cpu(), bg=bg.cpu(), reg=reg.cpu(), total=total.cpu(), report=report.cpu())
    return res

  got, exp = run(hip, DEV), run(ref, 'cpu')
  for k in ('logits', 'probs', 'dlogits', 'dx'):
    assert torch.allclose(got[k].reshape(-1), exp[k].reshape(-1), rtol=1e-5, atol=2e-6), k  # (a 64..256-term dot product in another order)
  for k in ('loss_partials', 'wb_partials', 'bn_partials', 'wg', 'bg'):
    if exp[k] is None:
      assert got[k] is None
      continue
    s = float(exp[k].abs().max()) + 1e-12
    assert float((got[k] - exp[k]).abs().max()) <= 1e-5 * s, (k, float((got[k] - exp[k]).abs().max()), s)
  for k in ('loss', 'reg', 'total', 'report'):
    assert abs(float(got[k]) - float(exp[k])) <= 2e-6 * max(1.0, abs(float(exp[k]))), (k, float(got[k]), float(exp[k]))
  assert float(exp['reg']) == 0.5 * 21.0 and abs(float(exp['total']) - (10.5 + float(exp['loss']) + 0.125)) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('B,width,ld,has_base,lam', [(4096, 624, 624, True, 5e-5), (8192, 1040, 1040, True, 1e-4),
                                                     (130, 81, 84, True, 5e-5), (64, 32, 36, False, 0.25),
                                                     (64, 32, 32, True, 0.0), (7, 12, 12, False, 0.0)])
def test_group_grad_finish_base_plus_lambda_out(B, width, ld, has_base, lam):
  """er_group_grad_finish without deferred terms - dout = [dout] + lambda * out (the embedding-output L2 of
  layers/input_layer.py:369-375) - on 16-byte aligned rows (the 4-columns-per-lane form) and on others (per element)."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(B + width)
  dbuf = torch.randn(B, ld, generator=g).to(DEV)
  obuf = torch.randn(B, ld, generator=g).to(DEV)
  dout, out = dbuf[:, :width], obuf[:, :width]
  exp = (dout.clone() if has_base else torch.zeros_like(dout))
  if lam != 0.0:
    exp = exp + lam * out
  pad_before = dbuf[:, width:].clone()
  hip.group_grad_finish([(dout, out, lam, has_base, [])])
  assert torch.equal(dout, exp)
  assert torch.equal(dbuf[:, width:], pad_before)


@pytest.mark.gpu
@pytest.mark.parametrize('B,widths', [(4096, (624, 256)), (8192, (256, 256, 256, 256, 16)), (130, (81, 64)), (7, (4, 8, 12)),
                                      (1, (128,) * 8)])
def test_concat_cols_equals_torch_cat(B, widths):
  """er_concat_cols (reference tf.concat(axis=1), model/deepfm.py:75-83): the 16-byte-lane form for aligned blocks and the
  per-element form otherwise are both exactly torch.cat; parts may be column views of wider buffers."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(B + len(widths))
  parts = []
  for k, w in enumerate(widths):
    buf = torch.randn(B, w + 4 * (k % 2), generator=g).to(DEV)
    parts.append(buf[:, :w])
  out = hip.concat_cols(parts)
  assert torch.equal(out, torch.cat(parts, dim=1))



@pytest.mark.parametrize('B,n_w,F,D,n_d', [(4096, 39, 39, 16, 64), (100, 5, 7, 3, 9)])
def test_wide_fm_concat_equals_the_three_launches(B, n_w, F, D, n_d):
  """er_wide_fm_concat = er_rowsum_fwd + er_fm_fwd + the concat, bit for bit (the same bodies as workgroup ranges)."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(B)
  wide_full = torch.randn(B, n_w + 3, generator=g).to(DEV)  # (a column block of a wider group output)
  x_full = torch.randn(B, F * D + (4 if D % 4 == 0 else 1), generator=g).to(DEV)
  deep = torch.randn(B, n_d, generator=g).to(DEV)
  wide, x = wide_full[:, :n_w], x_full[:, :F * D]
  out, S = hip.wide_fm_concat(wide, x, F, D, deep)
  fm, S2 = hip.fm_fwd(x, F, D)
  want = torch.cat([hip.rowsum_fwd(wide, n_w), fm, deep], dim=1)
  assert out.shape == want.shape and out.stride(0) % 4 == 0
  assert torch.equal(out, want) and torch.equal(S, S2)


@pytest.mark.gpu
@pytest.mark.parametrize('B,K,N,n_w,F,D', [(4096, 128, 64, 39, 39, 16), (8192, 64, 128, 5, 7, 8), (100, 32, 20, 5, 7, 3)])
def test_bn_apply_wide_fm_equals_the_two_launches(B, K, N, n_w, F, D):
  """er_bn_apply_wide_fm (the deep tower's last BatchNorm finalize + apply inside DeepFM's [sum(wide) | FM | deep] launch) =
  er_bn_apply_from_stats followed by er_wide_fm_concat, bit for bit: activations, saved statistics, moving statistics, the
  concat and the FM field sums."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(B + N)
  x = torch.randn(B, K, generator=g).to(DEV)
  w = (torch.randn(K, N, generator=g) * 0.3).to(DEV)
  bias = torch.randn(N, generator=g).to(DEV)
  gamma = (torch.rand(N, generator=g) + 0.5).to(DEV)
  beta = torch.randn(N, generator=g).to(DEV)
  wide_full = torch.randn(B, n_w + 3, generator=g).to(DEV)
  x_full = torch.randn(B, F * D + (4 if D % 4 == 0 else 1), generator=g).to(DEV)
  wide, fx = wide_full[:, :n_w], x_full[:, :F * D]
  chunks = hip.gemm_row_tiles(B)
  stats = torch.empty(chunks * N * 3, device=DEV)
  z = hip.gemm(kernels.GEMM_NN, x, w, bias=bias, col_stats=stats)
  mm1, mv1 = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
  y1, mean1, inv1 = hip.bn_apply_from_stats(z, None, stats, chunks, gamma, beta, 1e-3, 0.99, mm1, mv1, kernels.ACT_RELU)
  out1, S1 = hip.wide_fm_concat(wide, fx, F, D, y1)
  mm2, mv2 = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
  pend = dict(z=z, stats=stats, chunks=chunks, gamma=gamma, beta=beta, eps=1e-3, momentum=0.99, moving_mean=mm2, moving_var=mv2,
              act=kernels.ACT_RELU, y=torch.empty_like(z), mean=torch.empty(N, device=DEV), invstd=torch.empty(N, device=DEV))
  res = hip.bn_apply_wide_fm(pend, wide, fx, F, D)
  assert res is not None
  out2, S2 = res
  assert torch.equal(pend['y'], y1) and torch.equal(pend['mean'], mean1) and torch.equal(pend['invstd'], inv1)
  assert torch.equal(mm2, mm1) and torch.equal(mv2, mv1)
  assert out2.shape == out1.shape and torch.equal(out2, out1) and torch.equal(S2, S1)


@pytest.mark.gpu
@pytest.mark.parametrize('M,K0,K,N,act', [(204800, 128, 128, 64, 'relu'), (20000, 64, 64, 32, 'relu'), (16500, 48, 36, 128, 'none'),
                                         (777, 32, 256, 200, 'relu'), (4096, 40, 8, 4, 'relu'), (204800, 64, 32, 1, 'relu'), (5000, 16, 12, 3, 'none')])
def test_gemm_with_the_batchnorm_apply_in_its_staging_equals_the_two_launches(M, K0, K, N, act):
  """er_bn_finalize_from_stats + er_gemm_f32_bn_a (a tall layer's BatchNorm finalize as a launch of its own, its apply inside the
  NEXT layer's contraction while the A tiles are staged; reference layers/dnn.py:57-79) = er_bn_apply_from_stats followed by
  er_gemm_f32, bit for bit: the activations left behind for the backward, saved and moving statistics, the contraction's
  output and its column statistics.  (M > 16384: the statistics take the merge launch first, as in DIN's attention MLP.)"""
  hip = kernels.hip()
  a = kernels.ACT_RELU if act == 'relu' else kernels.ACT_NONE
  g = torch.Generator().manual_seed(M + K + N)
  x = torch.randn(M, K0, generator=g).to(DEV)
  w0 = (torch.randn(K0, K, generator=g) * 0.3).to(DEV)
  b0 = torch.randn(K, generator=g).to(DEV)
  gamma = (torch.rand(K, generator=g) + 0.5).to(DEV)
  beta = torch.randn(K, generator=g).to(DEV)
  w1 = (torch.randn(K, N, generator=g) * 0.2).to(DEV)
  b1 = torch.randn(N, generator=g).to(DEV)
  chunks = hip.gemm_row_tiles(M)
  stats = torch.empty(chunks * K * 3, device=DEV)
  z = hip.gemm(kernels.GEMM_NN, x, w0, bias=b0, col_stats=stats)
  mm1, mv1 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
  y1, mean1, inv1 = hip.bn_apply_from_stats(z, None, stats, chunks, gamma, beta, 1e-3, 0.99, mm1, mv1, a)
  s1 = torch.zeros(chunks * N * 3, device=DEV)
  out1 = hip.gemm(kernels.GEMM_NN, y1, w1, bias=b1, col_stats=s1)
  mm2, mv2 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
  pend = dict(z=z, stats=stats, chunks=chunks, gamma=gamma, beta=beta, eps=1e-3, momentum=0.99, moving_mean=mm2, moving_var=mv2,
              act=a, y=torch.full_like(z, float('nan')), mean=torch.empty(K, device=DEV), invstd=torch.empty(K, device=DEV))
  assert hip.bn_a_ok(pend, w1)
  s2 = torch.zeros(chunks * N * 3, device=DEV)
  out2 = hip.gemm_bn_a(pend, w1, b1, col_stats=s2)
  torch.cuda.synchronize()
  assert torch.equal(pend['mean'], mean1) and torch.equal(pend['invstd'], inv1)
  assert torch.equal(mm2, mm1) and torch.equal(mv2, mv1)
  assert torch.equal(pend['y'], y1)
  assert torch.equal(out2, out1) and torch.equal(s2, s1)


@pytest.mark.gpu
def test_narrow_column_sums_of_several_matrices_in_one_launch_equal_the_single_launches():
  """er_colsum_narrow_multi (the bias gradients of a multi-task model's tower heads [B, 1] and gates [B, experts] in ONE
  launch) = er_colsum_acc per matrix, bit for bit, accumulating and overwriting; column blocks of wider tensors included."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(5)
  shapes = [(8192, 1), (8192, 1), (8192, 4), (8192, 8), (5000, 3), (1, 2), (300, 1)] + [(777, 2)] * 12  # (> 16 jobs: two launches)
  xs = []
  for rows, cols in shapes:
    full = torch.randn(rows, cols + 3, generator=g).to(DEV)
    xs.append(full[:, 1:1 + cols])
  for acc in (True, False):
    base = [torch.randn(x.shape[1], generator=g).to(DEV) for x in xs]
    want = [b.clone() for b in base]
    for x, w in zip(xs, want):
      assert hip.colsum_is_narrow(x)
      hip.colsum(x, out=w, accumulate=acc)
    got = [b.clone() for b in base]
    hip.colsum_narrow_multi(list(zip(xs, got)), accumulate=acc)
    torch.cuda.synchronize()
    for a, b, x in zip(got, want, xs):
      assert torch.equal(a, b)
      assert torch.allclose(a - (0 if not acc else base[0] * 0), a)  # (finite)
    ref = xs[2].double().sum(0)
    assert float(((got[2].double() - (base[2].double() if acc else 0)) - ref).abs().max()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('M,K,N', [(204800, 32, 1), (20000, 64, 4), (16500, 8, 2), (17000, 12, 3), (30000, 256, 1), (16385, 4, 1)])
def test_tall_narrow_projection_and_its_batchnorm_in_staging_form(M, K, N):
  """er_gemv_f32_bn_a - a tall projection onto <= 4 columns (DIN's attention scores, reference model/multi_tower_din.py:80-84)
  without MFMA tiles - against a float64 product (1e-5 of the output's scale: another summation order than the tile kernel's),
  and its form with the producing layer's BatchNorm apply on the way against apply-then-project through the same kernel, bit
  for bit (activations, statistics, moving statistics, output)."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(M + K + N)
  xbuf = torch.randn(M, K + 4, generator=g).to(DEV)
  x = xbuf[:, :K]  # (a column block: row pitch K + 4)
  w = (torch.randn(K, N, generator=g) * 0.3).to(DEV)
  bias = torch.randn(N, generator=g).to(DEV)
  assert hip.gemv_ok(x, M, N, K)
  out = hip.gemm(kernels.GEMM_NN, x, w, bias=bias)
  ref = x.double() @ w.double() + bias.double()
  assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
  # the same columns through the MFMA tiles (col_stats asked for): the two kernels agree to rounding
  st = torch.zeros(hip.gemm_row_tiles(M) * N * 3, device=DEV)
  out_t = hip.gemm(kernels.GEMM_NN, x, w, bias=bias, col_stats=st)
  assert float((out - out_t).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
  # BatchNorm of the producing layer applied on the way
  gamma = (torch.rand(K, generator=g) + 0.5).to(DEV)
  beta = torch.randn(K, generator=g).to(DEV)
  z = x.contiguous()
  chunks = hip.gemm_row_tiles(M)
  stats = torch.empty(chunks * K * 3, device=DEV)
  # column statistics of z as a GEMM epilogue would have written them: z = z . I
  z2 = hip.gemm(kernels.GEMM_NN, z, torch.eye(K, device=DEV), col_stats=stats)
  assert torch.equal(z2, z)
  mm1, mv1 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
  y1, mean1, inv1 = hip.bn_apply_from_stats(z, None, stats, chunks, gamma, beta, 1e-3, 0.99, mm1, mv1, kernels.ACT_RELU)
  out1 = hip.gemm(kernels.GEMM_NN, y1, w, bias=bias)
  mm2, mv2 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
  pend = dict(z=z, stats=stats, chunks=chunks, gamma=gamma, beta=beta, eps=1e-3, momentum=0.99, moving_mean=mm2, moving_var=mv2,
              act=kernels.ACT_RELU, y=torch.full_like(z, float('nan')), mean=torch.empty(K, device=DEV),
              invstd=torch.empty(K, device=DEV))
  assert hip.bn_a_ok(pend, w)
  out2 = hip.gemm_bn_a(pend, w, bias)
  torch.cuda.synchronize()
  assert torch.equal(pend['mean'], mean1) and torch.equal(pend['invstd'], inv1) and torch.equal(mm2, mm1) and torch.equal(mv2, mv1)
  assert torch.equal(pend['y'], y1) and torch.equal(out2, out1)


@pytest.mark.gpu
@pytest.mark.parametrize('rows,K,N', [(204800, 32, 1), (20000, 64, 4), (16500, 8, 2), (17000, 12, 3), (30000, 256, 1), (16385, 4, 1)])
def test_weight_gradient_of_a_tall_narrow_projection(rows, K, N):
  """er_wgrad_tall_narrow: dW [K, N <= 4] (+)= x^T . dz over >> K rows (DIN's attention score layer) against a float64 product
  (2e-5 of the gradient's scale), overwriting and accumulating into a column block; run twice: the same bits."""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(rows + K + N)
  xbuf = torch.randn(rows, K + 4, generator=g).to(DEV)
  x = xbuf[:, :K]
  dzbuf = (torch.randn(rows, N + 2, generator=g) * 0.1).to(DEV)
  dz = dzbuf[:, 1:1 + N]
  assert hip.wgrad_tall_narrow_ok(x, dz)
  ref = x.double().t() @ dz.double()
  scale = max(1e-6, float(ref.abs().max()))
  out = torch.full((K, N), float('nan'), device=DEV)
  hip.wgrad_tall_narrow(x, dz, out, accumulate=False)
  assert float((out.double() - ref).abs().max()) <= 2e-5 * scale
  wide = torch.randn(K, N + 5, generator=g).to(DEV)
  base = wide.clone()
  hip.wgrad_tall_narrow(x, dz, wide[:, 2:2 + N], accumulate=True)
  assert torch.equal(wide[:, :2], base[:, :2]) and torch.equal(wide[:, 2 + N:], base[:, 2 + N:])
  assert float(((wide[:, 2:2 + N] - base[:, 2:2 + N]).double() - ref).abs().max()) <= 2e-5 * scale + 1e-6
  out2 = torch.empty(K, N, device=DEV)
  hip.wgrad_tall_narrow(x, dz, out2, accumulate=False)
  torch.cuda.synchronize()
  assert torch.equal(out, out2)
  # ... with the bias gradient (the column sums of dz) from the same pass
  out3, db = torch.empty(K, N, device=DEV), torch.full((N,), 0.5, device=DEV)
  hip.wgrad_tall_narrow(x, dz, out3, accumulate=False, bias_grad=db)
  torch.cuda.synchronize()
  assert torch.equal(out3, out)
  bref = dz.double().sum(0)
  assert float((db.double() - bref).abs().max()) <= 2e-5 * max(1e-6, float(bref.abs().max()), float(dz.abs().sum(0).max()) * 1e-2)
  db2 = torch.full((N,), 0.5, device=DEV)
  out4 = out.clone()
  hip.wgrad_tall_narrow(x, dz, out4, accumulate=True, bias_grad=db2)
  torch.cuda.synchronize()
  assert float(((db2 - 0.5).double() - bref).abs().max()) <= 2e-5 * max(1e-6, float(bref.abs().max()), float(dz.abs().sum(0).max()) * 1e-2) + 1e-6


def _misaligned(t):
  """a copy of t whose base address is 4 bytes past a 16-byte boundary (the library then takes its generic fetch path)"""
  buf = torch.empty(t.numel() + 5, dtype=t.dtype, device=t.device)
  off = 1 + ((4 - (buf.data_ptr() // 4) % 4) % 4)
  assert (buf.data_ptr() + 4 * off) % 16 == 4
  out = buf[off:off + t.numel()].view(t.shape)
  out.copy_(t)
  return out


@pytest.mark.gpu
@pytest.mark.parametrize('M,K,N,ldk', [(4096, 256, 128, 256), (4096, 128, 64, 128), (4096, 81, 256, 84), (100, 64, 37, 64),
                                      (1000, 224, 200, 224), (333, 32, 64, 40)])
def test_vector_fetch_gemm_equals_the_generic_fetch_bit_for_bit(M, K, N, ldk):
  """The contraction's 16-byte fetch path (aligned operands) against its generic fetch path, which an operand 4 bytes off
  a 16-byte boundary selects: the forward contraction with bias and BatchNorm column statistics (NN) and the input-gradient
  contraction with the BatchNorm-backward column sums in its epilogue (NT) - same fragments, same order, every output
  bit-identical.  (Written for round 6's panel form of short contractions, which was measured slower and removed.)"""
  hip = kernels.hip()
  g = torch.Generator().manual_seed(M + K + N)
  xbuf = torch.randn(M, ldk, generator=g).to(DEV)  # (K of its ldk columns: the padding must not be read as operand)
  xbuf[:, K:] = float('nan')
  x = xbuf[:, :K]
  w = (torch.randn(K, N, generator=g) * 0.2).to(DEV)
  bias = torch.randn(N, generator=g).to(DEV)
  chunks = hip.gemm_row_tiles(M)
  s1, s2 = torch.zeros(chunks * N * 3, device=DEV), torch.zeros(chunks * N * 3, device=DEV)
  if N % 4 == 0:
    z1 = hip.gemm(kernels.GEMM_NN, x, w, bias=bias, col_stats=s1)
    z2 = hip.gemm(kernels.GEMM_NN, x, _misaligned(w), bias=bias, col_stats=s2)
    assert torch.equal(z1, z2) and torch.equal(s1, s2)
    ref = x.double() @ w.double() + bias.double()
    assert float((z1.double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
  # input gradient: dx = dz . w^T (NT, contraction over N) with the BatchNorm-backward sums of x's producer
  if N <= 256 and N % 4 == 0 and K % 4 == 0:
    dz = torch.randn(M, N, generator=g).to(DEV)
    zsrc = torch.randn(M, K, generator=g).to(DEV)
    ysrc = torch.relu(zsrc + 0.1)
    mean, inv = torch.randn(K, generator=g).to(DEV) * 0.1, (torch.rand(K, generator=g) + 0.5).to(DEV)
    wk = w.contiguous()
    outs = []
    for wt in (wk, _misaligned(wk)):
      src = kernels.BnSource(zsrc, None, ysrc, mean, inv, kernels.ACT_RELU)
      part = torch.zeros(chunks * K * 2, device=DEV)
      dx = hip.gemm_bn_bwd(kernels.GEMM_NT, dz, wt, src, part)
      outs.append((dx, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = dz.double() @ wk.double().t()
    assert float((outs[0][0].double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))

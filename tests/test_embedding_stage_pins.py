"""The embedding stage and the sparse optimizer held to outputs of the REFERENCE'S OWN code.

tests/golden/embedding_stage_vectors.npz is written by tests/golden/make_embedding_stage_vectors.py, which executes
(unmodified, on a numpy stand-in for the TF ops they call)
  compat/embedding_ops.py:15-162                  safe_embedding_lookup_sparse and its two pruning helpers
  compat/feature_column/feature_column.py:189-244 embedding_lookup_ragged
  compat/feature_column/feature_column.py:248-357 embedding_parallel_lookup on 1 / 2 / 4 simulated Horovod ranks
  compat/adam_s.py:185-213, 235-246               AdamOptimizerS._apply_sparse_shared / _finish (float32)
  compat/regularizers.py:76-108, 138-208          l2_regularizer / sum_regularizer / apply_regularization
  compat/optimizers.py:453-481                    _get_grad_norm (+ the clip multiplier of :365-376)
Held to them here: the oracle's restatements (oracle/kernel_ref.py), the tensor-level entry points of the product on
the CPU stand-in backend, and - `-m gpu` - the same entry points on the HIP kernels through the C ABI (lookup rows
1e-6, routing sets and local rows bit-exact, the optimizer's float32 arithmetic bit for bit).
"""
import os

import numpy as np
import pytest
import torch

from easyrec_amd import kernels
from oracle import kernel_ref

V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'embedding_stage_vectors.npz'))
COMB = {'sum': 0, 'mean': 1, 'sqrtn': 2}


def _csr(dense_ids, dense_w=None):
  """dense [rows, width] ids with -2 = no entry -> (ids, offsets, weights) in row-major entry order"""
  ids, offs, ws = [], [0], []
  for r in range(dense_ids.shape[0]):
    for c in range(dense_ids.shape[1]):
      if dense_ids[r, c] != -2:
        ids.append(int(dense_ids[r, c]))
        if dense_w is not None:
          ws.append(float(dense_w[r, c]))
    offs.append(len(ids))
  return (np.asarray(ids, dtype=np.int64), np.asarray(offs, dtype=np.int32),
          None if dense_w is None else np.asarray(ws, dtype=np.float32))


def _backend(device):
  if device == 'cpu':
    return kernel_ref.RefBackend()
  assert torch.cuda.is_available(), 'gpu tests need an MI355X'
  return kernels.hip()


def _lookup(be, device, table, ids, offsets, weights, combiner, n_rows):
  """one lookup through the backend's forward entry point (er_emb_plan_create + er_emb_fwd)"""
  t = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32)).to(device)
  pad = lambda a, dt: torch.from_numpy(np.concatenate([a, np.zeros(1, dtype=a.dtype)]).astype(dt)).to(device)  # noqa: E731
  out = torch.zeros(n_rows, t.shape[1], dtype=torch.float32, device=device)
  spec = kernels.LookupSpec(table=t, ids=pad(ids, np.int64), offsets=torch.from_numpy(offsets).to(device),
                            weights=None if weights is None else pad(weights, np.float32), out=out, out_col=0,
                            rows=t.shape[0], key_base=0, dim=t.shape[1], combiner=combiner, n_rows=n_rows,
                            max_nnz=max(len(ids), 1))
  plan = be.emb_plan_create([spec])
  be.emb_fwd(plan)
  if device != 'cpu':
    torch.cuda.synchronize()
  be.emb_plan_destroy(plan)
  return out.cpu().numpy()


def _check_safe_lookup(device):
  be = _backend(device)
  checked = 0
  for i in range(int(V['safe/count'])):
    key = 'safe/%d' % i
    combiner, weighted, default_id = [str(x) for x in V[key + '/meta']]
    if default_id != 'None':
      continue  # the reference's call sites pass no default_id (feature_column_v2.py:3434-3462): empty rows are zeros
    ids, offs, w = _csr(V[key + '/ids'], V[key + '/weights'] if weighted == 'True' else None)
    want = V[key + '/out']
    table = V[key + '/table']
    got_oracle = kernel_ref.lookup_rows(table.astype(np.float32), ids, offs, w, COMB[combiner], len(offs) - 1)
    assert np.allclose(got_oracle, want, rtol=2e-6, atol=1e-6), (key, 'oracle')
    got = _lookup(be, device, table, ids, offs, w, COMB[combiner], len(offs) - 1)
    assert np.allclose(got, want, rtol=2e-6, atol=1e-6), (key, combiner, weighted, float(np.abs(got - want).max()))
    # rows whose ids were all pruned (id < 0, or weight <= 0 under mean / sqrtn) are exactly zero, not NaN
    assert np.isfinite(got).all()
    checked += 1
  assert checked == 6
  # rank-3 ids [batch, positions, values]: a sequence lookup - one combined row per (batch, position)
  idx, vals = V['safe3/indices'], V['safe3/values']
  dense = np.full((9, 3), -2, dtype=np.int64)
  for (b, p, k), v in zip(idx, vals):
    dense[b * 3 + p, k] = v
  ids, offs, _ = _csr(dense)
  got = _lookup(be, device, V['safe3/table'], ids, offs, None, COMB['mean'], 9)
  assert np.allclose(got.reshape(3, 3, -1), V['safe3/out'], rtol=2e-6, atol=1e-6)


def _check_ragged(device):
  be = _backend(device)
  for i in range(int(V['ragged/count'])):
    key = 'ragged/%d' % i
    combiner = str(V[key + '/meta'][0])
    lens = V[key + '/lens']
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = V[key + '/values'].astype(np.int64)
    got = _lookup(be, device, V[key + '/table'], ids, offs, None, COMB[combiner], len(lens))
    assert np.allclose(got, V[key + '/out'], rtol=2e-6, atol=1e-6), (key, combiner)
    assert np.allclose(kernel_ref.lookup_rows(V[key + '/table'].astype(np.float32), ids, offs, None, COMB[combiner],
                                               len(lens)), V[key + '/out'], rtol=2e-6, atol=1e-6), (key, 'oracle')


def _check_parallel(device):
  """The product's requester side (er_emb_group_set_routing + er_emb_route: de-duplicate, owner = id % W, local row =
  id / W), owner side (er_gather_rows on the owner's shard) and the lookup over the returned rows (er_emb_fwd), rank by
  rank, against embedding_parallel_lookup's outputs and against what its hvd.alltoall calls carried."""
  be = _backend(device)
  for i in range(int(V['parallel/count'])):
    key = 'parallel/%d' % i
    W, N = int(V[key + '/world']), int(V[key + '/features'])
    table = V[key + '/table']
    R, D = table.shape
    stride = (R + W - 1) // W
    shards = [torch.from_numpy(V['%s/rank%d/shard' % (key, r)].astype(np.float32)).to(device) for r in range(W)]
    for r in range(W):
      want = V['%s/rank%d/out' % (key, r)]
      feats = [_csr(V['%s/rank%d/ids%d' % (key, r, f)]) for f in range(N)]
      B = len(feats[0][1]) - 1
      dummy = torch.zeros(R, D, dtype=torch.float32, device=device)
      douts = [torch.zeros(B, D, dtype=torch.float32, device=device) for _ in range(N)]
      specs = []
      for f, (ids, offs, _) in enumerate(feats):
        ids_t = torch.from_numpy(np.concatenate([ids, np.zeros(1, np.int64)])).to(device)
        specs.append(kernels.LookupSpec(table=dummy, ids=ids_t, offsets=torch.from_numpy(offs).to(device), weights=None,
                                        out=douts[f], out_col=0, rows=R, key_base=0, dim=D, combiner=0, n_rows=B,
                                        max_nnz=max(len(ids), 1)))
      g = be.emb_group_create(specs, D, R, dummy, None, None, None)
      be.emb_group_set_routing(g, W, stride, [0] * N)
      n_ent = g['num_entries']
      ukeys = torch.zeros(n_ent, dtype=torch.int32, device=device)
      nu = torch.zeros(1, dtype=torch.int32, device=device)
      uidx = torch.zeros(n_ent, dtype=torch.int64, device=device)
      counts = torch.zeros(W, dtype=torch.int32, device=device)
      be.emb_route(g, ukeys, nu, uidx, counts)
      if device != 'cpu':
        torch.cuda.synchronize()
      n = int(nu.item())
      keys = ukeys[:n].cpu().numpy().astype(np.int64)
      cnt = counts.cpu().numpy()
      assert int(cnt.sum()) == n and np.all(np.diff(keys) > 0)
      owner, local = keys // stride, keys % stride
      if W > 1:  # bit-exact routing: the id sets per owner, and the local rows the owner gathers
        sent, splits = V['%s/rank%d/sent_ids' % (key, r)], V['%s/rank%d/sent_splits' % (key, r)]
        assert np.array_equal(cnt, splits), (key, r)
        b = np.concatenate([[0], np.cumsum(splits)])
        for w in range(W):
          ref_ids = np.sort(sent[b[w]:b[w + 1]])
          assert np.all(ref_ids % W == w)
          assert np.array_equal(np.sort(local[owner == w] * W + w), ref_ids), (key, r, w)
          assert np.array_equal(np.sort(local[owner == w]), ref_ids // W)
      # owners gather; the rows come back in key order; the lookup runs over them with the entries' unique positions
      recv = torch.zeros(max(n, 1), D, dtype=torch.float32, device=device)
      pos = 0
      for w in range(W):
        c = int(cnt[w])
        if c:
          be.gather_rows(shards[w], ukeys[pos:pos + c].contiguous(), c, w * stride, recv[pos:pos + c])
        pos += c
      u = uidx.cpu().numpy()
      base = 0
      for f, (ids, offs, _) in enumerate(feats):
        ent = u[base:base + len(ids)]
        assert np.all(ent >= 0)
        got = _lookup(be, device, recv.cpu().numpy(), ent.astype(np.int64), offs, None, 0, B)
        assert np.allclose(got, want[:, f * D:(f + 1) * D], rtol=2e-6, atol=1e-6), (key, r, f)
        base += max(len(ids), 1)
      be.emb_group_destroy(g)


def _check_adam_s(device):
  """`lazy_adam_optimizer` (builders/optimizer_builder.py:91-101 -> AdamOptimizerS): three applies, the host's lr_t and
  beta powers (OptimizerState) and the row kernel (er_emb_apply_unique) against the reference's float32 arithmetic."""
  from easyrec_amd.builders.optimizer_builder import OptimizerState
  be = _backend(device)
  var0 = V['adam_s/var0']
  rows, dim = var0.shape
  var = torch.from_numpy(var0.copy()).to(device)
  m, v = torch.zeros_like(var), torch.zeros_like(var)
  ids = torch.zeros(8, dtype=torch.int64, device=device)
  dout = torch.zeros(8, dim, device=device)
  spec = kernels.LookupSpec(table=var, ids=ids, offsets=None, weights=None, out=dout, out_col=0, rows=rows, key_base=0,
                            dim=dim, combiner=0, n_rows=8, max_nnz=8)
  g = be.emb_group_create([spec], dim, rows, var, m, v, None)
  lrs = [float(V['adam_s/step%d/lr' % s]) for s in range(int(V['adam_s/steps']))]
  opt = OptimizerState(kernels.OPT_LAZY_ADAM, lambda step: lrs[step])
  for s in range(len(lrs)):
    hyper = torch.from_numpy(opt.hyper_row(s)).to(device)
    idx, grad = V['adam_s/step%d/indices' % s], V['adam_s/step%d/grad' % s]
    keys = torch.from_numpy(idx.astype(np.int32)).to(device)
    grads = torch.from_numpy(grad).to(device)
    nu = torch.tensor([len(idx)], dtype=torch.int32, device=device)
    be.emb_apply_unique(g, keys, grads, nu, kernels.OPT_LAZY_ADAM, hyper)
    opt.finish_step()
    if device != 'cpu':
      torch.cuda.synchronize()
    assert np.array_equal(np.asarray([opt.beta1_power, opt.beta2_power], dtype=np.float32), V['adam_s/step%d/beta_powers' % s])
    for name, t in (('var', var), ('m', m), ('v', v)):
      want = V['adam_s/step%d/%s' % (s, name)]
      got = t.cpu().numpy()
      assert np.array_equal(got, want), (s, name, float(np.abs(got - want).max()))
    # and the oracle's row arithmetic
    h = opt_row = None  # noqa: F841
  be.emb_group_destroy(g)
  # oracle: apply_sparse over the same three steps
  var_o, m_o, v_o = var0.copy(), np.zeros_like(var0), np.zeros_like(var0)
  opt = OptimizerState(kernels.OPT_LAZY_ADAM, lambda step: lrs[step])
  for s in range(len(lrs)):
    h = opt.hyper_row(s)
    grads = {int(k): V['adam_s/step%d/grad' % s][j] for j, k in enumerate(V['adam_s/step%d/indices' % s])}
    kernel_ref.apply_sparse(var_o, m_o, v_o, grads, kernel_ref.OPT_LAZY_ADAM, h)
    opt.finish_step()
    assert np.array_equal(var_o, V['adam_s/step%d/var' % s]) and np.array_equal(m_o, V['adam_s/step%d/m' % s]) and \
        np.array_equal(v_o, V['adam_s/step%d/v' % s]), s


def _check_regularizers(device):
  """l2_regularizer(scale)(w) = scale * sum(w^2) / 2, summed over the weights by apply_regularization into
  REGULARIZATION_LOSSES: the product's er_l2_loss over (weights, per-element coefficient)."""
  be = _backend(device)
  scale = float(V['reg/l2_scale'])
  ws = [V['reg/w%d' % i] for i in range(3)]
  each = [scale * 0.5 * float((w.astype(np.float64) ** 2).sum()) for w in ws]
  assert np.allclose(each, V['reg/l2_each'], rtol=1e-6)
  flat = torch.from_numpy(np.concatenate([w.reshape(-1) for w in ws])).to(device)
  coef = torch.full_like(flat, scale)
  out = torch.zeros(1, device=device)
  be.l2_loss(flat, coef, out, accumulate=False)
  assert abs(float(out.item()) - float(V['reg/apply'])) <= 2e-6 * abs(float(V['reg/apply']))
  # sum_regularizer([l2(1e-3), l2(5e-4), disabled]) on the first weight: the scales add
  assert abs(float(V['reg/sum']) - 1.5 * float(V['reg/l2_each'][0])) <= 1e-6 * abs(float(V['reg/sum']))


def _check_grad_norm(device):
  """The global norm = sqrt(2 * sum of l2_loss over every dense gradient and over the UN-MERGED values of every
  IndexedSlices) and the multiplier clip * min(1 / norm, 1 / clip): er_gradsq_dense + er_gradsq_rows + er_clip_scale."""
  be = _backend(device)
  dense = [V['norm/dense%d' % i] for i in range(2)]
  flat = torch.from_numpy(np.concatenate([d.reshape(-1) for d in dense])).to(device)
  slices = torch.from_numpy(V['norm/slices_values']).to(device)
  hyper = torch.zeros(2, kernels.HYPER_FLOATS, device=device)
  hyper[:, kernels.HYPER_GSCALE] = 1.0
  normsq = torch.zeros(1, device=device)
  be.gradsq_dense(torch.zeros_like(flat), flat, None, hyper[1], normsq, accumulate=False)
  be.gradsq_rows(slices, slices.shape[1], 1.0, normsq, True)
  want_sparse, want_dense, want_all = [float(x) for x in V['norm/single']]
  for clip in (0.5, 100.0):
    norm_out = torch.zeros(1, device=device)
    be.clip_scale(normsq, clip, hyper, norm_out)
    if device != 'cpu':
      torch.cuda.synchronize()
    assert abs(float(norm_out.item()) - want_all) <= 2e-6 * want_all
    scale = float(hyper[0, kernel_ref.HYPER_CLIP].item())
    ref_scale = float(V['norm/clip%g/dense0' % clip].reshape(-1)[0] / dense[0].reshape(-1)[0])
    eff = scale if scale != 0 else 1.0  # (0 in the record = "no clipping": multiply by one)
    assert abs(eff - ref_scale) <= 3e-6 * ref_scale, (clip, scale, ref_scale)
  assert abs(np.sqrt(want_sparse ** 2 + want_dense ** 2) - want_all) <= 1e-6 * want_all
  # embedding-parallel: each rank's share of the sharded tables' squares is summed over the ranks
  per_rank = [V['norm/ep/rank%d/values' % r] for r in range(2)]
  total = np.sqrt(sum(float((d.astype(np.float64) ** 2).sum()) for d in dense) +
                  sum(float((p.astype(np.float64) ** 2).sum()) for p in per_rank))
  for r in range(2):
    assert abs(float(V['norm/ep/rank%d/norms' % r][2]) - total) <= 2e-6 * total


CHECKS = [_check_safe_lookup, _check_ragged, _check_parallel, _check_adam_s, _check_regularizers, _check_grad_norm]


@pytest.mark.parametrize('check', CHECKS, ids=lambda f: f.__name__[7:])
def test_stand_in_backend_and_oracle(check, built_lib):
  check('cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('check', CHECKS, ids=lambda f: f.__name__[7:])
def test_hip_kernels(check):
  check('cuda:0')


def test_fixture_is_what_the_generator_writes(tmp_path):
  """Where /root/reference exists: re-run the generator and compare array by array."""
  import subprocess
  import sys
  from conftest import REFERENCE, reference_available
  if not reference_available():
    pytest.skip('no /root/reference on this box')
  gen = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_embedding_stage_vectors.py')
  src = open(gen).read().replace("path = os.path.join(HERE, 'embedding_stage_vectors.npz')",
                                 "path = %r" % str(tmp_path / 'v.npz'))
  script = tmp_path / 'gen.py'
  script.write_text(src)
  subprocess.run([sys.executable, str(script), REFERENCE], check=True, capture_output=True, timeout=300)
  fresh = np.load(str(tmp_path / 'v.npz'))
  assert sorted(fresh.files) == sorted(V.files)
  for k in V.files:
    assert np.array_equal(fresh[k], V[k]), k

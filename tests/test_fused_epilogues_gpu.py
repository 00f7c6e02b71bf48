"""-m gpu: the epilogues the contractions carry since round 6 (include/easyrec_hip.h er_gemm_epilogue) through the C ABI:
the DCN-v2 cross layer inside its GEMM (reference layers/keras/interaction.py:249-286: x_{l+1} = x0 * (W x_l + b + diag * x_l)
+ x_l), forward and backward, fp32 (er_gemm_f32_cross) and bf16 (er_gemm_bf16_nt_epi); the BatchNorm statistics /
BatchNorm-backward sums of layers/dnn.py:57-79 in the bf16 contraction; the bf16 copies the producers write for their
consumers.  Checkers: the unfused library launches (bit-exact where the arithmetic order is the same) and fp64 torch
formulas of the reference's expressions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from easyrec_amd import kernels  # noqa: E402

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
  assert torch.cuda.is_available(), 'gpu tests need an MI355X'
  be = kernels.hip()
  be._ck(be.lib.er_gemm_bf16_nt_prepare(), 'prepare')
  return be


def _r16(x):
  return x.to(torch.bfloat16).to(torch.float32)


def _b16(hip, x):
  pad = kernels.Bf16Shadows.pad8
  xb = torch.empty(x.shape[0], pad(x.shape[1]), dtype=torch.bfloat16, device=DEV)
  hip.cast_bf16([(x, xb, False)])
  return xb


def _epi(**kw):
  return kernels.GemmEpilogue(**{k: (v.data_ptr() if torch.is_tensor(v) else v) for k, v in kw.items()})


@pytest.mark.parametrize('B,d', [(4096, 624), (300, 72), (64, 64), (130, 200)])
@pytest.mark.parametrize('diag', [0.0, 0.25])
@pytest.mark.parametrize('bias', [True, False])
def test_cross_forward_in_the_fp32_contraction_equals_gemm_plus_epilogue(hip, B, d, diag, bias):
  """er_gemm_f32_cross(ER_EPI_CROSS_FWD) against er_gemm_f32 + er_cross_v2_epilogue_fwd: same accumulators, same epilogue
  order -> the same bits, for out and for the kept product u."""
  g = torch.Generator().manual_seed(B + d)
  x0 = torch.randn(B, d, generator=g).to(DEV)
  x = torch.randn(B, d, generator=g).to(DEV)
  w = (torch.randn(d, d, generator=g) * 0.05).to(DEV)
  b = torch.randn(d, generator=g).to(DEV) if bias else None
  u_ref = hip.gemm(kernels.GEMM_NN, x, w)
  out_ref = hip.cross_v2_fwd(x0, x, u_ref, b, diag)
  out, u = hip.cross_fwd_fused(x0, x, w, b, diag, False)
  torch.cuda.synchronize()
  assert torch.equal(u, u_ref)
  assert torch.equal(out, out_ref)


def _cross_formula(x0, x, w, b, diag):
  u = x.double() @ w.double()
  t = u + (0 if b is None else b.double()) + diag * x.double()
  return x0.double() * t + x.double(), u


@pytest.mark.parametrize('B,d', [(4096, 624), (300, 72), (257, 128)])
@pytest.mark.parametrize('diag', [0.0, 0.5])
def test_cross_forward_in_the_bf16_contraction(hip, B, d, diag):
  """er_gemm_bf16_nt_epi(ER_EPI_CROSS_FWD): operands rounded to bf16, everything else fp32.  Against the fp64 formula over
  the same rounded operands: fp32 summation-order error only; the bf16 copy is the rounding of the fp32 output."""
  g = torch.Generator().manual_seed(B * 3 + d)
  x0 = torch.randn(B, d, generator=g).to(DEV)
  x = torch.randn(B, d, generator=g).to(DEV)
  w = (torch.randn(d, d, generator=g) * 0.05).to(DEV)
  b = torch.randn(d, generator=g).to(DEV)
  xb, wt = _b16(hip, x), _b16(hip, w.t().contiguous())
  out = torch.empty(B, d, device=DEV)
  u = torch.empty(B, d, device=DEV)
  outb = torch.zeros(B, kernels.Bf16Shadows.pad8(d), dtype=torch.bfloat16, device=DEV)
  epi = _epi(kind=kernels.EPI_CROSS_FWD, diag=diag, x0=x0, xl=x, u=u, ld_x0=d, ld_xl=d, ld_u=d)
  hip.gemm_bf16_nt(xb, wt, B, d, d, out=out, out_bf16=outb, bias=b, epi=epi)
  torch.cuda.synchronize()
  u_ref = _r16(x).double() @ _r16(w).double()
  scale = _r16(x).abs().double() @ _r16(w).abs().double()
  assert bool(((u.double() - u_ref).abs() <= 2e-5 * scale + 1e-6).all())
  t = u.double() + b.double() + diag * x.double()  # (the epilogue's own arithmetic from the kernel's u: fp32 rounding only)
  ref = x0.double() * t + x.double()
  assert float((out.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
  assert torch.equal(outb[:, :d], out.to(torch.bfloat16))


@pytest.mark.parametrize('M,N,K', [(4096, 256, 624), (4096, 64, 688), (300, 72, 40), (130, 128, 64), (64, 16, 32)])
@pytest.mark.parametrize('bias', [True, False])
def test_bf16_contraction_emits_batchnorm_statistics(hip, M, N, K, bias):
  """ER_EPI_STATS: per 64-row tile (count, mean, M2) of the output columns; er_bn_apply_from_stats on them must give what
  er_bn_act_fwd computes from the output itself."""
  g = torch.Generator().manual_seed(M + N + K)
  a = torch.randn(M, K, generator=g).to(DEV)
  w = torch.randn(K, N, generator=g).to(DEV)
  b = torch.randn(N, generator=g).to(DEV) if bias else None
  ab, wt = _b16(hip, a), _b16(hip, w.t().contiguous())
  T = hip.gemm_row_tiles(M)
  stats = torch.full((T * N * 3,), float('nan'), device=DEV)
  z = torch.empty(M, N, device=DEV)
  hip.gemm_bf16_nt(ab, wt, M, N, K, out=z, bias=b, epi=_epi(kind=kernels.EPI_STATS, col_stats=stats))
  z_plain = torch.empty(M, N, device=DEV)
  hip.gemm_bf16_nt(ab, wt, M, N, K, out=z_plain, bias=b)
  torch.cuda.synchronize()
  assert torch.equal(z, z_plain)
  st = stats.view(T, N, 3).cpu().double()
  zc = z.cpu().double()
  for t in range(T):
    rows = zc[t * 64:(t + 1) * 64]
    assert bool((st[t, :, 0] == rows.shape[0]).all())
    assert float((st[t, :, 1] - rows.mean(0)).abs().max()) <= 1e-5 * float(rows.abs().max())
    m2 = ((rows - rows.mean(0)) ** 2).sum(0)
    assert float((st[t, :, 2] - m2).abs().max()) <= 1e-4 * float(m2.abs().max() + 1e-6)
  gamma = (torch.rand(N, generator=g) + 0.5).to(DEV)
  beta = (torch.randn(N, generator=g) * 0.1).to(DEV)
  y, mean, invstd = hip.bn_apply_from_stats(z, None, stats, T, gamma, beta, 1e-3, 0.99, None, None, kernels.ACT_RELU)
  y2, mean2, invstd2 = hip.bn_act_fwd(z, None, gamma, beta, 1, 1e-3, 0.99, None, None, kernels.ACT_RELU)
  torch.cuda.synchronize()
  assert float((mean - mean2).abs().max()) <= 1e-5 * float(mean2.abs().max() + 1e-3)
  assert float((y - y2).abs().max()) <= 2e-5 * float(y2.abs().max())


@pytest.mark.parametrize('B,N,K,col0,n_src', [(4096, 256, 128, 0, 0), (4096, 688, 64, 0, 64), (300, 72, 40, 8, 32),
                                              (130, 128, 64, 0, 0)])
@pytest.mark.parametrize('act', [kernels.ACT_RELU, kernels.ACT_NONE])
def test_bf16_dgrad_emits_batchnorm_backward_sums(hip, B, N, K, col0, n_src, act):
  """ER_EPI_BN_BWD: dy = dz_next . W^T from bf16 operands also leaves (sum g, sum g xhat) per 64-row tile for the layer
  below (all columns, or the block [col0, col0 + n_src)); er_bn_act_bwd_from_partials must give what the two-pass
  er_bn_act_bwd gives from the same dy."""
  ns = n_src or N
  g = torch.Generator().manual_seed(B + N + K + act)
  z = torch.randn(B, ns, generator=g).to(DEV)
  gamma = (torch.rand(ns, generator=g) + 0.5).to(DEV)
  beta = (torch.randn(ns, generator=g) * 0.1).to(DEV)
  y, mean, invstd = hip.bn_act_fwd(z, None, gamma, beta, 1, 1e-3, 0.99, None, None, act)
  dz_next = (torch.randn(B, K, generator=g) * 0.1).to(DEV)
  w = torch.randn(N, K, generator=g).to(DEV)
  ab, wb = _b16(hip, dz_next), _b16(hip, w)
  T = hip.gemm_row_tiles(B)
  partial = torch.full((T * ns * 2,), float('nan'), device=DEV)
  dy = torch.empty(B, N, device=DEV)
  epi = _epi(kind=kernels.EPI_BN_BWD, bn_z=z, bn_y=y, bn_mean=mean, bn_invstd=invstd, bn_partial=partial, bn_ld=ns, bn_use_bn=1,
             bn_act=int(act), bn_col0=col0, bn_n_src=n_src)
  hip.gemm_bf16_nt(ab, wb, B, N, K, out=dy, epi=epi)
  dy_plain = torch.empty(B, N, device=DEV)
  hip.gemm_bf16_nt(ab, wb, B, N, K, out=dy_plain)
  torch.cuda.synchronize()
  assert torch.equal(dy, dy_plain) and not torch.isnan(partial).any()
  blk = dy[:, col0:col0 + ns]
  ref = hip.bn_act_bwd(z, None, gamma, y, mean, invstd, blk, 1, act, False, True)
  got = hip.bn_act_bwd(z, None, gamma, y, mean, invstd, blk, 1, act, False, True, partial=partial)
  torch.cuda.synchronize()
  for a, b_, what in zip(got, ref, ('dz', 'dbias', 'dgamma', 'dbeta')):
    if a is None:
      assert b_ is None
      continue
    scale = float(b_.abs().max()) + 1e-12
    assert float((a - b_).abs().max()) <= 2e-5 * scale, (what, float((a - b_).abs().max()), scale)


def _cross_bwd_reference(x0, xs, us, ws, bs, diag, dout_top):
  """fp64 gradients of L stacked cross layers by torch autograd of the reference's expression."""
  x0d = x0.double().requires_grad_(True)
  wd = [w.double().requires_grad_(True) for w in ws]
  bd = [b.double().requires_grad_(True) for b in bs]
  x = x0d
  for w, b in zip(wd, bd):
    x = x0d * (x @ w + b + diag * x) + x
  x.backward(dout_top.double())
  return x0d.grad, [w.grad for w in wd], [b.grad for b in bd]


@pytest.mark.parametrize('B,d,L', [(4096, 624, 3), (300, 72, 2), (130, 64, 1)])
@pytest.mark.parametrize('diag', [0.0, 0.3])
@pytest.mark.parametrize('bf16', [False, True])
def test_cross_stack_backward_through_the_fused_chain(hip, B, d, L, diag, bf16):
  """The backward of L stacked cross layers as the model issues it - er_cross_v2_bwd_top for the top layer, then one
  input-gradient contraction per layer whose epilogue (ER_EPI_CROSS_BWD) adds dout and runs the layer below, the bias
  gradients finished by er_colsum_partials_multi - against fp64 autograd of the reference's formula.  fp32: 2e-5 of each
  gradient's scale; bf16 operands: 2e-2 (three decimal digits in the operands of every contraction)."""
  g = torch.Generator().manual_seed(B + d + L)
  x0 = torch.randn(B, d, generator=g).to(DEV)
  ws = [(torch.randn(d, d, generator=g) * (0.5 / d ** 0.5)).to(DEV) for _ in range(L)]
  bs = [(torch.randn(d, generator=g) * 0.1).to(DEV) for _ in range(L)]
  dout_full = (torch.randn(B, d + 8, generator=g) * 0.1).to(DEV)
  dout = dout_full[:, 8:]  # (a column block of a wider gradient: tf.concat's backward)
  xs, us = [x0], []
  for w, b in zip(ws, bs):
    u = hip.gemm(kernels.GEMM_NN, xs[-1], w)
    us.append(u)
    xs.append(hip.cross_v2_fwd(x0, xs[-1], u, b, diag))
  st = None
  if bf16:
    class _VS(object):
      flat = torch.cat([w.reshape(-1) for w in ws])
      version = 0
    vs = _VS()
    # (weights as views of one flat buffer, as the product's VarStore holds them)
    off = 0
    for i, w in enumerate(ws):
      ws[i] = vs.flat[off:off + d * d].view(d, d)
      off += d * d
    st = hip.bf16_enable(vs)
  dx0 = torch.zeros(B, d, device=DEV)
  db = [torch.zeros(d, device=DEV) for _ in range(L)]
  dws = []
  jobs = []
  du, partial = hip.cross_bwd_top(x0, xs[L - 1], us[L - 1], bs[L - 1], diag, dout, dx0, True, bf16, ws[L - 1])
  jobs.append((partial, db[L - 1], d))
  g_in = dout
  for l in range(L - 1, -1, -1):
    dws.append(hip.gemm(kernels.GEMM_TN, xs[l], du))
    if l == 0:
      res = hip.cross_dgrad_fused(du, ws[l], g_in, diag, bf16, dx0, True)
      assert res == (None, None)
    else:
      t = torch.empty(B, d, device=DEV)
      prev = dict(x0=x0, u=us[l - 1], bias=bs[l - 1], xl=xs[l - 1], dx0=dx0, acc0=True)
      du, partial = hip.cross_dgrad_fused(du, ws[l], g_in, diag, bf16, t, False, prev=prev)
      jobs.append((partial, db[l - 1], d))
      g_in = t
  hip.colsum_partials_multi(jobs)
  torch.cuda.synchronize()
  dws = dws[::-1]
  rx0, rw, rb = _cross_bwd_reference(x0.cpu(), xs, us, [w.cpu() for w in ws], [b.cpu() for b in bs], diag, dout.cpu())
  tol = 2e-2 if bf16 else 2e-5
  def close(a, b_, what):
    scale = float(b_.abs().max()) + 1e-12
    err = float((a.cpu().double() - b_).abs().max())
    assert err <= tol * scale, (what, err, scale)
  close(dx0, rx0, 'dx0')
  for l in range(L):
    close(db[l], rb[l], 'db%d' % l)
    close(dws[l], rw[l], 'dw%d' % l)
  if st is not None:
    hip._bf16_states = [s for s in hip._bf16_states if s is not st]


def test_producers_write_the_bf16_copies_their_consumers_read(hip):
  """BatchNorm apply / BatchNorm backward / concat with a bf16 side output: the copy is the RNE rounding of the fp32 result
  (what er_cast_bf16 would have produced in a launch of its own), its padding columns zero."""
  g = torch.Generator().manual_seed(5)
  B, N = 4096, 64

  class _St(object):
    pad8 = staticmethod(kernels.Bf16Shadows.pad8)
    copies = {}

    def new_copy(self, x):
      xb = torch.full((x.shape[0], self.pad8(x.shape[1])), 7.0, dtype=torch.bfloat16, device=DEV)
      self.copies[x.data_ptr()] = xb
      return xb

    def register(self, x, xb):
      self.copies[x.data_ptr()] = xb

  st = _St()
  z = torch.randn(B, N, generator=g).to(DEV)
  gamma, beta = (torch.rand(N, generator=g) + 0.5).to(DEV), torch.randn(N, generator=g).to(DEV)
  T = hip.gemm_row_tiles(B)
  stats = torch.empty(T * N * 3, device=DEV)
  w = torch.eye(N, device=DEV)
  z2 = hip.gemm(kernels.GEMM_NN, z, w, col_stats=stats)
  y, mean, invstd = hip.bn_apply_from_stats(z2, None, stats, T, gamma, beta, 1e-3, 0.99, None, None, kernels.ACT_RELU,
                                            bf16_state=st)
  torch.cuda.synchronize()
  assert torch.equal(st.copies[y.data_ptr()], y.to(torch.bfloat16))
  dy = torch.randn(B, N, generator=g).to(DEV)
  dz, _, _, _ = hip.bn_act_bwd(z2, None, gamma, y, mean, invstd, dy, 1, kernels.ACT_RELU, False, True, bf16_state=st)
  torch.cuda.synchronize()
  assert torch.equal(st.copies[dz.data_ptr()], dz.to(torch.bfloat16))
  a, b = torch.randn(B, 64, generator=g).to(DEV), torch.randn(B, 13, generator=g).to(DEV)
  out = hip.concat_cols([a, b], bf16_state=st)
  torch.cuda.synchronize()
  ob = st.copies[out.data_ptr()]
  assert ob.shape[1] == 80 and torch.equal(ob[:, :77], torch.cat([a, b], 1).to(torch.bfloat16)) and bool((ob[:, 77:] == 0).all())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_dcn_v2_step_fused_cross_equals_the_separate_launches(hip, dtype, monkeypatch):
  """DCN-v2 (configs/dcn_v2_criteo_small.config) with the cross layers fused into their contractions and with round 5's
  GEMM + epilogue launches: the first step's loss to 1e-6 (fp32; the forward is bit-identical) / 2e-3 (bf16: the same
  arithmetic, but BatchNorm statistics from the bf16 kernel's own epilogue), the first Adam moments - (1 - beta1) times the
  gradients - to 2e-5 / 5e-2 of the largest moment (the backward sums in another order), the second step's loss to 1e-4 /
  1e-2."""
  import os
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cfg = os.path.join(root, 'configs', 'dcn_v2_criteo_small.config')

  def run(fused):
    monkeypatch.setattr(kernels.HipBackend, 'fused_cross', fused)
    monkeypatch.setattr(kernels.HipBackend, 'bf16_epilogues', fused)
    est = EasyRecEstimator(cfg, device=DEV, batch_size=512, seed=3, dense_dtype=dtype).build()
    gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=512, seed=11)
    first = float(est.train_step(gen.next_batch())['total_loss'])
    torch.cuda.synchronize()
    m = est.varstore.slots['m'].clone()
    second = float(est.train_step(gen.next_batch())['total_loss'])
    return first, second, m

  f1, s1, m1 = run(True)
  f0, s0, m0 = run(False)
  ltol, mtol, stol = (1e-6, 2e-5, 1e-4) if dtype == 'f32' else (2e-3, 5e-2, 1e-2)
  assert abs(f1 - f0) <= ltol * abs(f0), (f1, f0)
  assert float((m1 - m0).abs().max()) <= mtol * float(m0.abs().max()), float((m1 - m0).abs().max())
  assert abs(s1 - s0) <= stol * abs(s0), (s1, s0)


@pytest.mark.parametrize('B,L,E,N', [(4096, 50, 32, 128), (64, 12, 32, 128), (37, 7, 16, 72), (5, 3, 48, 64), (130, 50, 32, 36)])
def test_din_first_layer_contractions_generate_the_attention_input(hip, B, L, E, N):
  """er_din_gemm_fwd / _wgrad / _dgrad against the ordinary contractions over the BUILT [q, h, q - h, q * h] block
  (er_din_concat_fwd / _bwd around er_gemm_f32; reference model/multi_tower_din.py:62-80).  Forward: the same kernel body
  over the same operand values in the same k order -> bit-identical z and column statistics.  Weight gradient: another
  k-split, 2e-5 of the gradient's scale.  Input gradient: dh elementwise from the same accumulators (2e-6), dq summed per
  tile then per example instead of by one wave per example (2e-5)."""
  g = torch.Generator().manual_seed(B + L + E + N)
  q = torch.randn(B, E, generator=g).to(DEV)
  h = torch.randn(B, L, E, generator=g).to(DEV)
  w = (torch.randn(4 * E, N, generator=g) * 0.1).to(DEV)
  b = torch.randn(N, generator=g).to(DEV)
  cat = hip.din_concat_fwd(q, h).reshape(B * L, 4 * E)
  T = hip.gemm_row_tiles(B * L)
  st_ref = torch.empty(T * N * 3, device=DEV)
  z_ref = hip.gemm(kernels.GEMM_NN, cat, w, bias=b, col_stats=st_ref)
  st = torch.full((T * N * 3,), float('nan'), device=DEV)
  z = hip.din_gemm_fwd(q, h, w, b, col_stats=st)
  torch.cuda.synchronize()
  assert torch.equal(z, z_ref) and torch.equal(st, st_ref)
  dz = (torch.randn(B * L, N, generator=g) * 0.1).to(DEV)
  dw_ref = hip.gemm(kernels.GEMM_TN, cat, dz)
  base = torch.randn(4 * E, N, generator=g).to(DEV)
  dw = base.clone()
  hip.din_gemm_wgrad(q, h, dz, dw, accumulate=True)
  torch.cuda.synchronize()
  scale = float(dw_ref.abs().max())
  assert float((dw - base - dw_ref).abs().max()) <= 2e-5 * scale
  dcat = hip.gemm(kernels.GEMM_NT, dz, w).reshape(B, L, 4 * E)
  dq_ref, dh_ref = hip.din_concat_bwd(q, h, dcat)
  hbase = torch.randn(B, L, E, generator=g).to(DEV)
  dh = hbase.clone()
  dq, _ = hip.din_gemm_dgrad(dz, w, q, h, dh=dh, acc_h=True)
  dq2, dh2 = hip.din_gemm_dgrad(dz, w, q, h)
  torch.cuda.synchronize()
  assert float((dh2 - dh_ref).abs().max()) <= 2e-6 * float(dh_ref.abs().max())
  assert float((dh - hbase - dh_ref).abs().max()) <= 1e-5 * float(dh_ref.abs().max())
  assert torch.equal(dq, dq2)
  assert float((dq - dq_ref).abs().max()) <= 2e-5 * float(dq_ref.abs().max())


def test_din_step_with_the_generated_attention_input_equals_the_built_one(hip, monkeypatch):
  """MultiTowerDIN (configs/din_taobao_small.config) with the first attention layer on the generated operand and with the
  built [B, L, 4E] block: the first step's loss is identical (bit-identical forward), the first Adam moments within 2e-5 of
  the largest, the second step's loss within 1e-4."""
  import os
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cfg = os.path.join(root, 'configs', 'din_taobao_small.config')

  def run(fused):
    monkeypatch.setattr(kernels.HipBackend, 'din_fused', fused)
    est = EasyRecEstimator(cfg, device=DEV, batch_size=256, seed=3).build()
    gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=256, seed=11)
    n0 = len([1 for _ in ()])
    first = float(est.train_step(gen.next_batch())['total_loss'])
    torch.cuda.synchronize()
    m = est.varstore.slots['m'].clone()
    second = float(est.train_step(gen.next_batch())['total_loss'])
    return first, second, m

  f1, s1, m1 = run(True)
  f0, s0, m0 = run(False)
  assert f1 == f0, (f1, f0)
  assert float((m1 - m0).abs().max()) <= 2e-5 * float(m0.abs().max()), float((m1 - m0).abs().max())
  assert abs(s1 - s0) <= 1e-4 * abs(s0), (s1, s0)


def test_din_step_with_the_tall_batchnorm_applies_in_the_next_contraction_is_bit_identical(hip, monkeypatch):
  """MultiTowerDIN (configs/din_taobao_small.config at B = 2048, L = 12: 24,576 attention rows, more than the 16,384 above which
  column statistics take the merge launch) with the attention MLP's BatchNorm applies inside the next layer's contraction
  (HipBackend.bn_in_staging: er_bn_finalize_from_stats + er_gemm_f32_bn_a) and as launches of their own: the same arithmetic
  in the same order - losses of two steps and the first Adam moments bit-identical."""
  import os
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cfg = os.path.join(root, 'configs', 'din_taobao_small.config')
  calls = []
  real = kernels.HipBackend.gemm_bn_a

  def counted(self, *a, **k):
    calls.append(1)
    return real(self, *a, **k)

  monkeypatch.setattr(kernels.HipBackend, 'gemm_bn_a', counted)

  def run(on):
    monkeypatch.setattr(kernels.HipBackend, 'bn_in_staging', on)
    est = EasyRecEstimator(cfg, device=DEV, batch_size=2048, seed=3).build()
    gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=2048, seed=11)
    first = float(est.train_step(gen.next_batch())['total_loss'])
    torch.cuda.synchronize()
    m = est.varstore.slots['m'].clone()
    mv = {k: np.array(v) for k, v in est.varstore.state_dict().items() if 'moving_' in k}
    second = float(est.train_step(gen.next_batch())['total_loss'])
    return first, second, m, mv

  f1, s1, m1, mv1 = run(True)
  n_on = len(calls)
  f0, s0, m0, mv0 = run(False)
  assert n_on >= 2 and len(calls) == n_on, (n_on, len(calls))  # (the staging form ran, and only when switched on)
  assert f1 == f0 and s1 == s0, (f1, f0, s1, s0)
  assert torch.equal(m1, m0)
  assert len(mv0) > 0
  for k in mv0:
    assert np.array_equal(mv1[k], mv0[k]), k


@pytest.mark.parametrize('M,K,Ns', [(8192, 1152, (256, 256, 4)), (4096, 64, (192, 128, 64, 20)), (100, 36, (37, 64)), (8192, 256, (192,) * 4)])
def test_grouped_contraction_with_the_frozen_batchnorm_in_its_epilogue_equals_the_two_launches(hip, M, K, Ns):
  """er_gemm_grouped_f32 with er_gemm_problem.fz_* (bias + BatchNorm on the moving statistics + activation in the epilogue of
  the problems that ask for it; the experts of the reference's MMoE, layers/mmoe.py:62-83 -> layers/dnn.py:57-79) against the
  grouped contraction followed by the frozen BatchNorm launch: z, y, saved mean / invstd bit for bit; a problem of the same
  launch without the epilogue (a gate) is the plain contraction."""
  g = torch.Generator().manual_seed(M + K + sum(Ns))
  x = torch.randn(M, K, generator=g).to(DEV)
  ws = [(torch.randn(K, n, generator=g) * 0.1).to(DEV) for n in Ns]
  plain = len(Ns) - 1  # the last problem has no epilogue (and a GEMM bias)
  pb = torch.randn(Ns[-1], generator=g).to(DEV)
  P = []
  for n in Ns[:-1]:
    P.append(dict(bias=torch.randn(n, generator=g).to(DEV), gamma=(torch.rand(n, generator=g) + 0.5).to(DEV),
                  beta=torch.randn(n, generator=g).to(DEV), moving_mean=(torch.randn(n, generator=g) * 0.2).to(DEV),
                  moving_var=(torch.rand(n, generator=g) + 0.3).to(DEV), eps=1e-3, act=kernels.ACT_RELU))
  # apart
  z1 = [torch.empty(M, n, device=DEV) for n in Ns]
  hip.gemm_grouped(kernels.GEMM_NN, [(x, ws[e], z1[e], pb if e == plain else None, False) for e in range(len(Ns))])
  outs = hip.bn_fwd_multi([dict(x=z1[e], bias=P[e]['bias'], gamma=P[e]['gamma'], beta=P[e]['beta'], moving_mean=P[e]['moving_mean'],
                                moving_var=P[e]['moving_var'], col_stats=None, use_bn=kernels.BN_FROZEN, act=P[e]['act'], eps=1e-3,
                                momentum=0.99) for e in range(plain)])
  # one launch
  z2 = [torch.full((M, n), float('nan'), device=DEV) for n in Ns]
  fz = [dict(P[e], y=torch.full((M, Ns[e]), float('nan'), device=DEV), save=torch.empty(2, Ns[e], device=DEV)) for e in range(plain)]
  hip.gemm_grouped(kernels.GEMM_NN, [(x, ws[e], z2[e], pb if e == plain else None, False, None, None, fz[e] if e < plain else None)
                                     for e in range(len(Ns))])
  torch.cuda.synchronize()
  for e in range(len(Ns)):
    assert torch.equal(z2[e], z1[e]), e
  for e in range(plain):
    y, mean, invstd = outs[e]
    assert torch.equal(fz[e]['y'], y), e
    assert torch.equal(fz[e]['save'][0], mean) and torch.equal(fz[e]['save'][1], invstd), e


@pytest.mark.parametrize('name', ['mmoe_taobao_small.config', 'dbmtl_mmoe_taobao_small.config'])
def test_multi_task_step_with_the_frozen_batchnorm_in_the_contractions_epilogue_is_bit_identical(hip, monkeypatch, name):
  """MMoE / DBMTL-over-MMoE with the experts' bias + frozen BatchNorm + ReLU inside the grouped contraction's epilogue
  (HipBackend.frozen_bn_epilogue) and as the depth's BatchNorm launch: losses of two steps and the first Adam moments
  bit-identical."""
  import os
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cfg = os.path.join(root, 'configs', name)
  seen = []
  real = kernels.HipBackend._gemm_problems

  def spy(self, layout, problems, *a, **k):
    seen.append(sum(1 for pr in problems if len(pr) > 7 and pr[7] is not None))
    return real(self, layout, problems, *a, **k)

  monkeypatch.setattr(kernels.HipBackend, '_gemm_problems', spy)

  def run(on):
    monkeypatch.setattr(kernels.HipBackend, 'frozen_bn_epilogue', on)
    # (B = 4096: every forward problem of the grouped launches is one k-split either way - at B = 512 the launch WITHOUT the
    # epilogue splits the experts' contractions over k, another summation order)
    est = EasyRecEstimator(cfg, device=DEV, batch_size=4096, seed=3).build()
    gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=4096, seed=11)
    first = float(est.train_step(gen.next_batch())['total_loss'])
    torch.cuda.synchronize()
    m = est.varstore.slots['m'].clone()
    second = float(est.train_step(gen.next_batch())['total_loss'])
    return first, second, m

  f1, s1, m1 = run(True)
  n_on = sum(seen)
  f0, s0, m0 = run(False)
  assert n_on > 0 and sum(seen) == n_on, (n_on, sum(seen))  # (the epilogue form ran, and only when switched on)
  assert f1 == f0 and s1 == s0, (f1, f0, s1, s0)
  assert torch.equal(m1, m0)


@pytest.mark.parametrize('M,K,Ns', [(8192, 192, (256, 256)), (4096, 64, (128, 192, 33)), (100, 36, (37, 64))])
def test_input_gradient_contraction_runs_the_frozen_batchnorm_backward_in_its_epilogue(hip, M, K, Ns):
  """er_gemm_grouped_f32 (NT) with er_gemm_problem.bn_dz_out: dy = dz_next . W^T of a layer that normalises with the MOVING
  statistics leaves the launch as dz = gamma * invstd * (dy masked by the ReLU) - bit for bit what er_bn_bwd_multi's frozen
  form makes of the plain contraction's output - together with the column sums; the parameter gradients that
  er_bn_bwd_multi(dx = None) derives from those sums agree with the two-pass form to 2e-5 of their scale (another summation
  order: per 64-row tile instead of per row chunk)."""
  g = torch.Generator().manual_seed(M + K + sum(Ns))
  L = []
  for n in Ns:
    z = torch.randn(M, n, generator=g).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    gamma = (torch.rand(n, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(n, generator=g) * 0.2).to(DEV)
    mm, mv = (torch.randn(n, generator=g) * 0.2).to(DEV), (torch.rand(n, generator=g) + 0.3).to(DEV)
    y, mean, invstd = hip.bn_act_fwd(z, bias, gamma, beta, kernels.BN_FROZEN, 1e-3, 0.99, mm, mv, kernels.ACT_RELU)
    dzn = (torch.randn(M, K, generator=g) * 0.1).to(DEV)
    w = torch.randn(n, K, generator=g).to(DEV)
    L.append(dict(z=z, bias=bias, gamma=gamma, beta=beta, y=y, mean=mean, invstd=invstd, dzn=dzn, w=w))
  # apart: plain contraction, then the frozen backward (column sums + apply)
  dy1 = [torch.empty(M, n, device=DEV) for n in Ns]
  hip.gemm_grouped(kernels.GEMM_NT, [(l['dzn'], l['w'], dy1[i], None, False) for i, l in enumerate(L)])
  ref = hip.bn_bwd_multi([dict(x=l['z'], bias=l['bias'], gamma=l['gamma'], beta=l['beta'], y=l['y'], mean=l['mean'],
                               invstd=l['invstd'], dy=dy1[i], use_bn=kernels.BN_FROZEN, act=kernels.ACT_RELU, partial=None, into=None)
                          for i, l in enumerate(L)])
  # one launch + the parameter gradients from its sums
  dy2 = [torch.full((M, n), float('nan'), device=DEV) for n in Ns]
  parts = [torch.empty(hip.gemm_row_tiles(M) * n * 2, device=DEV) for n in Ns]
  srcs = []
  for l in L:
    src = kernels.BnSource(l['z'], l['bias'], l['y'], l['mean'], l['invstd'], kernels.ACT_RELU, l['gamma'], None, beta=l['beta'], fused=True)
    src.frozen = True
    srcs.append(src)
  hip.gemm_grouped(kernels.GEMM_NT, [(l['dzn'], l['w'], dy2[i], None, False, None, (srcs[i], parts[i], True)) for i, l in enumerate(L)])
  got = hip.bn_bwd_multi([dict(x=l['z'], bias=l['bias'], gamma=l['gamma'], beta=l['beta'], y=l['y'], mean=l['mean'],
                               invstd=l['invstd'], dy=dy2[i], use_bn=kernels.BN_FROZEN, act=kernels.ACT_RELU, partial=parts[i], into=None,
                               dx_done=True) for i, l in enumerate(L)])
  torch.cuda.synchronize()
  for i in range(len(Ns)):
    assert torch.equal(dy2[i], ref[i][0]), i                 # dz, bit for bit
    assert got[i][0].data_ptr() == dy2[i].data_ptr()
    for a, b in zip(got[i][1:], ref[i][1:]):                 # dbias, dgamma, dbeta
      assert float((a - b).abs().max()) <= 2e-5 * max(1e-6, float(b.abs().max())), i


def test_multi_task_step_with_the_frozen_batchnorm_backward_in_the_input_gradient_epilogue(hip, monkeypatch):
  """MMoE (configs/mmoe_taobao_small.config, B = 4096) with the experts' elementwise BatchNorm backward inside the epilogue of
  the input-gradient contraction of the layer above (HipBackend.frozen_dz_epilogue) and as the depth's BatchNorm-backward
  launches: the first step's loss identical (same forward), the first Adam moments within 2e-5 of the largest, the second
  step's loss within 1e-4."""
  import os
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cfg = os.path.join(root, 'configs', 'mmoe_taobao_small.config')
  seen = []
  real = kernels.HipBackend._gemm_problems

  def spy(self, layout, problems, *a, **k):
    seen.append(sum(1 for pr in problems if len(pr) > 6 and pr[6] is not None and len(pr[6]) > 2 and pr[6][2]))
    return real(self, layout, problems, *a, **k)

  monkeypatch.setattr(kernels.HipBackend, '_gemm_problems', spy)

  def run(on):
    monkeypatch.setattr(kernels.HipBackend, 'frozen_dz_epilogue', on)
    est = EasyRecEstimator(cfg, device=DEV, batch_size=4096, seed=3).build()
    gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=4096, seed=11)
    first = float(est.train_step(gen.next_batch())['total_loss'])
    torch.cuda.synchronize()
    m = est.varstore.slots['m'].clone()
    second = float(est.train_step(gen.next_batch())['total_loss'])
    return first, second, m

  f1, s1, m1 = run(True)
  n_on = sum(seen)
  f0, s0, m0 = run(False)
  assert n_on > 0 and sum(seen) == n_on, (n_on, sum(seen))
  assert f1 == f0, (f1, f0)
  assert float((m1 - m0).abs().max()) <= 2e-5 * float(m0.abs().max()), float((m1 - m0).abs().max())
  assert abs(s1 - s0) <= 1e-4 * abs(s0), (s1, s0)

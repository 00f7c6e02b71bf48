"""debug: lazy_roll vs sweep step by step (first divergence)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from easyrec_amd import kernels
sys.path.insert(0, 'tests')
DEV='cuda:0'
hip=kernels.hip()
def _hyper(lr, t, beta1=0.9, beta2=0.999, eps=1e-8, gscale=1.0):
  row = np.zeros(kernels.HYPER_FLOATS, dtype=np.float32)
  f=np.float32
  row[kernels.HYPER_LR]=lr
  row[kernels.HYPER_LR_T]=f(lr)*np.sqrt(f(1)-f(beta2)**f(t))/(f(1)-f(beta1)**f(t))
  row[kernels.HYPER_BETA1]=beta1; row[kernels.HYPER_BETA2]=beta2; row[kernels.HYPER_OMB1]=f(1)-f(beta1); row[kernels.HYPER_OMB2]=f(1)-f(beta2)
  row[kernels.HYPER_EPS]=eps; row[kernels.HYPER_GSCALE]=gscale
  return torch.from_numpy(row)
rng = np.random.default_rng(11)
rows, dim, B, T = 97, 16, 6, 1300
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
table0 = torch.from_numpy((rng.standard_normal((rows, dim)) * 0.05).astype(np.float32))
ids_all = rng.integers(0, rows, size=(T, B)).astype(np.int64)
ids_all[60:, :] = ids_all[60:, :] % 11
dout_all = (rng.standard_normal((T, B, dim)) * 0.01).astype(np.float32)
S = {}
for mode in ('sweep', 'lazy_roll'):
  var, m, v = table0.clone().to(DEV), torch.zeros(rows, dim, device=DEV), torch.zeros(rows, dim, device=DEV)
  ids = torch.zeros(B, dtype=torch.int64, device=DEV); dout = torch.zeros(B, dim, device=DEV)
  bitmap = torch.zeros((rows + 31) // 32, dtype=torch.int32, device=DEV) if mode == 'sweep' else None
  spec = kernels.LookupSpec(table=var, ids=ids, offsets=None, weights=None, out=dout, out_col=0, rows=rows, key_base=0, dim=dim, combiner=0, n_rows=B, max_nnz=B)
  g = hip.emb_group_create([spec], dim, rows, var, m, v, bitmap)
  counter = torch.zeros(1, dtype=torch.int64, device=DEV)
  cap = T + 8
  hist = torch.zeros(2 * cap, device=DEV); hyper = torch.zeros(kernels.HYPER_FLOATS, device=DEV)
  if mode != 'sweep':
    last = torch.full((rows,), -1, dtype=torch.int32, device=DEV)
    hip.emb_group_enable_lazy_decay(g, last, hist, counter)
    ukeys = torch.zeros(B, dtype=torch.int32, device=DEV); nu = torch.zeros(1, dtype=torch.int32, device=DEV)
    uidx = torch.zeros(B, dtype=torch.int64, device=DEV); cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
  rows_h = torch.stack([_hyper(lr=1e-2 * (0.5**(s // 400)), t=s + 1) for s in range(T)]).to(DEV)
  hist[:T] = rows_h[:, kernels.HYPER_LR_T]; hist[cap:cap + T] = torch.cummax(rows_h[:, kernels.HYPER_LR_T], 0).values
  snaps = []
  print('mode', mode, flush=True)
  for s in range(T):
    if s % 100 == 0 or s < 3: torch.cuda.synchronize(); print(' step', s, flush=True)
    hyper.copy_(rows_h[s]); counter.fill_(s + 1)
    ids.copy_(torch.from_numpy(ids_all[s])); dout.copy_(torch.from_numpy(dout_all[s]))
    if mode != 'sweep':
      hip.emb_route(g, ukeys, nu, uidx, cnt); hip.emb_catch_up(g, ukeys, nu, hyper)
    hip.emb_bwd_update(g, kernels.OPT_ADAM, hyper)
    if mode == 'lazy_roll':
      hip.emb_flush_window([g], W, hyper)
      # bring a COPY current for comparison: flush on clones is not possible; compare only rows that are current
      torch.cuda.synchronize()
      snaps.append((var.cpu().clone(), m.cpu().clone(), v.cpu().clone(), last.cpu().clone()))
    else:
      torch.cuda.synchronize()
      snaps.append((var.cpu().clone(), m.cpu().clone(), v.cpu().clone(), None))
  S[mode] = snaps
for s in range(T):
  va, ma, sa, _ = S['sweep'][s]; vb, mb, sb, last = S['lazy_roll'][s]
  cur = (last == s)
  for name, a, b in (('var', va, vb), ('m', ma, mb), ('v', sa, sb)):
    bad = (a[cur] != b[cur]).any(dim=1)
    if bad.any():
      r = torch.nonzero(cur)[bad][0].item()
      j = torch.nonzero(a[r] != b[r])[0].item()
      print('first divergence step', s, name, 'row', r, 'col', j, 'sweep', a[r, j].item(), 'lazy', b[r, j].item(), 'window', (s + 1) % W, 'row in window', r // (-(-rows // W)))
      print('  history of that element: ')
      for q in range(max(0, s - 6), s + 1):
        print('   step', q, 'sweep m', S['sweep'][q][1][r, j].item(), 'v', S['sweep'][q][2][r, j].item(), 'var', S['sweep'][q][0][r, j].item(), '| lazy m', S['lazy_roll'][q][1][r, j].item(), 'v', S['lazy_roll'][q][2][r, j].item(), 'var', S['lazy_roll'][q][0][r, j].item(), 'last', S['lazy_roll'][q][3][r].item())
      sys.exit(0)
print('no divergence on current rows over', T, 'steps')

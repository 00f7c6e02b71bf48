#!/usr/bin/env python
"""Probe: per-workgroup stamps of emb_bwd_own_kernel inside one eager DeepFM step (er_debug_stamps; 100 MHz wall clock).
Round 5 layout of the launch: paired tiles (the dim-1 group rides on the dim-16 group's) first, then the one-row tables'
column-reduction workgroups of both groups."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging; logging.disable(logging.WARNING)
from easyrec_amd import kernels
from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
est = EasyRecEstimator('configs/deepfm_criteo.config', device='cuda:0', seed=1).build()
gen = SyntheticCriteo(est.pipeline_config.data_config, est.feature_configs, batch_size=4096, seed=3)
bs = [gen.next_batch() for _ in range(6)]
for b in bs[:5]: est.train_step(b)
torch.cuda.synchronize()
be = kernels.hip()
buf = torch.zeros(4096 * 16, dtype=torch.int64, device='cuda:0')
be.lib.er_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
est.train_step(bs[5]); torch.cuda.synchronize()
be.lib.er_debug_stamps(None)
full = buf.cpu().numpy().reshape(-1, 16)
used = full[:, 0] > 0
idx = np.where(used)[0]
t0 = full[used, 0].min()
st = (full[:, 0] - t0) / 100.0
en = (np.where(full[:, 1] > 0, full[:, 1], full[:, 0]) - t0) / 100.0
print('workgroups stamped', used.sum(), 'first start 0, last end %.2f us' % en[used].max())
paired = os.environ.get('EASYREC_AMD_PAIR_TILES', '1') != '0'
proj_first = os.environ.get('EASYREC_AMD_PROJ_FIRST', '1') != '0'
n_tiles, n_proj = 624, 2 * 13 * 16
t_lo = n_proj if proj_first else 0   # (stamps are indexed by the physical block id)
p_lo = 0 if proj_first else (n_tiles if paired else 2 * n_tiles)
ranges = ((t_lo, t_lo + n_tiles, 'tiles (paired)'),) if paired else \
    ((t_lo, t_lo + n_tiles, 'tiles D16'), (t_lo + n_tiles, t_lo + 2 * n_tiles, 'tiles D1'))
ranges += ((p_lo, p_lo + n_proj, 'one-row tables'),)
for lo, hi, name in ranges:
  sel = np.array([i for i in idx if lo <= i < hi])
  if len(sel):
    d = en[sel] - st[sel]
    print('%-16s n %4d | start min %.2f p50 %.2f max %.2f | end p50 %.2f max %.2f | dur mean %.2f p50 %.2f max %.2f' % (
        name, len(sel), st[sel].min(), np.median(st[sel]), st[sel].max(), np.median(en[sel]), en[sel].max(), d.mean(), np.median(d), d.max()))
sel = np.array([i for i in idx if t_lo <= i < t_lo + n_tiles and full[i, 5] > 0])
if len(sel):
  f = full[sel].astype(np.float64)
  ph = {'setup+keys': f[:, 2] - f[:, 0], 'gather': f[:, 3] - f[:, 2], 'barrier': f[:, 4] - f[:, 3], 'scan': f[:, 5] - f[:, 4], 'run ends': f[:, 1] - f[:, 5]}
  print('paired tile phases (us): ' + ' | '.join('%s mean %.2f max %.2f' % (k, v.mean() / 100.0, v.max() / 100.0) for k, v in ph.items()))
late = idx[np.argsort(-en[idx])][:10]
print('last to finish:', ' '.join('wg%d[%.1f-%.1f]' % (i, st[i], en[i]) for i in late))

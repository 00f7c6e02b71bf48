#!/usr/bin/env python
"""Probe: per-workgroup start / end stamps of emb_bwd_own_kernel inside one eager DeepFM step."""
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
t = full[:, :2]
used = t[:, 1] > 0
t0 = t[used, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
idx = np.where(used)[0]
print('workgroups stamped', used.sum(), 'span us', en[used].max())
dur = en - st
order = idx[np.argsort(-dur[idx])][:12]
for i in order: print('wg %4d start %7.2f end %7.2f dur %7.2f' % (i, st[i], en[i], dur[i]))
for lo, hi, name in ((0, 625, 'tiles D16'), (625, 1250, 'tiles D1'), (1250, 1666, 'proj')):
  sel = [i for i in idx if lo <= i < hi]
  if sel: print(name, 'n', len(sel), 'start min/max %.2f %.2f' % (st[sel].min(), st[sel].max()), 'end max %.2f' % en[sel].max(), 'dur mean %.2f max %.2f' % (dur[sel].mean(), dur[sel].max()))

print('per-phase (us) for the slowest and some typical tile workgroups: setup | chunk | emit | follow (chunks) | total')
for i in list(order[:6]) + [300, 400, 500, 620, 900, 1000, 1100, 1240]:
  f = full[i]
  if f[1] == 0 or f[2] == 0: continue
  print('wg %4d: %6.2f | %6.2f | %6.2f | %6.2f (%d) | %6.2f   chunk: keys %.2f gather %.2f barrier %.2f scan %.2f' % (i, (f[2] - f[0]) / 100.0, (f[3] - f[2]) / 100.0, (f[4] - f[3]) / 100.0,
        ((f[5] - f[4]) / 100.0) if f[5] else 0.0, f[6], (f[1] - f[0]) / 100.0, (f[8]-f[2])/100.0, (f[9]-f[8])/100.0, (f[10]-f[9])/100.0, (f[3]-f[10])/100.0))

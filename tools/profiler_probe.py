"""Does torch.profiler (kineto over roctracer / rocprofiler-sdk) report per-kernel durations of OUR library's launches
on this box - eager and inside a replayed hipGraph?  bench.py's per-kernel breakdown relies on it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyrec_amd.input.criteo_synthetic import SyntheticCriteo  # noqa: E402
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402

cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo.config'))
est = EasyRecEstimator(cfg, device='cuda', batch_size=4096, seed=1).build()
gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=4096, seed=3)
batches = [gen.next_batch() for _ in range(4)]
est.features.load(batches[0])
est.capture(warmup=3)
for b in batches:
  est.train_step(b)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
  for i in range(20):
    est.train_step(batches[i % 4])
  torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type is not None and 'cuda' in str(e.device_type).lower()]
print('device events:', len(ev))
agg = {}
for e in ev:
  a = agg.setdefault(e.name[:90], [0, 0.0])
  a[0] += 1
  a[1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
  print('%8.1f us/step  x%5.1f  %s' % (us / 20, n / 20, name))
print('total us/step', sum(v[1] for v in agg.values()) / 20)

import sys, os, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.WARNING)
import numpy as np, torch
from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
from easyrec_amd.utils import config_util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs/deepfm_criteo_small.config'))
B = 256
if len(sys.argv) > 1 and sys.argv[1] == 'lazy':
  oc = cfg.train_config.optimizer_config[0]
  oc.lazy_adam_optimizer.learning_rate.CopyFrom(oc.adam_optimizer.learning_rate)
gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B)
batches = [gen.next_batch() for _ in range(6)]
mk = lambda: EasyRecEstimator(cfg, device='cuda:0', batch_size=B, seed=2).build()
a, b, c = mk(), mk(), mk()
for e in (a, b, c):
  e.features.load(batches[0])
def diff(x, y, tag):
  sx, sy = x.state_dict(slots=True), y.state_dict(slots=True)
  bad = [(k, float(np.max(np.abs(sx[k] - sy[k])))) for k in sx if not np.array_equal(sx[k], sy[k])]
  print(tag, 'differing tensors:', len(bad), bad[:6])
diff(a, b, 'init a-b')
for i in range(3):
  a.train_step(); b.train_step()
  diff(a, b, 'eager step %d a-b' % i)
c.capture(warmup=3)
diff(a, c, 'after warmup a-c')
for i, bt in enumerate(batches[1:]):
  for e in (a, b, c):
    e.train_step(bt)
  diff(a, b, 'step %d a-b' % i)
  diff(a, c, 'step %d a-c' % i)
print('---- buffers after the last step')
def cmp(name, x, y):
  print('%-40s equal=%s maxdiff=%g' % (name, torch.equal(x, y), float((x.float() - y.float()).abs().max())))
cmp('hash_ids', a.features.hash_ids, c.features.hash_ids)
cmp('raw_block', a.features.raw_block, c.features.raw_block)
cmp('labels', a.features.labels, c.features.labels)
cmp('hyper', a.hyper, c.hyper)
cmp('step_counter', a.step_counter, c.step_counter)
for k in a.engine.groups:
  cmp('out ' + k, a.engine.groups[k]['out'], c.engine.groups[k]['out'])
  cmp('dout ' + k, a.engine.groups[k]['dout'], c.engine.groups[k]['dout'])
cmp('flat_grad', a.varstore.flat_grad, c.varstore.flat_grad)

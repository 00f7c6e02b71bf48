"""How many of the reference's shipped configs BUILD (estimator constructed, one forward + backward on a tiny synthetic
batch) on the oracle's stand-in backend, and why the others do not.  Run where /root/reference exists.
usage: python tools/count_reference_configs.py [/root/reference]"""
import collections
import glob
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
logging.disable(logging.CRITICAL)
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'


def main():
  from easyrec_amd import kernels
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  from oracle.kernel_ref import RefBackend
  kernels._BACKEND = RefBackend()
  files = sorted(glob.glob(REF + '/samples/model_config/*.config') + glob.glob(REF + '/examples/configs/*.config'))
  ok, why = [], collections.Counter()
  failed = {}
  for f in files:
    name = os.path.basename(f)
    try:
      cfg = config_util.get_configs_from_pipeline_file(f)
      # shrink the tables: the count is about structure
      for fc in list(cfg.feature_config.features) + list(cfg.feature_configs):
        if fc.hash_bucket_size > 2000:
          fc.hash_bucket_size = 2000
        if fc.num_buckets > 2000:
          fc.num_buckets = 2000
      est = EasyRecEstimator(cfg, device='cpu', batch_size=8, seed=1).build()
      gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=8, seed=2)
      est.train_step(gen.next_batch())
      ok.append(name)
    except BaseException as e:  # noqa
      msg = '%s: %s' % (type(e).__name__, str(e).split('\n')[0][:110])
      why[msg] += 1
      failed[name] = msg
  print('%d of %d build and step' % (len(ok), len(files)))
  for msg, n in why.most_common(40):
    print('%4d  %s' % (n, msg))
  if '-v' in sys.argv:
    for k, v in sorted(failed.items()):
      print(k, '->', v)


if __name__ == '__main__':
  main()

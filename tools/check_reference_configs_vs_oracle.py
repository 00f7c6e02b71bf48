"""For every config the reference ships that builds here: two training steps of the product (host code on the oracle's
stand-in kernels) against the model oracle, on the same synthetic batches - loss by loss.  Run where /root/reference
exists.  usage: python tools/check_reference_configs_vs_oracle.py [/root/reference]"""
import collections
import glob
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
logging.disable(logging.CRITICAL)
REF = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else '/root/reference'


def main():
  from easyrec_amd import kernels
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  from oracle.kernel_ref import RefBackend
  from oracle.model_oracle import OracleTrainer
  kernels._BACKEND = RefBackend()
  files = sorted(glob.glob(REF + '/samples/model_config/*.config') + glob.glob(REF + '/examples/configs/*.config'))
  agree, differ, no_product, no_oracle = [], [], 0, collections.Counter()
  for f in files:
    name = os.path.basename(f)
    try:
      cfg = config_util.get_configs_from_pipeline_file(f)
      for fc in list(cfg.feature_config.features) + list(cfg.feature_configs):
        fc.hash_bucket_size = min(fc.hash_bucket_size, 2000) if fc.HasField('hash_bucket_size') else fc.hash_bucket_size
        if fc.num_buckets > 2000:
          fc.num_buckets = 2000
        for dr in fc.ListFields():
          pass
      est = EasyRecEstimator(cfg, device='cpu', batch_size=16, seed=1).build()
      gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=16, seed=2)
      batches = [gen.next_batch() for _ in range(2)]
      state = est.state_dict()
    except BaseException:  # noqa
      no_product += 1
      continue
    try:
      orc = OracleTrainer(cfg, state, batch_size=16)
      worst = 0.0
      for b in batches:
        est.train_step(b)
        got, exp = est.loss_values(), orc.train_step(b)
        assert sorted(got) == sorted(exp), (sorted(got), sorted(exp))
        worst = max(worst, max(abs(got[k] - exp[k]) / max(1.0, abs(exp[k])) for k in exp))
      (agree if worst <= 1e-4 else differ).append((name, worst))
    except BaseException as e:  # noqa
      no_oracle['%s: %s' % (type(e).__name__, str(e).split('\n')[0][:90])] += 1
      if '-v' in sys.argv:
        print('  (oracle) %s: %s: %s' % (name, type(e).__name__, str(e).split('\n')[0][:120]))
  print('%d configs; %d do not build in the product; of the %d that do: %d agree with the oracle over 2 steps (1e-4), '
        '%d differ, %d the oracle does not restate' % (len(files), no_product, len(files) - no_product, len(agree), len(differ),
                                                       sum(no_oracle.values())))
  for name, w in differ:
    print('  DIFFER %-60s %.3g' % (name, w))
  for msg, n in no_oracle.most_common(12):
    print('  oracle: %3d  %s' % (n, msg))


if __name__ == '__main__':
  main()

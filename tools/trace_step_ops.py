#!/usr/bin/env python
"""What one training step launches, in order: every library (C ABI) call and every torch-native (aten) op with the
innermost easyrec_amd source line that issued it.  The aten ops are the ones to get rid of (torch-native
kernels inside the step).  usage: python tools/trace_step_ops.py [config] [batch]"""
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logging  # noqa: E402

logging.disable(logging.WARNING)
from easyrec_amd import kernels  # noqa: E402
from easyrec_amd.input.synthetic import SyntheticBatches  # noqa: E402
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator  # noqa: E402

LOG = []


def _site():
  for fr in reversed(traceback.extract_stack()[:-2]):
    if 'easyrec_amd' in fr.filename and 'kernels.py' not in fr.filename:
      return '%s:%d' % (os.path.relpath(fr.filename, ROOT), fr.lineno)
  for fr in reversed(traceback.extract_stack()[:-2]):
    if 'easyrec_amd' in fr.filename:
      return '%s:%d' % (os.path.relpath(fr.filename, ROOT), fr.lineno)
  return '?'


class Aten(TorchDispatchMode):

  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    name = str(func)
    if not any(s in name for s in ('aten.view', 'aten.slice', 'aten.as_strided', 'aten.detach', 'aten.select',
                                   'aten.reshape', 'aten._unsafe_view', 'aten.alias', 'aten.t.', 'aten.expand',
                                   'aten.unsqueeze', 'aten.squeeze', 'aten.empty', 'aten.transpose', 'aten.permute',
                                   'aten._local_scalar', 'aten.is_', 'aten.stride', 'aten.sym_', 'aten.unbind',
                                   'aten.split')):
      LOG.append(('aten', name, _site()))
    return func(*args, **(kwargs or {}))


def main():
  cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'configs', 'deepfm_criteo.config')
  B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
  est = EasyRecEstimator(cfg, device='cuda:0', batch_size=B, seed=1).build()
  gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=B, seed=5)
  b = gen.next_batch()
  est.train_step(b)
  est.train_step(b)
  be = kernels.hip()
  for name in dir(be):
    fn = getattr(be, name)
    if callable(fn) and not name.startswith('_') and name not in ('wgrad_sink', 'gemm_row_tiles', 'require_device', 'device_info'):
      def wrap(fn=fn, name=name):
        def inner(*a, **k):
          LOG.append(('lib', name, _site()))
          return fn(*a, **k)
        return inner
      setattr(be, name, wrap())
  with Aten():
    est.features.version += 1
    est._refresh_hyper()
    est._device_step()
  torch.cuda.synchronize()
  n_lib = sum(1 for k, _, _ in LOG if k == 'lib')
  n_aten = len(LOG) - n_lib
  print('# %d library calls, %d aten ops in one step of %s' % (n_lib, n_aten, os.path.basename(cfg)))
  for kind, name, site in LOG:
    print('%-5s %-44s %s' % (kind, name, site))


if __name__ == '__main__':
  main()

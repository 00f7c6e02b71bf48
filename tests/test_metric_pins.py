"""easyrec_amd/core/metrics.py's grouped AUC and max-F1 against the REFERENCE'S OWN core/metrics.py (gauc / session_auc
-> _separated_auc_impl with sklearn's roc_auc_score; max_f1), run by tests/golden/make_metric_vectors.py where
/root/reference exists: a stream of four batches, users with a single class (skipped), tied predictions, the three
reductions, string keys."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'metric_vectors.npz'))


@pytest.mark.parametrize('reduction', ['mean', 'mean_by_sample_num', 'mean_by_positive_num'])
def test_gauc_stream(reduction):
  from easyrec_amd.core.metrics import gauc
  m = gauc(reduction)
  for b in range(4):
    m.update(G['labels_%d' % b], G['preds_%d' % b], G['keys_%d' % b])
    assert abs(m.result() - float(G['gauc_%s_after_%d' % (reduction, b)])) <= 1e-6, (reduction, b)


def test_session_auc_with_string_keys():
  from easyrec_amd.core.metrics import session_auc
  m = session_auc()
  m.update(G['labels_0'], G['preds_0'], G['session_keys'])
  assert abs(m.result() - float(G['session_auc'])) <= 1e-6


def test_max_f1_stream():
  from easyrec_amd.core.metrics import MaxF1
  m = MaxF1()
  for b in range(4):
    m.update(G['labels_%d' % b], G['preds_%d' % b])
    assert abs(m.result() - float(G['max_f1_after_%d' % b])) <= 1e-6, b


def _device_stream(device):
  """DeviceSeparatedAUC (what evaluate() uses: rows kept and reduced on the device, er_grouped_auc) on the reference's
  own numbers: the four-batch stream with the three reductions, and string keys."""
  from easyrec_amd.core.metrics import DeviceSeparatedAUC
  for reduction in ('mean', 'mean_by_sample_num', 'mean_by_positive_num'):
    m = DeviceSeparatedAUC(reduction, device)
    for b in range(4):
      m.update(G['labels_%d' % b], G['preds_%d' % b], G['keys_%d' % b])
      assert abs(m.result() - float(G['gauc_%s_after_%d' % (reduction, b)])) <= 1e-6, (reduction, b)
  m = DeviceSeparatedAUC('mean', device)
  m.update(G['labels_0'], G['preds_0'], G['session_keys'])
  assert abs(m.result() - float(G['session_auc'])) <= 1e-6
  assert DeviceSeparatedAUC('mean', device).result() == 0.0


def test_device_grouped_auc_on_the_stand_in_backend(ref_backend):
  _device_stream('cpu')


def _device_max_f1(device):
  from easyrec_amd.core.metrics import DeviceMaxF1
  m = DeviceMaxF1(200, device)
  for b in range(4):
    m.update(G['labels_%d' % b], G['preds_%d' % b])
    assert abs(m.result() - float(G['max_f1_after_%d' % b])) <= 1e-6, b


def test_device_max_f1_on_the_stand_in_backend(ref_backend):
  _device_max_f1('cpu')


@pytest.mark.gpu
def test_device_max_f1_on_the_gpu():
  _device_max_f1('cuda:0')


@pytest.mark.gpu
def test_device_grouped_auc_on_the_gpu():
  import torch
  from easyrec_amd.core.metrics import DeviceSeparatedAUC, SeparatedAUC
  _device_stream('cuda:0')
  # a larger stream against the host implementation: 200k rows, 5000 users, rounded predictions (ties inside users)
  rng = np.random.default_rng(11)
  n = 200000
  users = rng.integers(0, 5000, size=n) * 1000003 - 7
  y = (rng.random(n) < 0.2).astype(np.float32)
  p = np.round(np.clip(rng.normal(0.4 + 0.1 * y, 0.2), 0, 1), 2).astype(np.float32)
  for reduction in SeparatedAUC.REDUCTIONS:
    host, dev = SeparatedAUC(reduction), DeviceSeparatedAUC(reduction, 'cuda:0')
    for lo in range(0, n, 50000):
      host.update(y[lo:lo + 50000], p[lo:lo + 50000], users[lo:lo + 50000])
      dev.update(torch.from_numpy(y[lo:lo + 50000]).cuda(), torch.from_numpy(p[lo:lo + 50000]).cuda(), users[lo:lo + 50000])
    assert abs(host.result() - dev.result()) <= 1e-6, (reduction, host.result(), dev.result())


def test_key_codes_are_consistent_across_batches_and_types():
  """Grouping only needs equality: integer keys pass through, strings / bytes get a 64-bit digest that is the same in
  every batch (so a user's rows meet across update() calls) and differs between different keys."""
  import torch
  from easyrec_amd.core.metrics import DeviceSeparatedAUC
  ints = DeviceSeparatedAUC.key_codes(np.array([5, -3, 2 ** 40], dtype=np.int64))
  assert ints.tolist() == [5, -3, 2 ** 40] and ints.dtype == torch.int64
  a = DeviceSeparatedAUC.key_codes(np.array([b'u1', b'u2', b'u1', b''], dtype=object))
  b = DeviceSeparatedAUC.key_codes(np.array(['u2', 'u1'], dtype=object))
  assert a[0] == a[2] and a[0] != a[1] and a[3] not in (a[0], a[1])
  assert b[0] == a[1] and b[1] == a[0], 'bytes and str of the same text are one key'
  assert DeviceSeparatedAUC.key_codes(torch.tensor([[1, 2], [3, 4]])).tolist() == [1, 2, 3, 4]

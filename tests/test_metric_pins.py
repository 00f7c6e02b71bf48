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

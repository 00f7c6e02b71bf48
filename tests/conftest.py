import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

REFERENCE = '/root/reference'


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)')


def reference_available():
  return os.path.isdir(os.path.join(REFERENCE, 'easy_rec'))


@pytest.fixture
def ref_backend(monkeypatch):
  """Route easyrec_amd's kernel calls to the CPU oracle (host-logic tests only)."""
  from easyrec_amd import kernels
  from oracle.kernel_ref import RefBackend
  be = RefBackend()
  monkeypatch.setattr(kernels, '_BACKEND', be)
  return be


@pytest.fixture(scope='session')
def built_lib():
  """Make sure libeasyrec_hip.so exists (cross-compiled here, prebuilt on the GPU box)."""
  from easyrec_amd import kernels
  if not os.path.exists(kernels.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  return kernels.LIB_PATH

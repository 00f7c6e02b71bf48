"""The C-ABI library loads on a machine without a GPU and exports every symbol declared in
include/easyrec_hip.h (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'easyrec_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(er_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_hot_path():
  syms = declared_symbols()
  for must in ('er_hash_bucket_fast', 'er_emb_fwd', 'er_emb_bwd_update', 'er_fm_fwd', 'er_cross_v1_fwd',
               'er_cross_v2_epilogue_fwd', 'er_din_pool_fwd', 'er_bn_act_fwd', 'er_sigmoid_ce_fwd_bwd',
               'er_mmoe_mix_fwd', 'er_dense_opt_step', 'er_adam_decay_sweep'):
    assert must in syms


def test_library_exports_every_declared_symbol(built_lib):
  lib = ctypes.CDLL(built_lib)
  missing = [s for s in declared_symbols() if not hasattr(lib, s)]
  assert not missing, 'declared in include/easyrec_hip.h but not exported: %s' % missing
  assert lib.er_abi_version() == 1


def test_struct_layout_matches_header(built_lib):
  from easyrec_amd import kernels
  # er_lookup_desc: 5 pointers + 2 int64 + 7 int32 = 40 + 16 + 28 = 84, padded to the 8-byte alignment: 88 bytes
  assert ctypes.sizeof(kernels.LookupDesc) == 88
  assert kernels.HYPER_FLOATS == 16


def test_product_does_not_import_the_oracle():
  """easyrec_amd/ must never import oracle/ (the product path has no CPU fallback)."""
  bad = []
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'easyrec_amd')):
    for f in files:
      if f.endswith('.py'):
        src = open(os.path.join(dirpath, f)).read()
        if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
          bad.append(os.path.join(dirpath, f))
  assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
  import pytest
  from easyrec_amd import kernels
  monkeypatch.setattr(kernels, 'LIB_PATH', str(tmp_path / 'nope.so'))
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    kernels.HipBackend()

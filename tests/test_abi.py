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


MIRRORS = {  # ctypes class in easyrec_amd/kernels.py -> the C struct it mirrors
    'GemmEpilogue': 'er_gemm_epilogue', 'LookupDesc': 'er_lookup_desc', 'KvJob': 'er_kv_job', 'KvRouteJob': 'er_kv_route_job', 'CastDesc': 'er_cast_desc',
    'GemmProblem': 'er_gemm_problem', 'BnLayer': 'er_bn_layer', 'CeHead': 'er_ce_head', 'TailJob': 'er_tail_job',
    'LossTailJob': 'er_loss_tail_job', 'DenseOptJob': 'er_dense_opt_job', 'GradTerm': 'er_grad_term',
    'GradGroup': 'er_grad_group', 'DenseApplyDesc': 'er_dense_apply_desc', 'ColsumJob': 'er_colsum_job',
}


RENAMED = {('GradGroup', 'lam'): 'lambda'}  # (a Python keyword on the C side)


def test_every_ctypes_mirror_has_the_size_and_field_offsets_of_its_c_struct(tmp_path):
  """The host side hands the library arrays of records: a ctypes mirror that drifts from include/easyrec_hip.h (a field
  dropped on one side only) would corrupt every call silently.  The header is compiled as C (gcc) into a probe that prints
  sizeof and every field's offsetof; the mirrors must agree field by field, in order."""
  import shutil
  import subprocess
  from easyrec_amd import kernels
  if shutil.which('gcc') is None:
    import pytest
    pytest.skip('no gcc')
  lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "easyrec_hip.h"', 'int main(void) {']
  for cls_name, c_name in MIRRORS.items():
    cls = getattr(kernels, cls_name)
    lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (c_name, c_name))
    for field in cls._fields_:
      fname = field[0]
      if fname.endswith('_') and fname.startswith('pad'):
        continue
      lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (c_name, fname, c_name, RENAMED.get((cls_name, fname), fname)))
  lines += ['  return 0;', '}']
  src = tmp_path / 'probe.c'
  src.write_text('\n'.join(lines))
  exe = tmp_path / 'probe'
  subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
  out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
  got = {}
  for ln in out.splitlines():
    c_name, key, val = ln.split()
    got[(c_name, key)] = int(val)
  for cls_name, c_name in MIRRORS.items():
    cls = getattr(kernels, cls_name)
    assert ctypes.sizeof(cls) == got[(c_name, 'size')], (cls_name, ctypes.sizeof(cls), got[(c_name, 'size')])
    for field in cls._fields_:
      fname = field[0]
      if (c_name, fname) in got:
        assert getattr(cls, fname).offset == got[(c_name, fname)], (cls_name, fname)

#!/usr/bin/env python
"""Times er_gemm_bf16_nt (bf16 operands in HBM) against er_gemm_bf16 (fp32 operands rounded while staged) and
er_gemm_f32 on the contraction shapes of the DCN-v2 Criteo step (B = 4096) with HIP events around back-to-back launches
(the launch gap is included: these are upper bounds of the kernel durations; rocprofv3 gives the durations proper)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easyrec_amd import kernels  # noqa: E402

be = kernels.hip()
be._ck(be.lib.er_gemm_bf16_nt_prepare(), 'prepare')
dev = 'cuda:0'
pad = kernels.Bf16Shadows.pad8


def timed(fn, n=50):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(n):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / n * 1e3


nt_only = len(sys.argv) > 1 and sys.argv[1] == 'nt_only'
shapes = [(4096, 624, 624), (4096, 256, 624), (4096, 768, 256), (4096, 128, 256), (4096, 1024, 1024), (8192, 2048, 2048),
          (4096, 4096, 4096)]
for M, N, K in shapes:
  a, bt = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
  a16 = torch.empty(M, pad(K), dtype=torch.bfloat16, device=dev)
  b16 = torch.empty(N, pad(K), dtype=torch.bfloat16, device=dev)
  be.cast_bf16([(a, a16, False), (bt, b16, False)])
  out = torch.empty(M, N, device=dev)
  fl = 2.0 * M * N * K
  t_nt = timed(lambda: be.gemm_bf16_nt(a16, b16, M, N, K, out=out))
  if nt_only:
    print('M=%5d N=%5d K=%5d | bf16_nt %7.1f us %7.1f TF/s' % (M, N, K, t_nt, fl / t_nt / 1e6))
    continue
  t_cast = timed(lambda: be.cast_bf16([(a, a16, False)]))
  t_old = timed(lambda: be.gemm(kernels.GEMM_NT, a, bt, out=out, bf16=True))
  t_f32 = timed(lambda: be.gemm(kernels.GEMM_NT, a, bt, out=out))
  print('M=%5d N=%5d K=%5d | bf16_nt %7.1f us %7.1f TF/s | cast A %6.1f us | er_gemm_bf16 %7.1f us %6.1f TF/s | f32 %7.1f us %6.1f TF/s'
        % (M, N, K, t_nt, fl / t_nt / 1e6, t_cast, t_old, fl / t_old / 1e6, t_f32, fl / t_f32 / 1e6))

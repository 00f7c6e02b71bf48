#!/usr/bin/env python
"""Times er_gemm_f32 / er_gemm_bf16 on the GEMM shapes of the DeepFM-Criteo step (B=4096) with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easyrec_amd import kernels  # noqa: E402

be = kernels.hip()
be.gemm_reserve(1 << 24)
dev = 'cuda:0'
B = 4096
layers = [(624, 256), (256, 128), (128, 64), (81, 256), (64, 1)]
shapes = []
for k, n in layers:
  shapes.append(('NN fwd', kernels.GEMM_NN, (B, k), (k, n)))
  shapes.append(('NT dx ', kernels.GEMM_NT, (B, n), (k, n)))
  shapes.append(('TN dW ', kernels.GEMM_TN, (B, k), (B, n)))
only = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else None  # indices into the shape list
f32_only = len(sys.argv) > 2
tot = {False: 0.0, True: 0.0}
for si, (name, layout, sa, sb) in enumerate(shapes):
  if only is not None and si not in only:
    continue
  a, b = torch.randn(sa, device=dev), torch.randn(sb, device=dev)
  if layout == kernels.GEMM_NN:
    M, N, K = sa[0], sb[1], sa[1]
  elif layout == kernels.GEMM_NT:
    M, N, K = sa[0], sb[0], sa[1]
  else:
    M, N, K = sa[1], sb[1], sa[0]
  line = '%s M=%5d N=%4d K=%5d ' % (name, M, N, K)
  for bf16 in ((False,) if f32_only else (False, True)):
    for _ in range(5):
      be.gemm(layout, a, b, bf16=bf16)
    torch.cuda.synchronize()
    n = 30
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
      be.gemm(layout, a, b, bf16=bf16)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    tot[bf16] += us
    line += ' | %s %7.1f us %6.1f TF/s' % ('bf16' if bf16 else 'f32 ', us, 2.0 * M * N * K / us / 1e6)
  print(line)
print('sum f32 %.1f us, bf16 %.1f us (back-to-back launches, includes launch gaps)' % (tot[False], tot[True]))

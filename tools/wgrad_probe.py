#!/usr/bin/env python
"""Times the grouped TN launch (weight gradients dW = x^T . dz) on single problems - DIN's attention MLP (K = 204,800
rows), MMoE's expert layers (K = 8,192) - with HIP events.  --only i: one shape (for rocprofv3 --pmc passes).  (Round 3
compared the default 64 x 64 kernel with a natural-layout TN kernel here: profiles/r03_wgrad_probe.md; that kernel was
removed from the library in round 5.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easyrec_amd import kernels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--only', type=int, default=-1)
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
be = kernels.hip()
be.gemm_reserve(1 << 25)
dev = 'cuda:0'
shapes = [(128, 128, 204800), (128, 64, 204800), (64, 32, 204800), (32, 1, 204800), (1152, 256, 8192), (256, 192, 8192),
          (128, 128, 8192), (624, 256, 4096)]
for si, (M, N, K) in enumerate(shapes):
  if args.only >= 0 and si != args.only:
    continue
  a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
  out = torch.zeros(M, N, device=dev)
  line = 'TN M=%5d N=%4d K=%6d ' % (M, N, K)
  for mode in (0,):
    for _ in range(3):
      be.gemm_grouped(kernels.GEMM_TN, [(a, b, out, None, False)])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
      be.gemm_grouped(kernels.GEMM_TN, [(a, b, out, None, False)])
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / args.iters * 1e3
    line += ' | mode %d %7.1f us %6.1f TF/s %5.2f TB/s' % (mode, us, 2.0 * M * N * K / us / 1e6, 4.0 * K * (M + N) / us / 1e6)
  print(line, flush=True)

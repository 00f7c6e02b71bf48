#!/usr/bin/env python
"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md: on gfx950
FETCH_SIZE reads half the bytes of a wide coalesced stream; other patterns are uncalibrated).  Launches
er_stream_copy (exactly `bytes` read + `bytes` written, the sweep's nontemporal float4 pattern) and the D=16
dense-decay sweep on a table of known size.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easyrec_amd import kernels  # noqa: E402

be = kernels.hip()
dev = 'cuda:0'
n = 1 << 28  # 1 GiB of fp32
src = torch.randn(n, device=dev)
dst = torch.empty_like(src)
for _ in range(5):
  be.stream_copy(src, dst)
rows, dim = 4000000, 16
var, m, v = (torch.randn(rows, dim, device=dev) for _ in range(3))
v.abs_()
hyper = torch.zeros(kernels.HYPER_FLOATS, device=dev)
hyper[kernels.HYPER_LR_T], hyper[kernels.HYPER_BETA1], hyper[kernels.HYPER_BETA2], hyper[kernels.HYPER_EPS] = 1e-3, 0.9, 0.999, 1e-8
bitmap = torch.zeros((rows + 31) // 32, dtype=torch.int32, device=dev)
for _ in range(5):
  be.adam_decay_sweep(var, m, v, bitmap, rows, dim, hyper)
torch.cuda.synchronize()
print('copy bytes each way', n * 4, 'sweep bytes each way', rows * dim * 4 * 3)

# random 64-byte rows (the embedding kernels' access pattern): er_gather_rows of `g_rows` DISTINCT rows of a table far
# larger than the Infinity Cache (4 M x 16 floats = 256 MB per table; three tables cycled so nothing is resident)
g_rows = 1 << 20
tables = [torch.randn(rows, dim, device=dev) for _ in range(3)]
perm = torch.randperm(rows, device=dev)[:g_rows].to(torch.int32)
out = torch.empty(g_rows, dim, device=dev)
for i in range(6):
  be.gather_rows(tables[i % 3], perm, g_rows, 0, out)
torch.cuda.synchronize()
print('gather rows', g_rows, 'bytes read', g_rows * dim * 4, 'written', g_rows * dim * 4)

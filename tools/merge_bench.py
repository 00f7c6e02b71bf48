#!/usr/bin/env python
"""Time er_emb_owner_merge / er_emb_owner_serve against the sorted form on one GPU (keys of one embedding-parallel step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from easyrec_amd import kernels

hip = kernels.hip()
DEV = 'cuda'
rng = np.random.default_rng(0)
rows, dim = 1000000, 16
for counts in ([45000], [5600] * 8, [700] * 64):
  m = sum(counts)
  cap = m + 1000
  ids = np.full(cap, -1, dtype=np.int64)
  ids[:m] = np.concatenate([np.sort(rng.choice(rows, size=c, replace=False)) for c in counts])
  ids_dev = torch.from_numpy(ids).to(DEV)
  var = torch.zeros(rows, dim, device=DEV)
  grads = torch.zeros(cap, dim, device=DEV)
  out = torch.zeros(cap, dim, device=DEV)
  spec = kernels.LookupSpec(table=var, ids=ids_dev, offsets=None, weights=None, out=grads, out_col=0, rows=rows, key_base=0,
                            dim=dim, combiner=0, n_rows=cap, max_nnz=cap)
  g = hip.emb_group_create([spec], dim, rows, var, None, None, None)
  hip.emb_group_set_active(g, m)
  uk = torch.zeros(cap, dtype=torch.int32, device=DEV)
  nu = torch.zeros(1, dtype=torch.int32, device=DEV)

  def timed(fn, n=20):
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
      fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

  t_merge = timed(lambda: hip.emb_owner_merge(g, counts))
  t_both = timed(lambda: (hip.emb_owner_merge(g, counts), hip.emb_owner_serve([g], [out], None)))
  t_sort = timed(lambda: hip.emb_route(g, uk, nu, None, None))
  print('runs %3d x %6d: merge(+build) %7.1f us  merge+serve %7.1f us  route (radix sort form) %7.1f us' %
        (len(counts), counts[0], t_merge, t_both, t_sort))

#!/usr/bin/env python
"""Benchmark of the hot path: DeepFM training step on synthetic Criteo-shape input.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" = one pass of the hot path over one batch already resident in HBM: device-side id hashing
-> fused embedding lookup (78 lookups, one launch) -> wide sum / FM / MLP(+BatchNorm) -> sigmoid CE
-> backward -> sort-based de-duplicated embedding gradient + row-wise optimizer -> dense optimizer.
Workload at N=1 = BASELINE.json configs[1]: configs/deepfm_criteo.config (39 features, 26 x 1M-row
hashed tables, D=16, batch 4096, fp32, `adam_optimizer` with TF's dense-decay sparse apply).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COMPACT_ORACLE_ABOVE_BYTES = 48 * 2 ** 30  # tables + Adam slots beyond this: parity_full_size fetches rows by id
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable by a float4 copy


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'deepfm_criteo.config'))
  ap.add_argument('--batch_size', type=int, default=0, help='per-GPU batch (default: data_config.batch_size)')
  ap.add_argument('--ids', default='zipf', choices=['zipf', 'uniform'])
  ap.add_argument('--optimizer', default='config', choices=['config', 'adam', 'lazy_adam'])
  ap.add_argument('--no_graph', action='store_true', help='eager launches instead of hipGraph replay')
  ap.add_argument('--dense_dtype', default='f32', choices=['f32', 'bf16'],
                  help="bf16: BASELINE config 3 (bf16 dense contractions with fp32 accumulate, fp32 embeddings and master weights)")
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--dense_sweep', action='store_true', help='TF-exact Adam by streaming every row every step (default: lazy dense decay, bit-identical)')
  ap.add_argument('--force_ep', action='store_true', help='run the embedding-parallel code path even at 1 GPU')
  ap.add_argument('--overlap', action='store_true', help='TF-exact Adam: dense-decay sweep on a second stream (measured slower)')
  ap.add_argument('--cpu_seconds', type=float, default=12.0, help='time budget of the CPU baseline sample')
  ap.add_argument('--rccl', action='store_true',
                  help='with --force_ep at 1 GPU: issue the collectives through a world-1 RCCL process group (not local copies)')
  ap.add_argument('--ring', type=int, default=256, help='distinct device-generated batches the timed loop cycles over')
  ap.add_argument('--precondition', type=int, default=1024,
                  help='untimed steps over as many further distinct batches BEFORE the warm-up, so that the timed steps '
                       'see tables in their steady state (rows with Adam history and realistic idle lengths) instead of '
                       'freshly initialised ones, whose lazy dense decay is trivially cheap; 0 = skip')
  ap.add_argument('--steady_steps', type=int, default=2048,
                  help='N=1: after the timed region, this many further steps over as many DISTINCT device-generated '
                       'batches (realistic idle lengths for the lazy dense decay); 0 = skip')
  ap.add_argument('--from_file', default='', choices=['', 'csv', 'criteo'],
                  help='N=1, Criteo-shaped configs: after the device-resident measurement, train from FILES written to a temp '
                       'directory - Input -> pack (prefetch thread) -> train_step - and report the end-to-end examples/s as an '
                       'extra key (never the headline value)')
  ap.add_argument('--parity_only', action='store_true',
                  help='with --no_cpu_baseline: still run the full-size GPU-vs-oracle parity check (the 200 M-row config: the '
                       'oracle then holds only the rows the parity batches read - no CPU timing is possible on that state)')
  ap.add_argument('--parity_steps', type=int, default=2,
                  help='N=1: steps of the full-size GPU-vs-oracle loss comparison (0 = skip; needs the CPU baseline)')
  return ap.parse_args()


def workload_name(cfg, args):
  base = os.path.basename(args.config)
  name = 'DeepFM' if 'deepfm' in base else 'DCN-v2' if 'dcn_v2' in base else cfg.model_config.model_class or base
  return name + (' (bf16 dense, fp32 embeddings)' if args.dense_dtype == 'bf16' else '')


def workload_text(cfg, args, est, criteo, B, graph_note, n_ring):
  base = os.path.basename(args.config)
  opt = est.opt_emb.name + (' (dense sweep)' if getattr(est, 'dense_sweep', False) and est.opt_emb.name == 'adam_optimizer'
                            else ' (lazy dense decay, closed-form replay)' if getattr(est, 'decay_tables', None) is not None
                            else ' (lazy dense decay, exact step-by-step replay + rolling flush)'
                            if est.opt_emb.name == 'adam_optimizer' else '')
  if criteo:
    head = '%s synthetic Criteo: %s (39 features: 26 hashed x %d rows + 13 projected, D=16%s, ' % (
        workload_name(cfg, args), base, cfg.feature_config.features[13].hash_bucket_size,
        ' deep + D=1 wide' if 'deepfm' in base else '')
    src = 'distinct device-generated batches'
  else:
    feats = cfg.feature_config.features
    rows = sum((f.hash_bucket_size or f.num_buckets) for f in feats)
    seqs = [f.max_seq_len for f in feats if f.HasField('max_seq_len')]
    head = '%s synthetic Taobao-shaped: %s (%d features, %.1f M embedding rows, D=%d%s, ' % (
        workload_name(cfg, args), base, len(feats), rows / 1e6, feats[0].embedding_dim,
        ', %d sequences of max length %d' % (len(seqs), max(seqs)) if seqs else '')
    src = 'distinct host-generated, device-resident batches (pre-conditioning cycles over the same ring)'
  return head + 'batch %d per GPU, optimizer %s, ids %s, %s); timed steps cycle over %d %s after %d untimed ' \
      'pre-conditioning steps' % (B, opt, args.ids, graph_note, n_ring, src, args.precondition)


def is_criteo_shaped(cfg):
  names = [f.input_names[0] for f in cfg.feature_config.features if len(f.input_names)]
  return 'C1' in names and 'C26' in names and not any(f.HasField('max_seq_len') or f.feature_type == f.TagFeature
                                                      for f in cfg.feature_config.features)


class RingSource(object):
  """next_packed() over a fixed ring of device-resident batches (schemas DeviceCriteo does not generate)."""

  def __init__(self, ring):
    self.ring, self.i = ring, 0

  def next_packed(self):
    self.i += 1
    return self.ring[self.i % len(self.ring)]


def baseline_metric():
  """BASELINE.json's metric string (the embedding-stage GB/s half of it is reported under `embedding_stage`)."""
  try:
    return json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
  except Exception:  # noqa: BLE001
    return 'global examples/sec + emb HBM GB/s, DeepFM Criteo-shape b4096 @1/2/4/8 GPU'


def self_spawn(n_gpus):
  """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (what the driver's
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` does) and hand their output through."""
  import socket
  import subprocess
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr',
         '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  sys.exit(subprocess.call(cmd, env=env))


def dist_setup(n_gpus, rccl_world1=False):
  import torch.distributed as dist
  if n_gpus > 1 and 'WORLD_SIZE' not in os.environ:
    self_spawn(n_gpus)
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  assert world == n_gpus or world == 1 and n_gpus == 1, 'WORLD_SIZE %d != --gpus %d' % (world, n_gpus)
  torch.cuda.set_device(local)
  if world > 1 or rccl_world1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
  return rank, world, local


def switch_optimizer(cfg, which):
  oc = cfg.train_config.optimizer_config[0]
  cur = oc.WhichOneof('optimizer')
  want = {'adam': 'adam_optimizer', 'lazy_adam': 'lazy_adam_optimizer'}[which]
  if cur == want:
    return
  lr = getattr(oc, cur).learning_rate
  getattr(oc, want).learning_rate.CopyFrom(lr)


def to_device_batch(batch, device):
  return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in batch.items()}


def embedding_bytes_per_step(est, batches):
  """Algorithmic bytes of the embedding stage per step:
     per valid lookup fwd 8+8D, bwd 8+4D; per unique row update 8D*(1+S); dense-decay Adam adds the
     sweep R*D*4*6 over every table.  -> (lookup + update bytes, sweep bytes, {kernel: its share per launch}): the
     lookup kernel reads id + row and writes the row; the segmented reduction reads id + upstream gradient per lookup
     and reads + writes var, m, v of every distinct row; the catch-up reads + writes var, m, v of every distinct row;
     the sort reads the id and writes key + permutation entry."""
  from easyrec_amd import kernels
  lazy, n_steps, per = 0.0, 0, {}
  S = 2 if est.opt_emb.kind in (kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM) else (1 if est.opt_emb.kind == kernels.OPT_ADAGRAD else 0)
  for b in batches[:4]:
    est.features.load(b)
    est.features.transform()
    torch.cuda.synchronize()
    ids = est.features.hash_ids.cpu().numpy()
    tot = 0.0
    for dim in est.engine.storage:
      n_valid = int((ids >= 0).sum())
      uniq = sum(len(np.unique(r[r >= 0])) for r in ids)
      tot += n_valid * (16 + 12 * dim) + uniq * 8 * dim * (1 + S)
      n_proj = len(est.schema.raw)
      tot += n_proj * est.batch_size * (4 + 8 * dim) + n_proj * 8 * dim * (1 + S)
      per['emb_fwd_kernel'] = per.get('emb_fwd_kernel', 0.0) + n_valid * (8 + 8 * dim) + n_proj * est.batch_size * (4 + 4 * dim)
      per['emb_bwd_tile_multi_kernel'] = per.get('emb_bwd_tile_multi_kernel', 0.0) + n_valid * (8 + 4 * dim) + \
          uniq * 8 * dim * (1 + S) + n_proj * est.batch_size * 4 * dim + n_proj * 8 * dim * (1 + S)
      per['emb_catch_up_closed_kernel'] = per.get('emb_catch_up_closed_kernel', 0.0) + (uniq + n_proj) * 8 * dim * (1 + S)
    per['emb_segment_sort_kernel'] = per.get('emb_segment_sort_kernel', 0.0) + (n_valid + n_proj * est.batch_size) * 16
    lazy += tot
    n_steps += 1
  lazy /= max(n_steps, 1)
  sweep = 0.0
  if est.opt_emb.kind == kernels.OPT_ADAM and est.dense_sweep:
    sweep = sum(st['total_rows'] * dim * 4 * 6 for dim, st in est.engine.storage.items())
  per = {k: v / max(n_steps, 1) for k, v in per.items()}
  per['emb_catch_up_multi_kernel'] = per['emb_catch_up_closed_kernel']
  # the fused single-GPU embedding step (er_emb_front / er_emb_bwd_fused) does the same work under other names: the
  # catch-up from the per-lookup key lists, the segmented reduction + row update with the gradient finish folded in,
  # the sort with the entry build in front of it (the id hash itself runs inside the step prologue's launch)
  per['emb_catch_up_heads_kernel'] = per['emb_catch_up_closed_kernel']
  per['emb_bwd_own_kernel'] = per['emb_bwd_tile_multi_kernel']
  per['emb_front_sort_kernel'] = per['emb_segment_sort_kernel']
  # round 5: sort + lookup in one launch, the lookup reading whole row records (lazy decay evaluated in registers)
  per['emb_front_fwd_kernel'] = per['emb_segment_sort_kernel'] + per['emb_fwd_kernel']
  per['emb_fwd_lazy_kernel'] = per['emb_fwd_kernel']
  return lazy, sweep, per


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X fp32 MFMA (= fp32 vector) peak, MI355X_MICROARCH.md


def time_sweep_kernel(est, launches):
  """--dense_sweep: average duration of the D=16 dense-decay sweep, HIP events on the launch stream (torch's
  current stream is the stream the C ABI launches on)."""
  from easyrec_amd import kernels
  be = kernels.hip()
  dim = max(est.engine.storage, key=lambda d: est.engine.storage[d]['total_rows'] * d)
  st = est.engine.storage[dim]
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
  for _ in range(3):
    be.adam_decay_sweep(st['var'], st['m'], st['v'], st['bitmap'], st['total_rows'], dim, est.hyper[0])
  torch.cuda.synchronize()
  for a, b in evs:
    a.record()
    be.adam_decay_sweep(st['var'], st['m'], st['v'], st['bitmap'], st['total_rows'], dim, est.hyper[0])
    b.record()
  torch.cuda.synchronize()
  ms = [a.elapsed_time(b) for a, b in evs]
  alg_bytes = st['total_rows'] * dim * 4 * 6
  return {'kernel': 'er::adam_decay_sweep_vec4_kernel<4> (dim %d)' % dim, 'avg_ms': float(np.mean(ms)),
          'min_ms': float(np.min(ms)), 'bytes': alg_bytes, 'launches': launches}


def time_gemm_kernel(est, launches):
  """Default (lazy dense decay / lazy Adam): the largest share of the step is the forward GEMM kernel
  er::gemm_f32_kernel<NN>; its biggest launch - the first deep layer, [B, 624] x [624, 256] with the bias and the
  BatchNorm column statistics in the epilogue, exactly as the step issues it - is timed live with HIP events."""
  from easyrec_amd import kernels
  be = kernels.hip()
  vs = est.varstore
  bf16 = est.ctx.dense_dtype == 'bf16'
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]

  def timed(fn):
    for _ in range(5):
      fn()
    torch.cuda.synchronize()
    for a, e in evs:
      a.record()
      fn()
      e.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(e) for a, e in evs]

  if not bf16 and 'deep_feature/dnn_0/kernel' in vs._vars and 'group:deep' in est.engine.groups:
    x = est.engine.groups['group:deep']['out']
    w = vs._vars['deep_feature/dnn_0/kernel']['tensor'].detach()
    b = vs._vars['deep_feature/dnn_0/bias']['tensor'].detach()
    M, K = x.shape
    N = w.shape[1]
    stats = torch.empty(be.gemm_row_tiles(M) * N * 3, dtype=torch.float32, device=x.device)
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    ms = timed(lambda: be.gemm(kernels.GEMM_NN, x, w, out=out, bias=b, col_stats=stats))
    name = 'er::gemm_f32_kernel<NN> %dx%dx%d (+bias, BatchNorm column statistics)' % (M, N, K)
  else:
    # the largest weight matrix of the dense part (DCN-v2: a 624 x 624 cross kernel), on a batch-sized activation
    wname, w = max(((n, r['tensor'].detach()) for n, r in vs._vars.items() if r['tensor'].dim() == 2),
                   key=lambda nw: nw[1].numel())
    K, N = w.shape
    M = est.batch_size
    x = torch.randn(M, K, device=w.device)
    out = torch.empty(M, N, dtype=torch.float32, device=w.device)
    if bf16 and getattr(est, '_bf16', None) is not None and be.bf16_nt:
      x16 = est._bf16.act(x)
      wt = est._bf16.weight(w)[2]
      ms = timed(lambda: be.gemm_bf16_nt(x16, wt, M, N, K, out=out))
      name = 'er::gemm_bf16_nt_kernel<4> %dx%dx%d (%s: bf16 operands in HBM, fp32 out)' % (M, N, K, wname)
    else:
      ms = timed(lambda: be.gemm(kernels.GEMM_NN, x, w, out=out, bf16=bf16))
      name = 'er::gemm_%s_kernel<NN> %dx%dx%d (%s)' % ('bf16' if bf16 else 'f32', M, N, K, wname)
  return {'kernel': name, 'avg_ms': float(np.mean(ms)), 'min_ms': float(np.min(ms)), 'flops': 2.0 * M * N * K,
          'launches': launches}


FAMILIES = (  # first match wins; names as rocprofv3 / the profiler print them
    ('decay replay', ('catch_up', 'flush_window', 'flush_decay', 'flush_mark', 'decay_tables', 'adam_decay_sweep')),
    # the step's tail in one grid (er_emb_bwd_fused_wgrad): the dense layers' weight gradients NEXT TO the embedding row
    # update, the split-K reduce next to the cross-tile fix - neither the GEMM family's nor the embedding family's alone
    ('tail (weight gradients + embedding update)', ('emb_bwd_own_wgrad', 'emb_bwd_fix_reduce', 'emb_bwd_fix_opt',
                                                     'emb_reduce_local_wgrad')),
    ('collectives (RCCL)', ('rccl', 'nccl')),
    ('gemm', ('gemm_', 'gemv_', 'wgrad_narrow')),
    ('batchnorm', ('er::bn_', 'dice', 'colsum')),
    ('embedding', ('er::emb_', 'hash_bucket', 'group_grad_finish', 'er::kv_', 'gather_rows', 'scatter_unique', 'rocprim')),
    ('interaction', ('fm_', 'cross_', 'din_', 'mmoe_', 'cin_', 'dot_interaction', 'rowsum', 'concat_cols', 'cast_bf16')),
    ('loss / optimizer / prologue', ('er::',)),
    ('input copy', ('Memcpy', 'copyBuffer')),
)


def family_of(name):
  for fam, pats in FAMILIES:
    if any(p in name for p in pats):
      return fam
  return 'other'


def kernel_breakdown(est, ring, n_steps=32):
  """Per-kernel device time of the STEP AS BENCHMARKED (hipGraph replays over ring batches), from torch.profiler's
  device activity records (roctracer kernel timestamps - what rocprofv3 --kernel-trace reports), plus the
  algorithmic flops behind every GEMM kernel from one eager step with the backend's op log on.
  -> ({kernel name: (launches per step, us per step)}, {kernel name: flops per step})"""
  from torch.profiler import ProfilerActivity, profile
  from easyrec_amd import kernels
  for i in range(4):
    est.train_step(ring[i % len(ring)])
  torch.cuda.synchronize()
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for i in range(n_steps):
      est.train_step(ring[(4 + i) % len(ring)])
    torch.cuda.synchronize()
  agg = {}
  for e in prof.events():
    if 'cuda' not in str(getattr(e, 'device_type', '')).lower():
      continue
    us = getattr(e, 'device_time', None)
    if us is None:
      us = getattr(e, 'cuda_time', 0.0)
    a = agg.setdefault(e.name, [0, 0.0])
    a[0] += 1
    a[1] += float(us)
  per_kernel = {k: (v[0] / n_steps, v[1] / n_steps) for k, v in agg.items()}
  be = kernels.hip()
  saved_graph, saved_graphs = est.graph, getattr(est, '_graphs', None)
  est.graph = None
  if saved_graphs is not None:
    est._graphs = None
  be.op_log = []
  try:
    est.train_step(ring[0])
    torch.cuda.synchronize()
    flops = {}
    for name, f in be.op_log:
      flops[name] = flops.get(name, 0.0) + f
  finally:
    be.op_log = None
    est.graph = saved_graph
    if saved_graphs is not None:
      est._graphs = saved_graphs
  return per_kernel, flops


def short_name(name):
  return name if len(name) <= 96 else name[:93] + '...'


def step_roofline(est, per_kernel, flops, emb_bytes, pmc):
  """The bench line's `roofline`: the kernel with the largest total duration per step (chosen from the step's own
  per-kernel timings), its algorithmic work and counter traffic, the kernel families, and the embedding stage (the
  second half of BASELINE.json's metric) as a fraction of the HBM peak."""
  total = sum(us for _, us in per_kernel.values())
  fams = {}
  for name, (n, us) in per_kernel.items():
    f = fams.setdefault(family_of(name), [0.0, 0.0])
    f[0] += n
    f[1] += us
  compute = {k: v for k, v in per_kernel.items() if family_of(k) not in ('input copy', 'other')}
  dom = max(compute, key=lambda k: compute[k][1])
  n, us = per_kernel[dom]
  avg_ms = us / max(n, 1e-9) * 1e-3
  peak_tf = 2500.0 if 'bf16' in dom else MFMA_F32_PEAK_TFLOPS
  # flops of a logged name: exact kernel-name prefix match (the profiler prints `void er::gemm_f32_kernel<true, false>(...)`)
  f_dom = sum(f for k, f in flops.items() if k in dom)
  out = {'kernel': short_name(dom), 'launches_per_step': n, 'us_per_step': us, 'share_of_step_kernel_time': us / total,
         'avg_kernel_ms': avg_ms,
         'selection': 'largest total device time per step among the step\'s kernels; durations from the profiler\'s '
                      'device activity records (roctracer kernel timestamps, = rocprofv3 --kernel-trace) over graph '
                      'replays of the benchmarked step'}
  if f_dom > 0:
    ach = f_dom / (us * 1e-6) / 1e12
    out.update({'bound': 'mfma', 'achieved': ach, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': ach / peak_tf,
                'algorithmic_flops_per_launch': f_dom / max(n, 1e-9), 'algorithmic_flops_per_step': f_dom})
  else:
    b = (emb_bytes.get(dom_key(dom)) or emb_bytes.get(dom_key(dom).split('<')[0])) if emb_bytes else None
    ach = (b / (us * 1e-6) / 1e9) if b else None
    out.update({'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (ach / HBM_PEAK_GBS) if ach else None, 'algorithmic_bytes_per_launch': (b / max(n, 1e-9)) if b else None})
  t = (pmc or {}).get('by_kernel', {}).get(dom_key(dom))
  out['traffic'] = t.get('bytes_per_launch') if t else None
  if t:
    out['traffic_source'] = t.get('source')
  out['families'] = [{'family': k, 'us_per_step': v[1], 'share': v[1] / total, 'launches_per_step': v[0]}
                     for k, v in sorted(fams.items(), key=lambda kv: -kv[1][1])]
  out['kernels'] = [{'kernel': short_name(k), 'launches_per_step': v[0], 'us_per_step': v[1]}
                    for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:14]]
  out['kernel_time_us_per_step'] = total
  TAIL = 'tail (weight gradients + embedding update)'
  tail_flops = sum(f for k, f in flops.items() if family_of(k) == TAIL)
  gemm_flops = sum(f for k, f in flops.items() if family_of(k) == 'gemm')
  gemm_us = fams.get('gemm', [0, 0.0])[1]
  if TAIL in fams:
    out['tail'] = {'us_per_step': fams[TAIL][1], 'launches_per_step': fams[TAIL][0], 'gemm_flops_per_step': tail_flops,
                   'note': 'two launches: [grouped weight-gradient GEMM | embedding gradient finish + segmented reduce + row '
                           'update | scalar loss tail] and [cross-tile fix | dense optimizer with the split-K reduce folded '
                           'into its gradient read], each in one grid; the GEMM family below excludes these flops and this time'}
  if gemm_flops > 0 and gemm_us > 0:
    out['gemm_family'] = {'flops_per_step': gemm_flops, 'us_per_step': gemm_us, 'TFLOPs': gemm_flops / (gemm_us * 1e-6) / 1e12,
                          'frac_of_mfma_peak': gemm_flops / (gemm_us * 1e-6) / 1e12 / peak_tf}
  if emb_bytes and emb_bytes.get('stage'):
    stage_us = fams.get('embedding', [0, 0.0])[1] + fams.get('decay replay', [0, 0.0])[1] + fams.get(TAIL, [0, 0.0])[1]
    gbps = emb_bytes['stage'] / (stage_us * 1e-6) / 1e9
    out['embedding_stage'] = {'algorithmic_bytes_per_step': emb_bytes['stage'], 'us_per_step': stage_us, 'GBps': gbps,
                              'frac_of_hbm_peak': gbps / HBM_PEAK_GBS,
                              'note': 'hash + sort + catch-up + lookup + gradient finish + segmented reduction + row '
                                      'update: the sum of those kernels\' durations INSIDE the replayed graph'
                                      + (' - the WHOLE duration of the tail launches is counted although the dense '
                                         'layers\' weight gradients run in the same grids (a lower bound of the stage\'s '
                                         'rate; embedding_stage_alone: the stage\'s own launches, tail unfused)'
                                         if TAIL in fams else '')}
  return out


def tail_unfused(est, ring, emb_bytes, fused_per_kernel):
  """The step with the tail as its four launches (grouped weight-gradient GEMM, split-K reduce, embedding backward, fix)
  in a second replayed graph: what each half costs alone, and the embedding stage's rate from its own kernels."""
  from easyrec_amd import kernels
  be = kernels.hip()
  saved = est.graph
  be.fused_tail = False
  try:
    est.graph = None
    est.capture(warmup=1)
    per_kernel, _ = kernel_breakdown(est, ring)
  finally:
    del be.fused_tail
    est.graph = saved
  fams = {}
  for name, (n, us) in per_kernel.items():
    f = fams.setdefault(family_of(name), [0.0, 0.0])
    f[0] += n
    f[1] += us
  total = sum(us for _, us in per_kernel.values())
  pick = lambda pat: sum(us for k, (n, us) in per_kernel.items() if pat in k)  # noqa: E731
  out = {'kernel_time_us_per_step': total,
         'kernel_time_us_per_step_fused': sum(us for _, us in fused_per_kernel.values()),
         'wgrad_gemm_us': pick('gemm_f32_grouped_kernel<false, false>'), 'splitk_reduce_us': pick('gemm_splitk_reduce_grouped'),
         'emb_bwd_own_us': pick('emb_bwd_own_kernel'), 'emb_bwd_fix_us': pick('emb_bwd_fix_multi_kernel'),
         'loss_tail_us': pick('loss_tail_kernel'), 'dense_opt_us': pick('dense_opt_kernel')}
  if emb_bytes and emb_bytes.get('stage'):
    stage_us = fams.get('embedding', [0, 0.0])[1] + fams.get('decay replay', [0, 0.0])[1]
    gbps = emb_bytes['stage'] / (stage_us * 1e-6) / 1e9
    out['embedding_stage_alone'] = {'us_per_step': stage_us, 'GBps': gbps, 'frac_of_hbm_peak': gbps / HBM_PEAK_GBS}
  return out


def dom_key(name):
  """`void er::gemm_f32_kernel<true, false>(er::GemmArgs)` -> `gemm_f32_kernel<true, false>` (the keys of
  profiles/pmc_traffic.json `by_kernel` and of embedding_bytes_per_step)"""
  return name.split('(')[0].replace('void ', '').replace('er::', '').strip()


class DeviceCriteo(object):
  """SyntheticCriteo's distribution (Zipf(1.05) ids over per-feature vocabularies, 8-hex-digit strings,
  2 % empty, labels ~ Bernoulli(0.25), raw = u^3 scaled) generated ON THE DEVICE straight into the packed arena image
  DeviceFeatures.load takes: thousands of distinct batches cost milliseconds instead of 20 ms of host time each."""

  def __init__(self, host_gen, features, device, seed):
    self.f, self.dev = features, device
    self.B = host_gen.B
    self.g = torch.Generator(device=device)
    self.g.manual_seed(seed)
    self.n_hash = host_gen.n_hash
    self.mix = torch.from_numpy(host_gen.mix.astype(np.int64)).to(device)
    self.mode, self.empty_frac = host_gen.mode, host_gen.empty_frac
    self.cdf = {}
    for V in sorted(set(host_gen.vocab)):
      p = torch.arange(1, V + 1, dtype=torch.float64, device=device) ** (-1.05)
      self.cdf[V] = torch.cumsum(p / p.sum(), 0)
    self.vocab = list(host_gen.vocab)
    self.raw_rows = [(host_gen.schema.raw[n]['row'], float(fc.min_val), float(fc.max_val)) for n, fc in host_gen.raw_cfg]
    self.hex = torch.tensor(list(b'0123456789abcdef'), dtype=torch.uint8, device=device)
    self.shifts = torch.arange(28, -4, -4, dtype=torch.int64, device=device)

  def next_packed(self):
    f, B, dev, g = self.f, self.B, self.dev, self.g
    end = f._layout['str_bytes'][0] + f._layout['str_bytes'][1]
    img = torch.zeros(end, dtype=torch.uint8, device=dev)

    def section(name):
      o, nbytes, shape, dt = f._layout[name]
      return img[o:o + nbytes].view(dt).view(shape)

    section('labels')[0] = (torch.rand(B, device=dev, generator=g) < 0.25).float()
    raw = section('raw_block')
    for row, lo, hi in self.raw_rows:
      u = torch.rand(B, device=dev, generator=g, dtype=torch.float64)
      x = (lo + (hi - lo) * u ** 3).float()
      raw[row] = (x - lo) / (hi - lo) if hi > lo else x
    vals = torch.empty(self.n_hash, B, dtype=torch.int64, device=dev)
    for i in range(self.n_hash):
      if self.mode == 'uniform':
        vals[i] = torch.randint(0, 2 ** 32, (B,), device=dev, generator=g, dtype=torch.int64)
      else:
        r = torch.searchsorted(self.cdf[self.vocab[i]], torch.rand(B, device=dev, generator=g, dtype=torch.float64))
        vals[i] = (r * self.mix[i]) & 0xFFFFFFFF
    n = self.n_hash * B
    nib = (vals.view(-1, 1) >> self.shifts.view(1, -1)) & 0xF
    chars = self.hex[nib]  # [n, 8]
    empty = torch.rand(n, device=dev, generator=g) < self.empty_frac
    offs = section('str_offsets')
    offs[1:] = torch.cumsum(torch.where(empty, 0, 8), 0)
    body = chars[~empty].reshape(-1)
    section('str_bytes')[:body.numel()] = body
    return {'packed': img, 'packed_has_strings': True}


def steady_state(est, gen, n_steps):
  """n_steps further training steps over n_steps DISTINCT batches (a real epoch: most rows of a 1M-row table stay idle
  for hundreds to thousands of steps, so the lazy dense decay's catch-up replays long histories), per-step wall time
  from HIP events between graph replays, the catch-up launches alone over an eager tail, and one er_emb_flush_decay
  (what a checkpoint / evaluation owes)."""
  t0 = time.perf_counter()
  batches = [gen.next_packed() for _ in range(n_steps)]
  torch.cuda.synchronize()
  gen_s = time.perf_counter() - t0
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
  evs[0].record()
  for i, b in enumerate(batches):
    est.train_step(b)
    evs[i + 1].record()
  torch.cuda.synchronize()
  ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_steps)])
  p50 = float(np.percentile(ms, 50))
  slow = np.nonzero(ms > 1.3 * p50)[0]
  out = {'steps': n_steps, 'distinct_batches': n_steps, 'batch_generation_s': gen_s,
         'ms_per_step_mean': float(ms.mean()), 'ms_per_step_p50': p50,
         'ms_per_step_p99': float(np.percentile(ms, 99)), 'ms_per_step_last_256_mean': float(ms[-256:].mean()),
         'ms_per_step_max': float(ms.max()),
         # steps slower than 1.3 x the median, where they sit and what they cost in all: a tail that is a few isolated
         # steps (the host refreshing the per-step scalar table every HYPER_SLOTS / 2 steps behind a stream
         # synchronisation, a clock dip) reads differently from one that is a drift of every step
         'slow_steps': {'count': int(slow.size), 'first_indices': [int(i) for i in slow[:12]],
                        'excess_ms_total': float((ms[slow] - p50).sum()) if slow.size else 0.0,
                        'mean_without_them': float(np.delete(ms, slow).mean()) if slow.size < ms.size else None,
                        'hyper_table_refresh_every': int(getattr(est, 'HYPER_SLOTS', 0) // 2)},
         'ms_per_step_quarters_mean': [float(q.mean()) for q in np.array_split(ms, 4)],
         'examples_per_s': float(est.batch_size / (ms.mean() * 1e-3))}
  eng = est.engine
  if getattr(eng, 'lazy_decay', False):
    # the catch-up launches alone: eager forward passes over a tail of further distinct batches
    tail = [gen.next_packed() for _ in range(64)]
    cu = []
    saved_graph, est.graph = est.graph, None
    try:
      for b in tail:
        probe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        eng.catch_up_probe = probe
        est.train_step(b)
        torch.cuda.synchronize()
        cu.append(probe[0].elapsed_time(probe[1]))
    finally:
      eng.catch_up_probe = None
      est.graph = saved_graph
    out['catch_up_ms_p50'] = float(np.percentile(cu, 50))
    out['catch_up_ms_p99'] = float(np.percentile(cu, 99))
    out['catch_up_note'] = 'the general path\'s catch-up launches between HIP events, eager; a fused single-GPU step has no ' \
                           'catch-up launch since round 5 (the lookup and the row update evaluate a row\'s pending decay in registers)' 
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.flush_decay()
    b.record()
    torch.cuda.synchronize()
    out['flush_decay_ms'] = float(a.elapsed_time(b))
    out['flush_decay_note'] = 'one er_emb_flush_decay over every table group: owed once per checkpoint / evaluation'
  return out


def gpu_clocks():
  """What `rocm-smi --showclocks` says right after the timed region (boxes of this pool differ by up to 20 % on the same
  tree: the clocks beside every number tell box variance from a regression)."""
  import re
  import subprocess
  try:
    txt = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
    out = {}
    for line in txt.splitlines():
      m = re.search(r'GPU\[(\d+)\]\s*:\s*(\w+) clock level:\s*\d+:?\s*\(?(\d+)\s*Mhz\)?', line, flags=re.I)
      if m and m.group(1) == '0':
        out[m.group(2).lower() + '_mhz'] = int(m.group(3))
    return out or {'raw': txt.strip()[-400:]}
  except Exception as e:  # noqa: BLE001
    return {'error': str(e)[:120]}


def from_file_rate(cfg, est, kind, B, n_batches=48, steps=96):
  """End to end from files: `Input -> pack (background thread, input/prefetch.py) -> train_step` over a Criteo-layout
  file set written to a temp directory (reference input/csv_input.py:33-76, input/criteo_binary_reader.py) - what the device
  step sustains when the host feeds it, beside the device-resident headline."""
  import shutil
  import tempfile
  from easyrec_amd.input.prefetch import Prefetcher
  tmp = tempfile.mkdtemp(prefix='er_bench_')
  rng = np.random.default_rng(7)
  n = n_batches * B
  try:
    feats = list(cfg.feature_config.features) or list(cfg.feature_configs)
    if kind == 'csv':
      from easyrec_amd.input.input import Input
      path = os.path.join(tmp, 'train.tsv')
      vocab = np.array(['%08x' % v for v in rng.integers(0, 2 ** 32, size=200000)])
      ints = rng.integers(0, 1000, size=(n, 13)).astype(str)
      cats = vocab[np.minimum((rng.pareto(1.05, size=(n, 26))).astype(np.int64), len(vocab) - 1)]
      labels = (rng.random(n) < 0.25).astype(np.int64).astype(str)
      sep = cfg.data_config.separator or '\t'
      with open(path, 'w') as f:
        for i in range(n):
          f.write(sep.join([labels[i]] + list(ints[i]) + list(cats[i])) + '\n')
      inp = Input.create(cfg.data_config, feats, path, batch_size=B, hash_on_host=False)
    else:
      from easyrec_amd.input.criteo_input import CriteoInput
      (rng.random(n) < 0.25).astype(np.int32).tofile(os.path.join(tmp, 'p0_label.bin'))
      rng.random((n, 13), dtype=np.float32).tofile(os.path.join(tmp, 'p0_dense.bin'))
      rng.integers(0, 2 ** 32, size=(n, 26), dtype=np.uint32).tofile(os.path.join(tmp, 'p0_category.bin'))
      inp = CriteoInput(cfg.data_config, feats,
                        {'label_path': [os.path.join(tmp, 'p0_label.bin')], 'dense_path': [os.path.join(tmp, 'p0_dense.bin')],
                         'category_path': [os.path.join(tmp, 'p0_category.bin')]}, batch_size=B)
    epochs = (steps + 16) // n_batches + 2
    # where the time goes, per batch: the producer thread's read + decode (`input_ms`) and pack (`pack_ms`), the consumer's
    # wait for the producer (`wait_ms`) and its host time inside train_step (`step_host_ms`: copy + graph launch)
    stage = {'input': 0.0, 'pack': 0.0, 'wait': 0.0, 'step_host': 0.0, 'n_in': 0}

    def timed_source(src):
      src = iter(src)
      while True:
        t = time.perf_counter()
        try:
          b = next(src)
        except StopIteration:
          return
        stage['input'] += time.perf_counter() - t
        stage['n_in'] += 1
        yield b

    def timed_pack(b):
      t = time.perf_counter()
      out = est.features.pack(b)
      stage['pack'] += time.perf_counter() - t
      return out

    it = Prefetcher(timed_source(inp.batches(num_epochs=epochs, drop_remainder=True) if kind == 'csv' else
                                 (b for _ in range(epochs) for b in inp.batches())), depth=4, transform=timed_pack)
    k, t0 = 0, None
    while True:
      tw = time.perf_counter()
      try:
        b = next(it)
      except StopIteration:
        break
      if k >= 16:
        stage['wait'] += time.perf_counter() - tw
      if b['labels'].shape[0] != B if isinstance(b, dict) and 'labels' in b else False:
        continue
      if k == 16:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
      ts = time.perf_counter()
      est.train_step(b)
      if k >= 16:
        stage['step_host'] += time.perf_counter() - ts
      k += 1
      if k == 16 + steps:
        break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it.close()
    done = k - 16
    return {'format': kind, 'examples_per_s': done * B / dt, 'ms_per_step': dt / done * 1e3, 'steps': done,
            'host_threads': os.cpu_count(), 'file_batches': n_batches,
            'stage_ms_per_batch': {'input': stage['input'] / max(stage['n_in'], 1) * 1e3,
                                   'pack': stage['pack'] / max(stage['n_in'], 1) * 1e3,
                                   'consumer_wait': stage['wait'] / max(done, 1) * 1e3,
                                   'train_step_host': stage['step_host'] / max(done, 1) * 1e3},
            'note': 'Input -> pack in one background thread (depth 4) -> one packed host-to-device copy + the replayed step; '
                    'the device-resident rate is `value`'}
  finally:
    shutil.rmtree(tmp, ignore_errors=True)


def parity_full_size(cfg, est, ring, host_batches, batch_size, n_steps):
  """The GPU path and the CPU oracle from the SAME full-size training state (weights, Adam slots, step), the same
  batches: relative loss differences over n_steps steps.  Tolerance: north_star's 1e-4 for fp32; with bf16 dense
  contractions (operands carry 8 significant bits; north_star states no bar for them) 1e-3 against the fp32 oracle, the bar
  tests/test_models_gpu.py holds the bf16 step to."""
  from oracle.model_oracle import OracleTrainer
  tables = {n: (t['rows'], t['dim']) for n, t in est.engine.tables.items() if not t.get('kv')}
  table_bytes = sum(r * d * 4 for r, d in tables.values())
  compact = None
  if table_bytes * 3 > COMPACT_ORACLE_ABOVE_BYTES:
    # tables (+ Adam's two slots) too large for a host copy - BASELINE config 5 at its stated 200 M rows x 64 is 153 GB: the
    # oracle runs over the rows the parity batches read, fetched by id.  Exactly the full oracle's losses (a row no lookup
    # reads influences nothing; TF-Adam's every-row decay acts row by row): tests/test_compact_oracle.py
    dense_only = {k: v for k, v in est.varstore.state_dict().items()}
    compact = OracleTrainer(cfg, dense_only, batch_size=batch_size).probe_ids(
        [host_batches[k % len(host_batches)] for k in range(n_steps)], tables)
  state = est.state_dict(slots=True, rows_of=compact)
  weights = {k: v for k, v in state.items() if not (k.endswith('/m') or k.endswith('/v'))}
  orc = OracleTrainer(cfg, weights, batch_size=batch_size, compact_ids=compact)
  orc.resume(est.global_step, {k: v for k, v in state.items() if k.endswith('/m') or k.endswith('/v')})
  del state, weights
  worst, per_step = 0.0, []
  for k in range(n_steps):
    est.train_step(ring[k % len(ring)])
    got = est.loss_values()
    exp = orc.train_step(host_batches[k % len(host_batches)])
    d = max(abs(got[n] - exp[n]) / max(abs(exp[n]), 1e-3) for n in exp)
    per_step.append({'gpu_total_loss': got['total_loss'], 'oracle_total_loss': exp['total_loss'], 'max_rel_diff': d})
    worst = max(worst, d)
  tol = 1e-3 if getattr(est.ctx, 'dense_dtype', 'f32') == 'bf16' else 1e-4
  return {'max_rel_loss_diff': worst, 'steps': n_steps, 'tolerance': tol, 'ok': bool(worst <= tol),
          'oracle_tables': 'full' if compact is None else
          'compact: the %d rows (of %d) the parity batches read, fetched by id' % (sum(len(v) for v in compact.values()),
                                                                                  sum(r for r, _ in tables.values())),
          'per_step': per_step, 'from_global_step': int(est.global_step - n_steps)}, orc


def cpu_baseline(cfg, est_state, batches, batch_size, budget_s=12.0, max_steps=8, orc=None):
  """The CPU restatement of the reference path (oracle/model_oracle.py) on the host cores: a bounded
  sample (about `budget_s` seconds) of the same workload."""
  from oracle.model_oracle import OracleTrainer
  threads = min(os.cpu_count() or 1, 64)
  torch.set_num_threads(threads)
  if orc is None:
    orc = OracleTrainer(cfg, est_state, batch_size=batch_size)
    orc.train_step(batches[0])  # warm-up (first-touch of the optimizer slots)
  t0 = time.perf_counter()
  steps = 0
  while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
    orc.train_step(batches[(steps + 1) % len(batches)])
    steps += 1
  dt = time.perf_counter() - t0
  return {
      'value': steps * batch_size / dt,
      'unit': 'examples/s',
      'cores': threads,
      'kind': 'port',
      'sample': '%d steps of batch %d in %.1f s (same config, same synthetic batches, same optimizer semantics, fp32); '
                'torch-CPU ops on %d threads; CPU restatement of the reference path - TensorFlow itself is not '
                'installable here' % (steps, batch_size, dt, threads),
  }


def main():
  args = parse_args()
  logging.disable(logging.WARNING)
  rank, world, local = dist_setup(args.gpus, args.rccl and args.force_ep)
  dev = torch.device('cuda', local)
  from easyrec_amd import kernels
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util

  cfg = config_util.get_configs_from_pipeline_file(args.config)
  if args.optimizer != 'config':
    switch_optimizer(cfg, args.optimizer)
  B = args.batch_size or cfg.data_config.batch_size
  if world > 1 or args.force_ep:
    from easyrec_amd.model.embedding_parallel import EmbeddingParallelEstimator
    comm = None
    if args.rccl and world == 1:
      from easyrec_amd.core.comm import TorchDistComm
      comm = TorchDistComm()
    est = EmbeddingParallelEstimator(cfg, device=dev, batch_size=B, seed=1, rank=rank, world=world, comm=comm,
                                     dense_sweep=args.dense_sweep, dense_dtype=args.dense_dtype).build()
  else:
    est = EasyRecEstimator(cfg, device=dev, batch_size=B, seed=1, overlap_sweep=args.overlap,
                           dense_sweep=args.dense_sweep, dense_dtype=args.dense_dtype).build()
  criteo = is_criteo_shaped(cfg)
  pack = lambda b: {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if isinstance(v, np.ndarray) else v)  # noqa: E731
                    for k, v in est.features.pack(b, device=dev).items()}
  if criteo:
    gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=B, seed=20240607 + rank, mode=args.ids)
    # a few host batches (the CPU baseline and the full-size parity check need the same batch on both sides) ...
    host_batches = [gen.next_batch() for _ in range(4)]
    host_ring = [pack(b) for b in host_batches]
    # ... and the timed loop's batches: the same distribution generated on the device, resident in the packed layout of
    # the input arena (loading one is a single device-to-device copy)
    dgen = DeviceCriteo(gen, est.features, dev, seed=977 + rank)
    ring = [dgen.next_packed() for _ in range(max(args.ring, 1))]
  else:
    # any other schema (Taobao-shaped DIN / MMoE: sequences, tag lists): schema-driven host generator, a ring of
    # distinct batches made resident on the device; pre-conditioning and the steady-state pass cycle over the ring
    from easyrec_amd.input.synthetic import SyntheticBatches
    gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=20240607 + rank, mode=args.ids)
    host_batches = [gen.next_batch() for _ in range(4)]
    host_ring = [pack(b) for b in host_batches]
    ring = host_ring + [pack(gen.next_batch()) for _ in range(max(min(args.ring, 64), 4) - 4)]
    dgen = RingSource(ring)
  est.features.load(ring[0])
  torch.cuda.synchronize()
  ep = world > 1 or args.force_ep
  graph_note = 'eager launches'
  if not args.no_graph:
    try:
      est.capture(warmup=3)
      graph_note = 'hipGraph replay'
      if ep:
        graph_note = ('one hipGraph (collectives inside)' if getattr(est, '_whole_graph', None) is not None
                      else 'hipGraph segments + eager all-to-alls')
    except Exception as e:  # noqa: BLE001  (multi-GPU only: keep the run alive, say so in the output)
      if not ep:
        raise
      est._graphs = None
      graph_note = 'eager launches (graph capture failed: %s)' % str(e)[:80]
      torch.cuda.synchronize()

  def barrier():
    if world > 1:
      import torch.distributed as dist
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.precondition):  # untimed: bring the tables into their steady state (see --precondition)
    est.train_step(dgen.next_packed())
  for i in range(args.warmup):
    est.train_step(ring[i % len(ring)])
  barrier()
  t0 = time.perf_counter()
  for i in range(args.steps):
    est.train_step(ring[(args.warmup + i) % len(ring)])
  barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  losses = est.loss_values()
  assert all(np.isfinite(v) for v in losses.values()), losses

  if rank != 0:
    return
  ms_per_step = dt / args.steps * 1e3
  value = world * B * args.steps / dt
  rccl_world = None
  if world > 1 or (args.rccl and args.force_ep):
    import torch.distributed as dist
    rccl_world = dist.get_world_size()
  out = {
      'metric': baseline_metric(),
      'value': value,
      'unit': 'examples/s',
      'n_gpus': world,
      'steps': args.steps,
      'warmup': args.warmup,
      'ms_per_step': ms_per_step,
      'higher_is_better': True,
      'scaling': 'weak',
      'vs_baseline': None,
      'dtype': args.dense_dtype,
      'data': 'synthetic',
      'config': {
          'workload': workload_text(cfg, args, est, criteo, B, graph_note, len(ring)),
          'global_batch': world * B,
          'rccl_world_size': rccl_world,
          'parallelism': ('embedding-parallel x%d (row-sharded tables, RCCL all-to-all) + dense DP' % world if world > 1 else
                          'single GPU' if not ep else 'single GPU through the embedding-parallel code path (%s)' %
                          ('world-1 RCCL process group' if args.rccl else 'local copies for the collectives')),
      },
      'final_loss': losses.get('total_loss'),
      'device': kernels.hip().device_info(),
      'clocks': gpu_clocks(),
  }
  if world == 1 and not ep:
    lazy_bytes, sweep_bytes, per_kernel_bytes = embedding_bytes_per_step(est, host_ring) if criteo else (0.0, 0.0, {})
    if criteo:  # (the byte accounting of SURVEY.md 8d is written for the Criteo schema)
      out['embedding_stage'] = {
          'algorithmic_bytes_per_step': lazy_bytes + sweep_bytes,
          'lookup_update_bytes_per_step': lazy_bytes,
          'dense_decay_sweep_bytes_per_step': sweep_bytes,
          'whole_step_GBps': (lazy_bytes + sweep_bytes) / (ms_per_step * 1e-3) / 1e9,
          'note': 'whole_step_GBps divides the embedding stage\'s algorithmic bytes by the WHOLE step time; the stage '
                  'inside the replayed graph is roofline.embedding_stage; dense_decay_sweep_bytes_per_step is 0 unless '
                  '--dense_sweep (default: lazy dense decay)',
      }
    n_launch = max(10, min(args.steps, 50))
    pmc = None
    try:
      pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    except Exception:  # noqa: BLE001
      pmc = None
    # the committed counter passes were collected over the DEFAULT workload (DeepFM-Criteo, fp32): another config's
    # launches of a kernel of the same name move other bytes - its line carries traffic null
    if os.path.basename(args.config) != 'deepfm_criteo.config' or args.dense_dtype != 'f32' or args.batch_size:
      pmc = {k: v for k, v in (pmc or {}).items() if k != 'by_kernel'}
    if sweep_bytes > 0 and est.dense_sweep:
      dom = time_sweep_kernel(est, n_launch)
      ach = dom['bytes'] / (dom['avg_ms'] * 1e-3) / 1e9
      out['roofline'] = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': (pmc or {}).get('adam_decay_sweep_dim16_bytes_per_launch'),
                         'kernel': dom['kernel'], 'avg_kernel_ms': dom['avg_ms'],
                         'algorithmic_bytes_per_launch': dom['bytes'], 'launches_timed': dom['launches']}
    elif not args.no_graph:
      try:
        per_kernel, flops = kernel_breakdown(est, ring)
        emb_bytes = dict(per_kernel_bytes)
        if criteo:
          emb_bytes['stage'] = lazy_bytes
        out['roofline'] = step_roofline(est, per_kernel, flops, emb_bytes, pmc)
        if 'tail' in out['roofline']:
          # the embedding stage's OWN launches: the same step re-captured with the tail as four launches (bit-identical
          # results, tests/test_deepfm_gpu.py::test_fused_step_variants_change_no_bit), after the timed region
          out['roofline']['tail_unfused'] = tail_unfused(est, ring, emb_bytes, per_kernel)
      except Exception as e:  # noqa: BLE001
        out['roofline_error'] = str(e)[:300]
      try:  # cross-check of the profiler's durations: the largest GEMM launch alone, HIP events on the launch stream
        dom = time_gemm_kernel(est, n_launch)
        ach = dom['flops'] / (dom['avg_ms'] * 1e-3) / 1e12
        peak = 2500.0 if est.ctx.dense_dtype == 'bf16' else MFMA_F32_PEAK_TFLOPS
        out.setdefault('roofline', {})['largest_gemm_launch'] = {
            'kernel': dom['kernel'], 'avg_kernel_ms': dom['avg_ms'], 'achieved': ach, 'unit': 'TFLOP/s', 'frac': ach / peak,
            'algorithmic_flops_per_launch': dom['flops'], 'launches_timed': dom['launches'],
            'timed_with': 'HIP events on the launch stream, one launch per event pair',
            'traffic': (pmc or {}).get('gemm_f32_nn_4096x256x624_bytes_per_launch')
            if dom['flops'] == 2.0 * 4096 * 256 * 624 and est.ctx.dense_dtype != 'bf16' else None}
      except Exception as e:  # noqa: BLE001
        out.setdefault('roofline', {})['largest_gemm_launch'] = {'error': str(e)[:200]}
    if args.steady_steps > 0 and not args.no_graph:
      try:
        out['steady_state'] = steady_state(est, dgen, args.steady_steps)
        # (beside the headline: `value` is K timed steps over the resident ring; this is the mean over steady_steps
        # further DISTINCT batches, host batch hand-over included)
        out['ms_per_step_steady_mean'] = out['steady_state'].get('ms_per_step_mean')
        out['value_steady'] = out['steady_state'].get('examples_per_s')
      except Exception as e:  # noqa: BLE001
        out['steady_state'] = {'error': str(e)[:300]}
    if args.no_cpu_baseline and args.parity_steps > 0 and args.parity_only:
      try:
        out['parity_full_size'], _ = parity_full_size(cfg, est, host_ring, host_batches, B, args.parity_steps)
      except Exception as e:  # noqa: BLE001
        out['parity_full_size'] = {'error': str(e)[:300]}
    if not args.no_cpu_baseline:
      orc = None
      torch.set_num_threads(min(os.cpu_count() or 1, 64))
      if args.parity_steps > 0:
        try:
          # the oracle continues from the device's state: its steps double as the CPU baseline's warm-up
          out['parity_full_size'], orc = parity_full_size(cfg, est, host_ring, host_batches, B, args.parity_steps)
        except Exception as e:  # noqa: BLE001
          out['parity_full_size'] = {'error': str(e)[:300]}
      try:
        # fresh state for the CPU run = the device state now (any state is as good for timing)
        out['cpu_baseline'] = cpu_baseline(cfg, None if orc is not None else est.state_dict(), host_batches, B,
                                           args.cpu_seconds, orc=orc)
      except Exception as e:  # noqa: BLE001
        out['cpu_baseline'] = {'value': None, 'unit': 'examples/s', 'cores': 0, 'kind': 'port',
                               'sample': 'failed: %s' % str(e)[:200]}
  if world == 1 and not ep and args.from_file and criteo:
    try:
      out['from_file'] = from_file_rate(cfg, est, args.from_file, B)
    except Exception as e:  # noqa: BLE001
      out['from_file'] = {'format': args.from_file, 'error': str(e)[:300]}
  if world == 1 and ep and not args.no_graph:
    # the embedding-parallel step at W = 1 (--force_ep): the same per-kernel / per-family table as the single-GPU line,
    # collectives as a family of their own (rank 0 alone may run extra steps only at world 1)
    try:
      per_kernel, flops = kernel_breakdown(est, ring)
      out['roofline'] = step_roofline(est, per_kernel, flops, {}, None)
    except Exception as e:  # noqa: BLE001
      out['roofline_error'] = str(e)[:300]
  if 'roofline' not in out and not getattr(est, 'dense_sweep', False):
    # embedding-parallel runs (N > 1, --force_ep): the dense part is the same per rank; time the same GEMM on rank 0
    try:
      dom = time_gemm_kernel(est, max(10, min(args.steps, 50)))
      ach = dom['flops'] / (dom['avg_ms'] * 1e-3) / 1e12
      out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / MFMA_F32_PEAK_TFLOPS, 'traffic': None, 'kernel': dom['kernel'],
                         'avg_kernel_ms': dom['avg_ms'], 'algorithmic_flops_per_launch': dom['flops'],
                         'launches_timed': dom['launches'], 'note': 'rank 0'}
    except Exception as e:  # noqa: BLE001
      out['roofline_error'] = str(e)[:200]
  print(json.dumps(out))


if __name__ == '__main__':
  main()

"""Multi-GPU training step: row-sharded embedding tables + data-parallel dense variables.

Mirror of the reference's `train_distribute: EmbeddingParallelStrategy` path
(protos/train.proto:26-27, docs/source/train.md:191-227): one process per GPU; each rank reads its
own B examples (weak scaling); embedding tables are sharded by `id % world`
(layers/sharded_embedding.py); dense variables are replicated, their gradients all-reduced and
AVERAGED (compat/optimizers.py:328-331); embedding gradients are divided by the world size
(:315-316); BatchNorm statistics stay per-rank (the reference does not sync them); dense variables
start identical on every rank (same seed; the reference broadcasts rank 0's, utils/hvd_utils.py:43-56).

Per step (fixed-capacity exchange, the default: layers/sharded_embedding.py), per route (dim groups with the same
routed keys share one): an all-to-all of [count, keys] per owner, one of the rows and one of the row gradients
of all the route's dim groups side by side - all with equal, build-time split sizes, so there is NO host
synchronisation in the step; + ONE all-reduce (dense gradients, with the replicated small tables' gradients
behind them in the same buffer).  The static device work between the collectives (route | owner merge + serve |
lookup+forward+backward | local reduce | owner update + replicated apply + dense optimizer) is captured as five
hipGraphs; the collectives are issued eagerly between them and the host runs ahead of the device.
Overlap (round 5, more than one rank, no gradient clipping): everything the all-reduce carries - the dense gradients
and, behind them in the same buffer, the replicated small tables' row sums - is final once the model's backward has
returned and the replicated half of the local reduction has run, long before the sharded tables' gradients are
de-duplicated and exchanged.  The all-reduce is issued there, asynchronously on a second communicator (core/comm.py
all_reduce_sum_async), runs under the sharded half of the local reduction and next to the gradient all-to-all, and is
joined right before the optimizer kernels.  (With clipping by global norm the buffer's tail also carries this rank's share
of the embedding gradients' norm, which only the sharded reduction produces: the step then keeps the serial order.)
The compact exchange (variable split sizes through one host sync per step, three segments) remains for lookups
the per-lookup routed sort does not cover.
"""
import os

import torch

from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.core.comm import LocalComm, TorchDistComm
from easyrec_amd.layers.sharded_embedding import ShardedEmbeddingEngine
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator


class EmbeddingParallelEstimator(EasyRecEstimator):

  def __init__(self, pipeline_config, device='cuda', batch_size=None, seed=0, rank=0, world=1, comm=None,
               schema_kwargs=None, is_training=True, replicate_bytes=256 * 1024, recv_slack=2.0, dense_dtype=None,
               dense_sweep=None):
    if comm is None:
      comm = TorchDistComm() if world > 1 else LocalComm()
    assert comm.rank == rank and comm.world == world, 'comm (%d/%d) does not match rank/world (%d/%d)' % (
        comm.rank, comm.world, rank, world)
    self.comm = comm
    self.rank, self.world = rank, world
    self._engine_kwargs = dict(replicate_bytes=replicate_bytes, recv_slack=recv_slack)
    self._graphs = None
    self._whole_graph = None
    self._dense_work = None
    self._overlap = os.environ.get('EASYREC_AMD_EP_OVERLAP', 'auto')  # 'auto': more than one rank; '1' / '0': A/B switch
    super(EmbeddingParallelEstimator, self).__init__(pipeline_config, device=device, batch_size=batch_size, seed=seed,
                                                     schema_kwargs=schema_kwargs, is_training=is_training,
                                                     overlap_sweep=False, dense_dtype=dense_dtype,
                                                     dense_sweep=dense_sweep)
    # embedding gradients are divided by the world size (compat/optimizers.py:315-316)
    self.emb_grad_scale = self.emb_grad_scale / float(world)

  def _make_engine(self):
    return ShardedEmbeddingEngine(self.device, self.batch_size, self.comm, seed=self.seed, **self._engine_kwargs)

  def _dense_grad_scale(self):
    return 1.0 / float(self.world)  # hvd.allreduce(op=Average)

  # the replicated small tables' dense gradient buffer lives behind the dense variables' gradients: zeroed by the
  # same fill and summed over the ranks by the same all-reduce
  # (+ with gradient clipping, 4 floats whose first holds this rank's share of the embedding gradients' squared norm)
  def _extra_grad_floats(self):
    return (self.engine.rep_flat.numel() if self.engine.rep else 0) + (4 if self.clip_norm > 0 else 0)

  def _after_pack(self):
    n_rep = self.engine.rep_flat.numel() if self.engine.rep else 0
    if self.engine.rep:
      self.engine.set_rep_flat(self.varstore.grad_tail[:n_rep])
    if self.clip_norm > 0:
      self._norm_slot = self.varstore.grad_tail[n_rep:n_rep + 1]

  def _clip_from_reduced(self):
    """After the all-reduce: norm^2 = dense part (identical on every rank) + the summed per-rank embedding shares
    (compat/optimizers.py:453-481: hvd.grouped_allreduce of the sharded tables' l2 sums) -> the step's multiplier."""
    be, vs = kernels.hip(), self.varstore
    be.gradsq_dense(vs.flat, vs.flat_grad, vs.l2coef if vs.any_l2 else None, self.hyper[1], self._normsq, accumulate=False)
    be.reduce_sum(self._norm_slot, 1.0, self._normsq, accumulate=True)
    be.clip_scale(self._normsq, self.clip_norm, self.hyper, self.grad_norm)

  def _sync_dense_grads(self):
    if self.world > 1 or not isinstance(self.comm, LocalComm):
      self.comm.all_reduce_sum(self.varstore.flat_grad_all)

  # -- overlap: the dense all-reduce in flight under the local reduction and the gradient all-to-all
  @property
  def overlap(self):
    """At world 1 there is nothing to overlap and the split costs a graph segment and the second communicator's stream
    hand-over (`bench.py --force_ep --rccl`: 0.686 against 0.601 ms, profiles/r05_ep_world1_rccl_lines.txt): the switch
    EASYREC_AMD_EP_OVERLAP=1 forces it on there (the tests do), =0 off everywhere."""
    if self._overlap == '0' or self.clip_norm > 0 or not hasattr(self.comm, 'all_reduce_sum_async'):
      return False
    return self.world > 1 or (self._overlap == '1' and not isinstance(self.comm, LocalComm))

  def _start_dense_allreduce(self):
    """right after the model's backward and the replicated tables' reduction: every gradient the buffer carries is final
    (compat/optimizers.py:315-331 issues the dense ones one by one as backward produces them)"""
    self._dense_work = self.comm.all_reduce_sum_async(self.varstore.flat_grad_all)

  def _finish_exchanges(self):
    self.engine.exchange_row_grads()
    self.comm.wait(self._dense_work)  # joined here: the optimizer kernels of the next segment read the summed gradients
    self._dense_work = None

  # -- the step in phases: static device work (capturable) around the data-dependent exchanges
  def _phase_prologue(self):
    hash_job = self.features.hash_job() if self.fused_front else None
    kernels.hip().step_prologue(self.hyper_table, self.step_counter, self.hyper, history=self.lr_hist,
                                zero=self.varstore.flat_grad_all, decay_tables=self.decay_tables, hash_job=hash_job)
    self.features.transform(hashed=hash_job is not None)

  def _phase_route(self):
    self._phase_prologue()
    self.engine.route()

  # hash-table (ev_params) tables: the step's ids visit their owners before the route (two more all-to-alls)
  def _phase_kv_bucket(self):
    self._phase_prologue()
    self.engine.kv_bucket()

  def _phase_kv_owner(self):
    self.engine.kv_owner_translate()

  def _phase_kv_route(self):
    self.engine.kv_unbucket()
    self.engine.route()

  def _phase_compute(self, reduce=True):
    """lookup -> forward -> losses -> backward -> local gradient reduction (no collective inside).  reduce=False: the step
    overlaps the dense all-reduce with the SHARDED half of the local reduction, which is then a segment of its own
    (_phase_reduce_sharded); this one ends with the replicated half."""
    if self.is_training and not self.engine.inference:
      # the owned rows' rolling flush next to the lookup and the dense part (the owners' catch-up ran in an earlier
      # phase, their row update comes in a later one); fork and join inside this phase: one hipGraph holds both
      self.engine._start_window_flush()
    self.engine.lookup()
    self.engine._ran_version = self.features.version  # the model's input-layer calls find the lookup done
    with context.use(self.ctx):
      self.model.begin_step()
      self.model.build_predict_graph()
      loss_dict = self.model.build_loss_graph()
      self._loss_tail(loss_dict)
      if self.is_training:
        self.model.backward()
        self.engine.reduce_local_replicated()
        if reduce:
          self._phase_reduce_sharded()
    self.engine._join_window_flush()

  def _phase_forward_backward(self):
    self._phase_compute(reduce=False)

  # -- round 6: the requester's tail merged (er_emb_reduce_local_tail): weight gradients + every local reduction + the loss
  # tail as two launches at the end of the compute segment; the dense all-reduce (asynchronous where the communicator has a
  # second one) and the gradient all-to-all are issued back to back behind it and overlap EACH OTHER
  @property
  def merged_reduce(self):
    be = kernels.hip()
    return bool(getattr(be, 'ep_merged_reduce', False)) and self.engine.padded and self.is_training

  def _phase_compute_merged(self):
    if not self.engine.inference:
      self.engine._start_window_flush()
    self.engine.lookup()
    self.engine._ran_version = self.features.version
    be = kernels.hip()
    with context.use(self.ctx):
      self.model.begin_step()
      self.model.build_predict_graph()
      loss_dict = self.model.build_loss_graph()
      riders = bool(getattr(be, 'tail_riders', False)) and bool(getattr(be, 'fused_tail', False))
      self._loss_tail(loss_dict, defer=riders)
      self.model.backward(flush=False)
      self.engine.reduce_local_tail(pending_wgrads=True)
      if riders:
        be.flush_loss_tail()  # (a tail that did not take it)
      if self.clip_norm > 0:
        self.engine.local_gradsq(self._norm_slot, self._emb_gradsq_weight())
    self.engine._join_window_flush()

  def _collectives_after_compute(self):
    # (at world 1 there is nothing to overlap and the second communicator's stream hand-over only costs: the serial order,
    # unless EASYREC_AMD_EP_OVERLAP=1 forces the asynchronous form - the collective-order test does)
    if self.clip_norm <= 0 and hasattr(self.comm, 'all_reduce_sum_async') and \
        ((self._overlap == 'auto' and self.world > 1) or (self._overlap == '1' and not isinstance(self.comm, LocalComm))):
      self._start_dense_allreduce()
      self._finish_exchanges()
    else:
      self._sync_dense_grads()
      self.engine.exchange_grads()

  def _phase_reduce_sharded(self):
    """the gradients of the sharded tables' rows, de-duplicated per (owner, id) for the exchange (+ with clipping this
    rank's share of the norm)"""
    self.engine.reduce_local_sharded()
    if self.clip_norm > 0:
      self.engine.local_gradsq(self._norm_slot, self._emb_gradsq_weight())

  def _phase_apply(self):
    vs = self.varstore
    self.engine.apply_replicated(self.opt_emb.kind, self.hyper[0])
    kernels.hip().dense_opt_step(vs.flat, vs.slots.get('m'), vs.slots.get('v'), vs.flat_grad,
                                 vs.l2coef if vs.any_l2 else None, self.opt_dense.kind, self.hyper[1],
                                 l2_partials=vs.l2_partials)

  def _phase_owner_serve(self):
    self.engine.owner_serve()

  def _phase_update(self):
    if self.clip_norm > 0:
      self._clip_from_reduced()
    be, eng, vs = kernels.hip(), self.engine, self.varstore
    owners = [sh['owner'] for sh in eng.shard.values()]
    if getattr(be, 'ep_update_tail', False) and 1 <= len(owners) <= 4 and len(eng.rep) <= 4:
      # the owners' row update; its second launch also carries the replicated tables' apply and the dense optimizer
      # (er_emb_owner_update_tail: same bodies, bit-identical to owner_update + _phase_apply)
      tables = [(r['st']['var'], r['st']['m'], r['st']['v'], r['dense']) for r in eng.rep.values()]
      be.emb_owner_update_tail(owners, self.opt_emb.kind, self.hyper[0], tables,
                               (vs.flat, vs.slots.get('m'), vs.slots.get('v'), vs.flat_grad,
                                vs.l2coef if vs.any_l2 else None, self.opt_dense.kind, self.hyper[1], vs.l2_partials))
      eng._roll_flush(self.hyper[0])
      return
    eng.owner_update(self.opt_emb.kind, self.hyper[0])
    self._phase_apply()

  def _compact_exchange_and_update(self):
    self._sync_dense_grads()
    if self.clip_norm > 0:
      self._clip_from_reduced()
    self.engine.exchange_grads_and_update(self.opt_emb.kind, self.hyper[0])

  def _phases(self):
    """[(static device work, the collectives that follow it)]: the static parts replay as hipGraphs."""
    eng = self.engine
    after_route = eng.exchange_keys if eng.padded else eng.exchange
    if eng.kv_jobs:
      head = [(self._phase_kv_bucket, eng.kv_exchange_ids), (self._phase_kv_owner, eng.kv_exchange_rows),
              (self._phase_kv_route, after_route)]
    else:
      head = [(self._phase_route, after_route)]
    if eng.padded:
      # fixed-capacity exchange: no host-side sizes anywhere, the host never waits for the device
      seq = head + [(self._phase_owner_serve, eng.exchange_rows)]
      if self.merged_reduce:
        seq += [(self._phase_compute_merged, self._collectives_after_compute), (self._phase_update, None)]
      elif self.is_training and self.overlap:
        seq += [(self._phase_forward_backward, self._start_dense_allreduce), (self._phase_reduce_sharded, self._finish_exchanges),
                (self._phase_update, None)]
      elif self.is_training:
        seq += [(self._phase_compute, lambda: (self._sync_dense_grads(), eng.exchange_grads())), (self._phase_update, None)]
      else:
        seq += [(self._phase_compute, None)]
      return seq
    seq = list(head)  # host sync (split sizes) + all-to-all keys / rows
    if self.is_training:
      seq += [(self._phase_compute, self._compact_exchange_and_update), (self._phase_apply, None)]
    else:
      seq += [(self._phase_compute, None)]
    return seq

  def _device_step(self):
    g = self._graphs
    if g is not None and self._whole_graph is not None:
      self._whole_graph.replay()  # (static work AND collectives: capture(whole=True))
      return
    for i, (static, collectives) in enumerate(self._phases()):
      if g is None:
        static()
      else:
        g[i].replay()
      if collectives is not None:
        collectives()

  OVERFLOW_CHECK_EVERY = 256  # steps between blocking reads of the exchange's sticky overflow flag

  def train_step(self, batch=None):
    assert self._built, 'call build() first'
    if batch is not None:
      self.features.load(batch)
    else:
      self.features.version += 1
    self._refresh_hyper()
    self._device_step()
    self.global_step += 1
    # An owner that receives more keys than the exchange's capacity voids the step (keys are clamped): the flag is
    # sticky, so a periodic blocking read bounds how long a skewed id distribution can go unnoticed; evaluate(),
    # state_dict(), checkpoint.save() and loss_values() check it too.
    if self.global_step % self.OVERFLOW_CHECK_EVERY == 0:
      self.engine.check_overflow()
    return self.losses

  def evaluate(self, batches, eval_config=None):
    self.engine.check_overflow()
    return super(EmbeddingParallelEstimator, self).evaluate(batches, eval_config)

  def state_dict(self, slots=False):
    self.engine.check_overflow()
    return super(EmbeddingParallelEstimator, self).state_dict(slots=slots)

  def whole_graph_default(self):
    """May capture() put the collectives INSIDE the graph?  Only the fixed-capacity exchange (nothing in it waits for the
    host), and by default only where it has been run on hardware: one rank (local copies, or a world-1 RCCL process group:
    `bench.py --force_ep --rccl`, where every eager collective between two graph segments costs 25-35 us of idle device);
    EASYREC_AMD_EP_WHOLE_GRAPH=1 asks for it on more ranks (RCCL collectives are capturable: the kernels' plans become
    persistent), =0 switches it off."""
    sw = os.environ.get('EASYREC_AMD_EP_WHOLE_GRAPH', 'auto')
    if sw == '0' or not self.engine.padded or torch.device(self.device).type != 'cuda':
      return False
    return sw == '1' or self.world == 1

  def capture(self, warmup=3, whole=None):
    """Capture the step as hipGraphs: the static segments between the collectives (which then stay eager), or -
    whole=True, default whole_graph_default() - ONE graph of static work and collectives.  A whole-step capture that
    fails falls back to the segments.  Runs `warmup` eager steps on the loaded batch first."""
    assert self._built and self._graphs is None and self._whole_graph is None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
      for _ in range(warmup):
        self.train_step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    if whole is None:
      whole = self.whole_graph_default()
    if whole:
      try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
          for phase, collectives in self._phases():
            phase()
            if collectives is not None:
              collectives()
        self._whole_graph = g
        self._graphs = [g]
        return self._graphs
      except Exception as e:  # noqa: BLE001  (a communicator that cannot be captured: the segments below)
        import logging
        logging.warning('easyrec_amd: whole-step hipGraph capture failed (%s); capturing the static segments', str(e)[:200])
        self._whole_graph = None
        self._dense_work = None
        torch.cuda.synchronize()
    pool = torch.cuda.graph_pool_handle()
    graphs = []
    for phase, _ in self._phases():
      g = torch.cuda.CUDAGraph()
      # thread_local: the RCCL watchdog thread of the process group may touch its own events meanwhile
      with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
        phase()
      graphs.append(g)
    # captures do not execute: device state (tables, step counter) is exactly as after the warm-up steps
    self._graphs = graphs
    return graphs

  def loss_values(self, average=False):
    vals = super(EmbeddingParallelEstimator, self).loss_values()
    self.engine.check_overflow()  # (the losses were just read back: the device is idle anyway)
    if average and self.world > 1:
      keys = sorted(vals)
      t = torch.tensor([vals[k] for k in keys], dtype=torch.float64, device=self.device)
      self.comm.all_reduce_sum(t)
      vals = {k: float(v) / self.world for k, v in zip(keys, t.tolist())}
    return vals

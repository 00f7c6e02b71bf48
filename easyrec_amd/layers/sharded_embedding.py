"""Embedding-parallel input layer engine: tables row-sharded over the GPUs of one node.

Mirror of the reference's EmbeddingParallelStrategy path
(compat/feature_column/feature_column.py:248-357 `embedding_parallel_lookup`, :416-625
`_get_logits_embedding_parallel`; compat/optimizers.py:285-345):
  owner of id = id % world, local row = id // world, shard rows = ceil(rows / world)   (:296,317,461-463)
  forward : unique ids -> all-to-all ids -> owner gathers rows -> all-to-all rows -> combine
  backward: row gradients -> all-to-all to the owners -> owners reduce + apply the optimizer with the
            gradient divided by the world size (optimizers.py:315-316)
MI355X design: the de-duplication, the grouping by owner and the backward's segmented
reduction all come out of ONE per-lookup LDS sort per route and step (`er_emb_route`: key = owner * stride + local
row); dim groups that read the same ids share the route.  The exchange has a fixed capacity per peer, so its
all-to-alls (keys with their counts in front, rows, row gradients - the route's dim groups side by side) have
equal build-time splits: no host synchronisation, static launches that replay as hipGraphs.  The lookup itself stays
the single fused `er_emb_fwd` launch, reading the received rows; the owner MERGES the sorted runs it receives
(`er_emb_owner_merge_padded`), catches the rows up and replies in one launch (`er_emb_owner_serve`), and applies
the same in-order reduce + row-wise optimizer as single-GPU training (`er_emb_bwd_update_multi`).

Small tables (<= `replicate_bytes`) are replicated and trained data-parallel: their per-row gradient sums go
straight into a dense buffer behind the dense variables' gradients (`er_emb_bwd_reduce_dense`), are summed by the
same all-reduce, and applied on every rank identically in one pass (`er_emb_dense_apply`) - the same math as
sharding them (SURVEY.md 8e allows it).
"""
import os
from collections import OrderedDict

import torch

from easyrec_amd import kernels
from easyrec_amd.layers.input_layer import EmbeddingEngine


class ShardedEmbeddingEngine(EmbeddingEngine):

  def __init__(self, device, batch_size, comm, seed=0, replicate_bytes=256 * 1024, recv_slack=2.0):
    super(ShardedEmbeddingEngine, self).__init__(device, batch_size, seed=seed)
    self.comm = comm
    self.rank, self.world = comm.rank, comm.world
    self.replicate_bytes = replicate_bytes
    self.share_route = os.environ.get('EASYREC_AMD_SHARE_ROUTE', '1') != '0'  # A/B switch
    # the owner merges the runs it receives instead of sorting them: 10 us against 52 us at world 1, 24 us at world 8
    # (45 k keys); one binary search per run and key, so beyond 16 runs the radix sort wins again (117 us at 64)
    self.owner_merge = os.environ.get('EASYREC_AMD_OWNER_MERGE', '1') != '0' and self.world <= 16
    self.padded_exchange = os.environ.get('EASYREC_AMD_PADDED_EXCHANGE', '1') != '0'  # A/B switch
    self.padded = False  # set by finalize(): every sharded dim group uses the fixed-capacity exchange
    self.recv_slack = recv_slack
    self.shard = OrderedDict()  # dim -> dict of the sharded half of the dim group
    self.rep = OrderedDict()    # dim -> dict of the replicated half
    self.placement = {}         # table name -> ('shard' | 'rep', dim, local_base, local_rows)

  def _is_replicated(self, t):
    return t['rows'] * t['dim'] * 4 <= self.replicate_bytes

  # -- storage
  def finalize(self, opt_kind):
    assert not self.finalized
    be = kernels.hip()
    W, rank, dev = self.world, self.rank, self.device
    # 1. placement of every table
    shard_rows, rep_rows = OrderedDict(), OrderedDict()
    for name, t in self.tables.items():
      dim = t['dim']
      if t.get('kv'):
        # a hash-table table: rank r owns the map and the arena of the ids with id % W == r (the reference: SOK's
        # DynamicVariable, compat/feature_column/feature_column.py:470-503); its lookups carry the virtual dense ids
        # arena_row * W + owner made by translate_kv_ids(), so from here on it is a sharded table of n_local * W rows
        n_local = (t['rows'] + W - 1) // W
        t['rows'] = n_local * W
        base = shard_rows.get(dim, 0)
        shard_rows[dim] = base + n_local
        self.placement[name] = ('shard', dim, base, n_local)
      elif self._is_replicated(t):
        base = rep_rows.get(dim, 0)
        rep_rows[dim] = base + t['rows']
        self.placement[name] = ('rep', dim, base, t['rows'])
      else:
        n_local = (t['rows'] + W - 1) // W
        base = shard_rows.get(dim, 0)
        shard_rows[dim] = base + n_local
        self.placement[name] = ('shard', dim, base, n_local)
    # 2. storage + initial values (the same full table on every rank, then the rank's rows)
    for dim, total in shard_rows.items():
      self.shard[dim] = {'st': self._alloc_storage(total, dim, opt_kind), 'stride': total}
    for dim, total in rep_rows.items():
      # replicated (small) tables keep the streaming sweep of TF-exact Adam: a few KB per step
      self.rep[dim] = {'st': self._alloc_storage(total, dim, opt_kind, force_bitmap=True)}
    for name, t in self.tables.items():
      kind, dim, base, n_local = self.placement[name]
      if t.get('kv'):
        arena = self.shard[dim]['st']['var'][base:base + n_local]
        arena.zero_()  # rows are created on first sight (er_kv_translate: a pure function of seed, id and column)
        mean, std = self._init_mean_std(t)
        assert t['kv_steps_to_live'] == 0 or self._clock is not None, 'steps_to_live needs the step clock (set_step_clock)'
        from easyrec_amd.layers.input_layer import _stable_seed
        self.kv_tables[name] = be.kv_create(arena, n_local, _stable_seed(name, self.seed), mean, std,
                                            filter_freq=t['kv_filter_freq'], steps_to_live=t['kv_steps_to_live'],
                                            step=self._clock[0] if t['kv_steps_to_live'] > 0 else None)
      elif kind == 'rep':
        self.init_table_values(name, self.rep[dim]['st']['var'][base:base + n_local])
      else:
        full = torch.empty(t['rows'], dim, dtype=torch.float32, device=dev)
        self.init_table_values(name, full)
        mine = full[rank::W]
        var = self.shard[dim]['st']['var']
        var[base:base + n_local].zero_()
        var[base:base + mine.shape[0]].copy_(mine)
        del full
    self.storage = {}  # the base class' per-dim storage view: shard halves (bench / reporting)
    for dim, sh in self.shard.items():
      self.storage[dim] = sh['st']
    # 3. lookups, split per dim into the sharded and the replicated half (entry order = spec order)
    fwd_specs, reg_count = [], 0
    per_dim = OrderedDict()
    for gkey, g in self._ordered_groups():
      g['specs'] = []
      for (tname, ids, offsets, weights, col, combiner, n_rows, max_nnz, name) in g['pending']:
        t = self.tables[tname]
        kind, dim, base, n_local = self.placement[tname]
        per_dim.setdefault(dim, {'shard': [], 'rep': []})[kind].append(
            dict(tname=tname, ids=ids, offsets=offsets, weights=weights, col=col, combiner=kernels.COMBINERS[combiner],
                 n_rows=n_rows, max_nnz=max_nnz, name=name, group=g, t=t, base=base, slot=len(fwd_specs)))
        fwd_specs.append(None)
        if g['reg'] > 0:
          reg_count += 1
    self.padded = bool(self.shard) and self.padded_exchange and self.owner_merge
    # routes: dim groups whose sharded lookups read the same ids against tables of the same geometry (DeepFM's wide
    # dim-1 and deep dim-16 tables of one feature set) have the SAME routed keys: the later one follows the first
    # one's route - one sort, one de-duplication, one key all-to-all and one owner-side merge serve both - and, in
    # the fixed-capacity exchange, their rows (and row gradients) lie side by side in ONE buffer, [slots, 16 | 1 | pad]:
    # one all-to-all of rows and one of gradients per ROUTE instead of per dim group.
    classes = OrderedDict()
    for dim, halves in per_dim.items():
      if halves['shard']:
        sig = self._route_sig(self.shard[dim]['stride'], halves['shard'])
        classes.setdefault(sig if self.share_route else (sig, dim), []).append(dim)
    self._route_plan = {}
    for dims in classes.values():
      col, cols = 0, []
      for d in dims:
        cols.append(col)
        col += (d + 3) // 4 * 4  # 16-byte aligned column blocks
      for d, c in zip(dims, cols):
        self._route_plan[d] = (dims[0], c, col) if self.padded else (dims[0], 0, d)
    for dim, halves in per_dim.items():
      if halves['shard']:
        self._build_shard_half(dim, halves['shard'], fwd_specs, opt_kind)
      if halves['rep']:
        self._build_rep_half(dim, halves['rep'], fwd_specs, opt_kind)
    assert all(s is not None for s in fwd_specs)
    self.fwd_specs = fwd_specs
    self.n_reg_specs = reg_count
    if fwd_specs:
      self.plan = be.emb_plan_create(fwd_specs)
      self.sumsq = torch.zeros(max(self.plan['num_blocks'], 1), dtype=torch.float32, device=dev)
      from easyrec_amd.layers.input_layer import _blocks_of
      self.reg_blocks = _blocks_of(fwd_specs[:reg_count])
    self.reg_lambda = max([g['reg'] for g in self.groups.values()] + [0.0])
    # one flat buffer behind the replicated halves' dense gradients: ONE all-reduce per step
    if self.rep:
      sizes = [(dim, r['st']['total_rows'] * (dim + 1)) for dim, r in self.rep.items()]
      self._rep_sizes = sizes
      self.set_rep_flat(torch.zeros(sum(n for _, n in sizes), dtype=torch.float32, device=dev), own=True)
    self.counts_dev = torch.zeros(max(len(self.shard), 1), W, dtype=torch.int32, device=dev)
    self.lazy_decay = self.lazy_decay and opt_kind == kernels.OPT_ADAM
    self.finalized = True

  @staticmethod
  def _route_sig(stride, lookups):
    def ptr(t):
      return None if t is None else t.data_ptr()

    return (stride, tuple((ptr(lk['ids']), ptr(lk['offsets']), ptr(lk['weights']), lk['t']['rows'], lk['base'],
                           lk['n_rows'], lk['max_nnz'], lk['combiner']) for lk in lookups))

  def set_rep_flat(self, buf, own=False):
    """The dense gradient buffer of the replicated tables.  own=False: it is the tail of the dense variables' flat
    gradient buffer - zeroed and all-reduced together with it (one collective instead of two)."""
    assert buf.numel() == sum(n for _, n in self._rep_sizes) and buf.is_contiguous()
    self.rep_flat, self.rep_flat_own = buf, own
    off = 0
    for dim, n in self._rep_sizes:
      r = self.rep[dim]
      r['dense'] = buf[off:off + n].view(r['st']['total_rows'], dim + 1)
      off += n

  def _build_shard_half(self, dim, lookups, fwd_specs, opt_kind):
    be = kernels.hip()
    dev, W = self.device, self.world
    sh = self.shard[dim]
    caps = [(lk['max_nnz'] if lk['offsets'] is not None else lk['n_rows']) for lk in lookups]
    n_ent = sum(caps)
    m_cap = max(int(self.recv_slack * n_ent), 1024)

    lead_dim, col, ld = self._route_plan[dim]
    lead_dim = None if lead_dim == dim else lead_dim
    sh['leader'] = lead_dim
    lead = self.shard[lead_dim] if lead_dim is not None else None
    # Fixed-capacity ("padded") exchange: owner w's keys sit at [w * C, w * C + count[w]) of every buffer, so the
    # all-to-alls have equal splits known at build time: no count all-gather, no host synchronisation, every launch
    # static (hipGraph segments; the host runs ahead of the device).  C = slack x the even share, at most all entries.
    # It needs the owner-side merge (world <= 16); otherwise the compact exchange with host-side split sizes is used.
    padded = self.padded
    peer_cap = min(n_ent, -(-int(self.recv_slack * n_ent) // W)) if padded else 0
    n_slots = W * peer_cap if padded else n_ent  # rows of the requester-side buffers
    if padded:
      m_cap = W * peer_cap
    sh.update(n_entries=n_ent, m_cap=m_cap, peer_cap=peer_cap)
    for k, n in (('recv_rows', n_slots), ('ugrads', n_slots), ('rows_out', m_cap), ('recv_grads', m_cap)):
      own = lead is None or not padded  # (the compact exchange moves every dim group's rows on their own)
      if own:  # [slots, ld]: this group's columns first, its followers' after (ld == dim without followers)
        sh[k + '_all'] = torch.zeros(n, ld, dtype=torch.float32, device=dev)
      sh[k] = (sh if own else lead)[k + '_all'][:, col:col + dim]
    if lead is not None:
      for k in ('ukeys', 'n_unique', 'uidx', 'recv_keys', 'recv_ids', 'recv_cnt'):
        sh[k] = lead[k]
    else:
      sh.update(
          ukeys=torch.zeros(n_slots + (W if padded else 0), dtype=torch.int32, device=dev),  # padded: [count, keys] x W
          n_unique=torch.zeros(1, dtype=torch.int32, device=dev),
          uidx=torch.full((n_ent,), -1, dtype=torch.int64, device=dev),
          recv_keys=torch.zeros(m_cap + (W if padded else 0), dtype=torch.int32, device=dev),
          recv_ids=torch.full((m_cap,), -1, dtype=torch.int64, device=dev),
          recv_cnt=torch.zeros(W, dtype=torch.int32, device=dev))
    req_specs, off = [], 0
    for lk, cap in zip(lookups, caps):
      g = lk['group']
      # requester group: the local lookups against their upstream-gradient buffers (routing + reduce)
      req_specs.append(kernels.LookupSpec(
          table=sh['recv_rows'], ids=lk['ids'], offsets=lk['offsets'], weights=lk['weights'], out=g['dout'],
          out_col=lk['col'], rows=lk['t']['rows'], key_base=0, dim=dim, combiner=lk['combiner'], n_rows=lk['n_rows'],
          max_nnz=lk['max_nnz'], name=lk['name']))
      # forward: the same lookup, reading the rows received from the owners through the entry -> unique map
      fwd_specs[lk['slot']] = kernels.LookupSpec(
          table=sh['recv_rows'], ids=sh['uidx'][off:off + cap], offsets=lk['offsets'], weights=lk['weights'],
          out=g['out'], out_col=lk['col'], rows=n_slots, key_base=0, dim=dim, combiner=lk['combiner'],
          n_rows=lk['n_rows'], max_nnz=lk['max_nnz'], name=lk['name'])
      g['specs'].append(fwd_specs[lk['slot']])
      off += cap
    span = max([lk['t']['rows'] for lk in lookups] + [n_slots, 1])  # validation bound only: keys are routed
    sh['req'] = be.emb_group_create(req_specs, dim, span, sh['recv_rows'], None, None, None)  # (updates no table)
    assert sh['req']['num_entries'] == n_ent
    be.emb_group_set_routing(sh['req'], W, sh['stride'], [lk['base'] for lk in lookups])
    if padded and lead is None:
      be.emb_group_set_peer_capacity(sh['req'], peer_cap, count_header=True)
    if lead is not None:
      assert be.emb_group_share_sort(sh['req'], lead['req']), 'dim %d: cannot follow the route of dim %d' % (dim, lead_dim)
    st = sh['st']
    owner_spec = kernels.LookupSpec(
        table=st['var'], ids=sh['recv_ids'], offsets=None, weights=None, out=sh['recv_grads'], out_col=0,
        rows=st['total_rows'], key_base=0, dim=dim, combiner=kernels.COMBINER_SUM, n_rows=m_cap, max_nnz=m_cap,
        name='owner_dim%d' % dim)
    sh['owner'] = be.emb_group_create([owner_spec], dim, st['total_rows'], st['var'], st['m'], st['v'], st['bitmap'])
    if lead is not None:  # the same received keys: the owner-side sort is shared too
      assert be.emb_group_share_sort(sh['owner'], lead['owner'])
    sh['lazy'] = None
    if self.lazy_decay and opt_kind == kernels.OPT_ADAM:
      # owner side of TF-exact Adam without the sweep: the received keys are sorted / de-duplicated once
      # (er_emb_route on the owner group), caught up before the rows are gathered, and the same sort serves
      # the owner-side update of the backward
      sh['lazy'] = self._enable_lazy_decay(dim, sh['owner'], st, m_cap)

  def _build_rep_half(self, dim, lookups, fwd_specs, opt_kind):
    be = kernels.hip()
    r = self.rep[dim]
    st = r['st']
    specs = []
    for lk in lookups:
      g = lk['group']
      view = st['var'][lk['base']:lk['base'] + lk['t']['rows']]
      spec = kernels.LookupSpec(
          table=view, ids=lk['ids'], offsets=lk['offsets'], weights=lk['weights'], out=g['out'], out_col=lk['col'],
          rows=lk['t']['rows'], key_base=lk['base'], dim=dim, combiner=lk['combiner'], n_rows=lk['n_rows'],
          max_nnz=lk['max_nnz'], name=lk['name'])
      fwd_specs[lk['slot']] = spec
      g['specs'].append(spec)
      specs.append(spec.with_out(g['dout']))
    r['group'] = be.emb_group_create(specs, dim, st['total_rows'], st['var'], None, None, None)
    for d0, r0 in self.rep.items():  # the same ids against tables of the same geometry: one sort per step for both
      if d0 != dim and 'group' in r0 and be.emb_group_share_sort(r['group'], r0['group']):
        break

  # -- per-step execution.  Three forward phases and three backward phases so that the static ones can be
  #    replayed as hipGraphs around the data-dependent exchanges (model/embedding_parallel.py):
  #      route()  [static]  -> exchange()  [host sync + all-to-alls] -> lookup() [static]
  #      reduce_local() [static] -> exchange_grads_and_update() [all-to-all, variable counts] -> apply_replicated() [static]
  def translate_kv_ids(self):
    """Hash-table lookups: this step's ids -> virtual dense ids (arena_row * world + owner).  The ids go to their
    owners (id % world) in a fixed-capacity all-to-all, the owners translate what they receive in the single-GPU
    engine's two launches - every occurrence of every rank counts towards filter_freq, unseen ids get a row while
    training and read zeros otherwise - and the arena rows come back the same way."""
    self.kv_bucket()
    self.kv_exchange_ids()
    self.kv_owner_translate()
    self.kv_exchange_rows()
    self.kv_unbucket()

  # (the same in pieces: static device work around the two all-to-alls, for the estimator's phases / hipGraph segments)
  def _kv_route(self):
    be = kernels.hip()
    if self._kv_handle is None:
      route = be.kv_route_create([tuple(job[1:]) for job in self.kv_jobs], self.world)
      # owner side: one translate job per (source rank, lookup) over the received block
      owner_jobs = []
      for src in range(self.world):
        for j, job in enumerate(self.kv_jobs):
          lo, hi = route['offs'][j], route['offs'][j + 1]
          if hi > lo:
            owner_jobs.append((self.kv_tables[job[0]], route['recv'][src, lo:hi], route['owner_rows'][src, lo:hi]))
      self._kv_handle = (route, be.kv_jobs_create(owner_jobs) if owner_jobs else None)
    return self._kv_handle

  def kv_bucket(self):
    kernels.hip().kv_bucket(self._kv_route()[0])

  def kv_exchange_ids(self):
    route = self._kv_route()[0]
    self.comm.all_to_all_equal(route['send'].view(-1), route['recv'].view(-1))

  def kv_owner_translate(self):
    owner = self._kv_route()[1]
    if owner is not None:
      kernels.hip().kv_translate_multi(owner, self.train_mode and not self.inference)

  def kv_exchange_rows(self):
    route = self._kv_route()[0]
    self.comm.all_to_all_equal(route['owner_rows'].view(-1), route['back'].view(-1))

  def kv_unbucket(self):
    kernels.hip().kv_unbucket(self._kv_route()[0])

  def route(self):
    be = kernels.hip()
    for gi, (dim, sh) in enumerate(self.shard.items()):
      if sh['leader'] is None:
        be.emb_route(sh['req'], sh['ukeys'], sh['n_unique'], sh['uidx'], self.counts_dev[gi])
      else:  # adopts the leader's sort and run heads (no launch: its entries were built with the leader's)
        be.emb_route(sh['req'], None, None, None, None)

  def exchange(self):
    if self.shard and self.padded:
      self.exchange_keys()
      self.owner_serve()
      self.exchange_rows()
    else:
      self._exchange_compact()

  # -- fixed-capacity exchange: collectives with build-time sizes around static device work
  def exchange_keys(self):
    comm = self.comm
    for gi, (dim, sh) in enumerate(self.shard.items()):
      if sh['leader'] is None:  # per owner [count, keys ...]: the counts ride with the keys
        comm.all_to_all_equal(sh['ukeys'], sh['recv_keys'])

  def owner_serve(self):
    """Received keys -> local rows, merged order, caught-up rows written to the reply buffers (static launches)."""
    be, W = kernels.hip(), self.world
    shs = list(self.shard.values())
    # (inference: the rows were flushed by begin_inference(); serving them must not replay anything)
    hyper = self._clock[2] if (any(sh['lazy'] is not None for sh in shs) and not self.inference) else None
    fused = getattr(be, 'ep_owner_fused', False)
    for sh in shs:
      if sh['leader'] is None:
        if fused:  # one launch; it also builds the lag-1 replay table the serve launch below reads
          be.emb_owner_ids_merge(sh['owner'], sh['recv_keys'], W, sh['peer_cap'], self.rank * sh['stride'], sh['recv_ids'],
                                 sh['recv_cnt'], hyper is not None and len(shs) <= 4)
        else:
          be.emb_owner_ids(sh['recv_keys'], None, W, sh['peer_cap'], self.rank * sh['stride'], sh['recv_ids'], sh['recv_cnt'])
          be.emb_owner_merge_padded(sh['owner'], sh['recv_cnt'], W, sh['peer_cap'])
    for i in range(0, len(shs), 4):
      be.emb_owner_serve([sh['owner'] for sh in shs[i:i + 4]], [sh['rows_out'] for sh in shs[i:i + 4]], hyper)

  def exchange_rows(self):
    for sh in self.shard.values():
      if sh['leader'] is None:  # the rows of every dim group of the route in one buffer
        self.comm.all_to_all_equal(sh['rows_out_all'], sh['recv_rows_all'])

  def exchange_grads(self):
    if self.rep and self.rep_flat_own:
      self.comm.all_reduce_sum(self.rep_flat)
    self.exchange_row_grads()

  def exchange_row_grads(self):
    for sh in self.shard.values():
      if sh['leader'] is None:
        self.comm.all_to_all_equal(sh['ugrads_all'], sh['recv_grads_all'])

  def owner_update(self, opt_kind, hyper):
    be = kernels.hip()
    owners = [sh['owner'] for sh in self.shard.values()]
    for i in range(0, len(owners), 4):  # the groups' reduce + optimizer kernels side by side in one launch
      be.emb_bwd_update_multi(owners[i:i + 4], opt_kind, hyper)
    self._roll_flush(hyper)

  def check_overflow(self):
    """Blocking: has any step so far routed more keys to one owner than the exchange's capacity (the flag is sticky)?"""
    be = kernels.hip()
    self.check_kv_overflow()
    for dim, sh in self.shard.items():
      if sh['peer_cap'] and sh['leader'] is None and be.emb_route_overflow(sh['req']):
        raise RuntimeError('embedding-parallel: rank %d routed more than %d keys of dim %d to one owner; raise recv_slack'
                           % (self.rank, sh['peer_cap'], dim))

  # -- compact exchange: split sizes through the host (one synchronisation per step)
  def _exchange_compact(self):
    be, comm = kernels.hip(), self.comm
    if not self.shard:
      return
    send, recv = comm.exchange_counts(self.counts_dev)  # host sync: split sizes
    # keys: one all-to-all per route (a follower group receives what its leader receives)
    for gi, (dim, sh) in enumerate(self.shard.items()):
      if sh['leader'] is not None:
        lead = self.shard[sh['leader']]
        sh['send_counts'], sh['recv_counts'], sh['m'] = lead['send_counts'], lead['recv_counts'], lead['m']
      else:
        sc, rc = send[gi], recv[gi]
        m = int(sum(rc))
        if m > sh['m_cap']:
          raise RuntimeError('embedding-parallel: rank %d receives %d keys for dim %d, capacity %d; raise recv_slack' %
                             (self.rank, m, dim, sh['m_cap']))
        sh['send_counts'], sh['recv_counts'], sh['m'] = sc, rc, m
        comm.all_to_all(sh['ukeys'], sc, sh['recv_keys'], rc)
        if m:
          torch.sub(sh['recv_keys'][:m], self.rank * sh['stride'], out=sh['recv_ids'][:m])
      be.emb_group_set_active(sh['owner'], sh['m'])
    # rows: per group (the tables differ)
    if self.owner_merge:
      # what a rank receives are `world` ascending duplicate-free runs: merged, not sorted (er_emb_owner_merge), and
      # every distinct row is caught up (lazy dense decay) and written to all its reply slots in one launch
      live = [sh for sh in self.shard.values() if sh['m']]
      for sh in live:
        if sh['leader'] is None:
          be.emb_owner_merge(sh['owner'], sh['recv_counts'])
      hyper = self._clock[2] if (any(sh['lazy'] is not None for sh in live) and not self.inference) else None
      for i in range(0, len(live), 4):
        be.emb_owner_serve([sh['owner'] for sh in live[i:i + 4]], [sh['rows_out'] for sh in live[i:i + 4]], hyper)
      for sh in self.shard.values():
        comm.all_to_all(sh['rows_out'], sh['recv_counts'], sh['recv_rows'], sh['send_counts'])
      return
    for dim, sh in self.shard.items():
      m, st = sh['m'], sh['st']
      lz = sh['lazy']
      if lz is not None and m and not self.inference:
        if sh['leader'] is None:
          be.emb_route(sh['owner'], lz['ukeys'], lz['n_unique'], None, None)
        else:
          be.emb_route(sh['owner'], None, None, None, None)
          lz = self.shard[sh['leader']]['lazy']
        be.emb_catch_up(sh['owner'], lz['ukeys'], lz['n_unique'], self._clock[2])
      be.gather_rows(st['var'], sh['recv_keys'], m, self.rank * sh['stride'], sh['rows_out'])
      comm.all_to_all(sh['rows_out'], sh['recv_counts'], sh['recv_rows'], sh['send_counts'])

  def lookup(self):
    for g in self.groups.values():
      g['got_grad'] = False
      g['terms'] = []
    if self.plan is not None:
      kernels.hip().emb_fwd(self.plan, self.sumsq if self.reg_lambda > 0 else None)

  def forward(self, version):
    if version == self._ran_version:
      return
    self._join_window_flush()  # (a forward that no row update followed)
    if self.kv_jobs:
      self.translate_kv_ids()
    self.route()
    self.exchange()
    self.lookup()
    self._ran_version = version

  def reduce_local(self):
    self.reduce_local_replicated()
    self.reduce_local_sharded()

  def reduce_local_tail(self, pending_wgrads=False):
    """reduce_local() with every reduction in ONE tile launch + ONE fix launch (er_emb_reduce_local_tail) and, with
    pending_wgrads (model.backward(flush=False)), the step's queued weight gradients contracted in the same grid, a
    deferred loss tail riding along - the single-GPU step's fused tail for the requester of the embedding-parallel step.
    Falls back to the launches apart when there are more than 4 table groups or the weight gradients do not fit."""
    be = kernels.hip()
    self.finish_group_grads()
    if self.rep and self.rep_flat_own:
      self.rep_flat.zero_()
    routed = [(sh['req'], sh['ugrads']) for sh in self.shard.values()]
    dense = [(r['group'], r['dense']) for r in self.rep.values()]
    wgrads = None
    if pending_wgrads:
      q, qb = be.take_wgrads()
      if qb:
        be.gemm_grouped(kernels.GEMM_TN, qb, bf16=True)
      if be.wgrads_fit_the_tail(q):
        wgrads = q
      elif q:
        be.gemm_grouped(kernels.GEMM_TN, q)
    if wgrads is None:
      be.flush_loss_tail()  # (a deferred loss tail rides with the weight gradients' grid only)
    if 1 <= len(routed) + len(dense) <= 4:
      be.emb_reduce_local_tail(routed, dense, wgrads=wgrads)
      return
    if wgrads:
      be.gemm_grouped(kernels.GEMM_TN, wgrads)
      be.flush_loss_tail()
    reps = list(self.rep.values())
    for i in range(0, len(reps), 4):
      be.emb_bwd_reduce_dense([r['group'] for r in reps[i:i + 4]], [r['dense'] for r in reps[i:i + 4]])
    self.reduce_local_sharded()

  def reduce_local_replicated(self):
    """first half of reduce_local: the group gradient buffers finished, the replicated tables' row sums in the dense buffer
    that is all-reduced with the dense gradients - after it everything the dense all-reduce carries is final"""
    be = kernels.hip()
    self.finish_group_grads()
    if self.rep:
      # replicated tables: the per-row gradient sums go straight into the dense buffer that is all-reduced
      if self.rep_flat_own:
        self.rep_flat.zero_()
      reps = list(self.rep.values())
      for i in range(0, len(reps), 4):
        be.emb_bwd_reduce_dense([r['group'] for r in reps[i:i + 4]], [r['dense'] for r in reps[i:i + 4]])

  def reduce_local_sharded(self):
    """second half: this rank's gradients of the sharded tables' rows, de-duplicated per (owner, id) for the exchange"""
    be = kernels.hip()
    for dim, sh in self.shard.items():
      be.emb_bwd_reduce_routed(sh['req'], sh['ugrads'])

  def local_gradsq(self, acc, weight):
    """Gradient clipping (compat/optimizers.py:453-481): acc[0] += weight * sum of squares of THIS rank's de-duplicated
    row gradients - the rows it is about to send to the owners (what arrives there is `values` of the owner's
    IndexedSlices: one row per (requester, distinct id), un-merged across requesters) and its share of the replicated
    tables' rows.  The all-reduce of the dense gradients carries the sum over the ranks (model/embedding_parallel.py).
    Call after reduce_local()."""
    be = kernels.hip()
    for gi, (dim, sh) in enumerate(self.shard.items()):
      if self.padded:
        if sh['leader'] is None:  # the route's dim groups side by side; pad columns stay zero
          be.gradsq_rows(sh['ugrads_all'], sh['ugrads_all'].shape[1], weight, acc, True, counts=self.counts_dev[gi],
                         seg_stride=sh['peer_cap'])
      else:
        be.gradsq_rows(sh['ugrads'], dim, weight, acc, True, counts=sh['n_unique'])
    for dim, r in self.rep.items():
      be.gradsq_rows(r['dense'], dim, weight, acc, True)  # [rows, dim | count]: zero where the rank had no gradient

  def exchange_grads_and_update(self, opt_kind, hyper):
    be, comm = kernels.hip(), self.comm
    if self.shard and self.padded:
      self.exchange_grads()
      self.owner_update(opt_kind, hyper)
      return
    if self.rep and self.rep_flat_own:
      comm.all_reduce_sum(self.rep_flat)
    for dim, sh in self.shard.items():
      comm.all_to_all(sh['ugrads'], sh['send_counts'], sh['recv_grads'], sh['recv_counts'])
    owners = [sh['owner'] for sh in self.shard.values()]
    for i in range(0, len(owners), 4):  # the groups' reduce + optimizer kernels side by side in one launch
      be.emb_bwd_update_multi(owners[i:i + 4], opt_kind, hyper)
    self._roll_flush(hyper)

  def apply_replicated(self, opt_kind, hyper):
    be = kernels.hip()
    tables = [(r['st']['var'], r['st']['m'], r['st']['v'], r['dense']) for r in self.rep.values()]
    for i in range(0, len(tables), 4):  # one pass over every replicated table: optimizer step or (TF-exact Adam) decay
      be.emb_dense_apply(tables[i:i + 4], opt_kind, hyper)

  def backward_update(self, opt_kind, hyper):
    self.reduce_local()
    self.exchange_grads_and_update(opt_kind, hyper)
    self.apply_replicated(opt_kind, hyper)
    self._decay_pending = True

  def flush_decay(self):
    if not self.lazy_decay:
      return
    be = kernels.hip()
    self._join_window_flush()
    for sh in self.shard.values():
      if sh['lazy'] is not None:
        be.emb_flush_decay(sh['owner'], self._clock[2])
    self._decay_pending = False

  def _lazy_states(self):
    return [sh['lazy'] for sh in self.shard.values() if sh.get('lazy')]

  def _lazy_groups(self):
    return [(sh['owner'], sh['lazy']) for sh in self.shard.values() if sh.get('lazy')]

  # -- host exchange (collective: every rank must call)
  def table_view(self, name):
    kind, dim, base, n_local = self.placement[name]
    st = self.rep[dim]['st'] if kind == 'rep' else self.shard[dim]['st']
    return st['var'][base:base + n_local]

  def slot_view(self, name, slot):
    kind, dim, base, n_local = self.placement[name]
    st = self.rep[dim]['st'] if kind == 'rep' else self.shard[dim]['st']
    return None if st[slot] is None else st[slot][base:base + n_local]

  def _gather_full(self, name, local):
    kind, dim, base, n_local = self.placement[name]
    if kind == 'rep':
      return local.detach().clone()
    rows = self.tables[name]['rows']
    parts = self.comm.all_gather_rows(local.contiguous())
    full = torch.empty(n_local * self.world, dim, dtype=local.dtype, device=local.device)
    for w, p in enumerate(parts):
      full[w::self.world] = p
    return full[:rows]

  def _kv_state(self, name, out, slots):
    """A hash-table table's entries of every rank, merged in key order (collective)."""
    import numpy as np
    be, kv, t = kernels.hip(), self.kv_tables[name], self.tables[name]
    dev = self.device
    if self._kv_filtered(name):
      seen, seen_rows, freq, version = be.kv_export_all(kv)
      has_row = seen_rows >= 0
      keys, rows = seen[has_row], seen_rows[has_row]
      gathered = [torch.cat(self.comm.all_gather_varlen(x.to(dev).contiguous())).cpu().numpy() for x in (seen, freq, version)]
      order = np.argsort(gathered[0], kind='stable')
      out[name + '/kv_seen_keys'] = gathered[0][order]
      out[name + '/kv_freq'] = np.minimum(gathered[1][order], max(t['kv_filter_freq'], 1)).astype(np.int32)
      out[name + '/kv_version'] = gathered[2][order]
    else:
      keys, rows = be.kv_export(kv)
    all_keys = torch.cat(self.comm.all_gather_varlen(keys.to(dev).contiguous())).cpu().numpy()
    order = np.argsort(all_keys, kind='stable')
    out[name + '/keys'] = all_keys[order]
    # (capacity: the whole table's, so that a single-process estimator - or the oracle - can hold the state)
    out[name + '/kv_meta'] = np.array([kv['seed'], kv['mean'], kv['stddev'], kv['capacity'] * self.world, t['kv_filter_freq'],
                                       t['kv_steps_to_live']], dtype=np.float64)
    views = [('', self.table_view(name))] + ([('/' + sl, self.slot_view(name, sl)) for sl in ('m', 'v')] if slots else [])
    for suffix, view in views:
      if view is not None:
        mine = view.detach()[rows.to(view.device)].contiguous()
        out[name + suffix] = torch.cat(self.comm.all_gather_varlen(mine)).cpu().numpy()[order]

  def state_dict(self, slots=False):
    out = OrderedDict()
    self.flush_decay()
    self.check_kv_overflow()
    for name in self.tables:
      if self.tables[name].get('kv'):
        self._kv_state(name, out, slots)
        continue
      out[name] = self._gather_full(name, self.table_view(name)).cpu().numpy()
      if slots:
        for s in ('m', 'v'):
          sv = self.slot_view(name, s)
          if sv is not None:
            out[name + '/' + s] = self._gather_full(name, sv).cpu().numpy()
    return out

  def load_kv_table(self, name, keys, values, slot_values, seen=None, freq=None, version=None):
    """This rank's share of the saved table: the ids with id % world == rank."""
    import numpy as np
    keys = np.asarray(keys, dtype=np.int64)
    mine = (keys % self.world) == self.rank
    pick = lambda a: None if a is None else np.asarray(a)[mine]
    if seen is not None:
      seen = np.asarray(seen, dtype=np.int64)
      smine = (seen % self.world) == self.rank
      seen, freq, version = seen[smine], np.asarray(freq)[smine], np.asarray(version)[smine]
    super(ShardedEmbeddingEngine, self).load_kv_table(name, keys[mine], pick(values), {k: pick(v) for k, v in slot_values.items()},
                                                      seen, freq, version)

  def load_state_dict(self, state):
    import numpy as np
    for name in self.tables:
      if name not in state:
        continue
      if self.tables[name].get('kv'):
        slot_values = {sl: state[name + '/' + sl] for sl in ('m', 'v') if (name + '/' + sl) in state}
        self.load_kv_table(name, state[name + '/keys'], state[name], slot_values, state.get(name + '/kv_seen_keys'),
                           state.get(name + '/kv_freq'), state.get(name + '/kv_version'))
        continue
      kind, dim, base, n_local = self.placement[name]
      full = torch.from_numpy(np.asarray(state[name], dtype=np.float32)).to(self.device)
      view = self.table_view(name)
      if kind == 'rep':
        view.copy_(full)
      else:
        mine = full[self.rank::self.world]
        view.zero_()
        view[:mine.shape[0]].copy_(mine)

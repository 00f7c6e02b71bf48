"""InputLayer: the embedding stage of the hot path, one fused HIP launch per step.

Host-side mirror of reference easy_rec/python/layers/input_layer.py:28-398 (`InputLayer`):
  __call__(features, group_name) -> (concat [B, sum(dim)], [per-feature tensors])  (:245-278)
  single_call_input_layer: columns in config order, concat on axis 1, L2 on the looked-up
  outputs of every embedding column (:280-376, compat/regularizers.py:76-108,169-208).
What the reference builds as ~6 TF ops per column (feature_column_v2.py:3434-3462) is compiled here
into a static list of `LookupSpec`s executed by `er_emb_fwd` (all groups, one launch) and
`er_emb_bwd_update` (one sort + reduce + row-wise optimizer per table dim).

HBM layout: all tables of one embedding dim are stored back to back in one [total_rows, dim] fp32
buffer (+ identical buffers for the optimizer slots), in creation order; a lookup addresses its
table by `key_base` (first row).  Variable names follow TF's scoping
(`input_layer[_k]/<column>_embedding/embedding_weights`, compat/feature_column/feature_column.py:
384-414) so per-table views can be exported under the reference's names.
"""
import logging
import os
import math
from collections import OrderedDict

import numpy as np
import torch

from easyrec_amd import kernels
from easyrec_amd.feature_column.feature_column import (EmbeddingColumn, FeatureColumnParser, NumericColumn,
                                                       SequenceNumericColumn)
from easyrec_amd.feature_column.feature_group import FeatureGroup
from easyrec_amd.protos.feature_config_pb2 import WideOrDeep


class FeatureList(list):
  """List of per-feature [B, dim] views that also remembers the concat block they live in, so that
  consumers (FM, DIN) can read the block directly instead of re-stacking (layers/fm.py:22)."""

  def __init__(self, views, base=None, col0=0, dims=None):
    super(FeatureList, self).__init__(views)
    self.base = base
    self.col0 = col0
    self.dims = list(dims) if dims is not None else None

  def uniform_block(self):
    """(base, col0, F, D) when all features have the same dim and are adjacent in `base`."""
    if self.base is None or not self.dims or len(set(self.dims)) != 1:
      return None
    return self.base, self.col0, len(self.dims), self.dims[0]


class _GroupOutFn(torch.autograd.Function):
  """Makes a persistent group-output buffer a differentiable leaf-like tensor: backward deposits
  (grad + lambda * out) into the group's static gradient buffer, which `er_emb_bwd_update` reads."""

  @staticmethod
  def forward(ctx, anchor, engine, gkey):
    ctx.engine, ctx.gkey = engine, gkey
    # consumers that deposit their gradient through the GradSink return None to autograd: without this autograd would
    # hand backward() a zero tensor (one fill + one add per group and step for nothing)
    ctx.set_materialize_grads(False)
    return engine.groups[gkey]['out'].view_as(engine.groups[gkey]['out'])

  @staticmethod
  def backward(ctx, g):
    if g is not None:
      ctx.engine._deposit_grad(ctx.gkey, g)
    return None, None, None


class GradSink(object):
  """Lets the backward kernel of a group output's consumer write d(loss)/d(out) straight into the group's gradient
  buffer (the one er_emb_bwd_update reads) instead of returning a tensor that autograd would add to the other
  consumers' and that _GroupOutFn.backward would then copy: `target()` gives (view of dout, accumulate flag) for the
  column block, `done()` records the deposit (and adds the output-L2 term once)."""

  def __init__(self, engine, gkey):
    self.engine, self.gkey = engine, gkey

  def target(self, col0=0, width=None):
    grp = self.engine.groups[self.gkey]
    dout = grp['dout']
    if width is None:
      width = dout.shape[1] - col0
    return dout[:, col0:col0 + width], grp['got_grad']

  def covers(self, col0, width):
    """A first deposit must initialise the WHOLE buffer: partial blocks may only accumulate."""
    grp = self.engine.groups[self.gkey]
    return grp['got_grad'] or (col0 == 0 and width == grp['dout'].shape[1])

  def done(self):
    self.engine._after_deposit(self.gkey)

  def defer(self, term):
    """A contribution that is cheap to add elementwise (`kernels.HipBackend.group_grad_finish` terms): recorded now,
    added when the engine finishes the group's gradient buffer - one launch for all groups of the model."""
    self.engine.groups[self.gkey]['terms'].append(term)


class EmbeddingEngine(object):
  """Owns tables, optimizer slots, lookup specs and the C handles."""

  def __init__(self, device, batch_size, seed=0):
    self.device = torch.device(device)
    self.batch_size = batch_size
    self.seed = seed
    self.tables = OrderedDict()  # var name -> dict(dim, rows, key_base, init)
    self.dim_rows = OrderedDict()  # dim -> total rows so far
    self.groups = OrderedDict()  # gkey -> dict(out, dout, specs, reg, width)
    self.finalized = False
    self.plan = None
    self.emb_groups = OrderedDict()  # dim -> C group handle
    self.storage = {}  # dim -> dict(var, m, v, bitmap)
    # value of an optimizer slot on a row no update has touched: 0 for Adam's m / v; the estimator sets 'v' to Adagrad's
    # initial_accumulator_value.  Rows of a hash-table arena that are freed (evict_stale) or not covered by a restored
    # checkpoint (load_kv_table) go back to it - NOT to 0, or a resumed Adagrad run would take lr * sign(g) steps on new ids
    self.slot_init = {'m': 0.0, 'v': 0.0}
    self._pending_tables = []
    self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
    self.sumsq = None
    self.reg_blocks = 0
    self._ran_version = -1
    self._sweep_stream = None
    self._sweep_pending = False
    # TF-exact Adam without the dense sweep (er_emb_catch_up): set by the estimator through set_step_clock()
    self.lazy_decay = False
    self._clock = None  # (step_counter int64[1], lr_t history fp32[capacity], hyper record of the embeddings)
    self._lazy = OrderedDict()  # dim -> route buffers + last_step
    self.share_sort = True  # er_emb_group_share_sort between table groups that read the same ids
    # lazy dense decay: rows carry pending decay steps after a training step (backward_update) until flush_decay();
    # inference lookups (predict / evaluate) must read current rows WITHOUT replaying anything twice
    self._decay_pending = False
    self.inference = False
    # rolling flush: every step one window of every table group is brought current, so no row is ever more than
    # this many steps behind (er_emb_flush_window); 0 = off (rows wait for their next touch or a full flush)
    self.flush_windows = 64
    # '1': the window's launch runs on a second stream, concurrently with the step (lag 1: see er_emb_flush_window), joined
    # after the step's row update; default: after the row update on the main stream (lag 0).  Measured on DeepFM-Criteo
    # (profiles/r02_overlap_flush.md): no gain - the steady-state step is bound by the SUM of its kernels' durations
    # (graph replay leaves no gaps) and the replay is VALU-bound, so running it next to the GEMMs only slows those down.
    self.overlap_flush = os.environ.get('EASYREC_AMD_OVERLAP_FLUSH', '0') != '0'
    self.flush_blocks = int(os.environ.get('EASYREC_AMD_FLUSH_BLOCKS', '1024'))  # grid of the concurrent launch (4 / CU)
    self._flush_stream = None
    self.train_mode = True  # (the estimator clears it for is_training=False: no hash-table rows are created then)
    self.kv_tables = {}  # table name -> map state (kernels.kv_create) of the hash-table backed tables
    self._kv_handle = None
    self.kv_jobs = []    # (table name, id array, arena-row array) translated before every lookup
    self._window_pending = False  # launched on the second stream, not joined yet
    self._window_started = False  # this step's window has been launched (the row update must not launch it again)
    self._sort_leader = {}  # dim -> dim of the group whose per-step sort it reuses
    # the fused single-GPU step (kernels.HipBackend.emb_front / emb_bwd_fused): decided per engine at the first step
    # (None = not tried yet; False = these groups need the general path); the estimator clears `allow_fused` when
    # something between lookup and row update needs the general path's buffers (gradient clipping by global norm)
    self.allow_fused = True
    self._fused = None

  # -- declaration (build pass)
  def declare_table(self, var_name, rows, dim, initializer=None, kv_capacity=None, kv_filter_freq=0, kv_steps_to_live=0):
    """kv_capacity: a hash-table (`ev_params`) table - `rows` is then the capacity of its arena, ids are translated to
    arena rows before every lookup (er_kv_translate) and rows are created on first sight - or, with kv_filter_freq > 1,
    once the id has been seen that often; kv_steps_to_live > 0: ids not seen for that many steps are dropped when a
    checkpoint is written (evict_stale)."""
    if kv_capacity is not None:
      rows = int(kv_capacity)
    if var_name in self.tables:
      t = self.tables[var_name]
      assert t['rows'] == rows and t['dim'] == dim, 'shared table %s shape mismatch' % var_name
      return t
    assert not self.finalized, 'table %s declared after finalize()' % var_name
    base = self.dim_rows.get(dim, 0)
    self.dim_rows[dim] = base + rows
    t = {'name': var_name, 'rows': int(rows), 'dim': int(dim), 'key_base': base, 'init': initializer,
         'kv': kv_capacity is not None, 'kv_filter_freq': int(kv_filter_freq), 'kv_steps_to_live': int(kv_steps_to_live)}
    self.tables[var_name] = t
    return t

  def declare_group(self, gkey, width, regularize):
    assert gkey not in self.groups
    B = self.batch_size
    g = {
        'out': torch.zeros(B, width, dtype=torch.float32, device=self.device),
        'dout': torch.zeros(B, width, dtype=torch.float32, device=self.device),
        'pending': [],  # (table name, ids, offsets, weights, col, combiner, n_rows, max_nnz, name)
        'reg': float(regularize or 0.0),
        'width': width,
        'got_grad': False,
        'terms': [],
    }
    self.groups[gkey] = g
    return g

  def declare_seq_output(self, gkey, n_rows, width, regularize):
    """Sequence lookups keep the time axis: output [B*L, width]."""
    g = {
        'out': torch.zeros(n_rows, width, dtype=torch.float32, device=self.device),
        'dout': torch.zeros(n_rows, width, dtype=torch.float32, device=self.device),
        'pending': [],
        'reg': float(regularize or 0.0),
        'width': width,
        'got_grad': False,
        'terms': [],
    }
    self.groups[gkey] = g
    return g

  def add_lookup(self, gkey, table_name, ids, offsets, weights, col, combiner, n_rows, max_nnz, name):
    self.groups[gkey]['pending'].append((table_name, ids, offsets, weights, col, combiner, n_rows, max_nnz, name))

  def set_step_clock(self, step_counter, lr_hist, hyper_emb, lazy_decay=True):
    """Device step counter + per-step lr_t history + the embeddings' hyper record: what the lazy dense-decay
    catch-up of TF-exact Adam reads.  Call before finalize()."""
    self._clock = (step_counter, lr_hist, hyper_emb)
    self.lazy_decay = bool(lazy_decay)

  def set_decay_tables(self, tabs):
    """The closed-form replay's tables (kernels.decay_tables_create; None = exact step-by-step replay).  With them a
    catch-up costs the same whatever the backlog, so the rolling flush that bounds the backlog is switched off."""
    self._decay_tables = tabs
    if tabs is not None:
      self.flush_windows = 0
    self._set_prologue_tables(tabs)
    for grp, _ in self._lazy_groups():
      kernels.hip().emb_group_set_decay_tables(grp, tabs)

  def _set_prologue_tables(self, tabs):
    """The step's lag-1 replay table from the prologue launch (er_decay_tables_set_prologue_build), so that the fused step's
    sort and lookup are ONE launch (er_emb_front_fwd).  Only the single-GPU engine's fused step keeps the prologue's
    counter word current by itself; forward() does it explicitly after any other kind of training lookup."""
    be = kernels.hip()
    self._prologue_tables = bool(tabs is not None and type(self) is EmbeddingEngine and getattr(be, 'prologue_tables', False) and
                                 getattr(be, 'defer_catch_up', False) and hasattr(be, 'emb_front_fwd'))
    if tabs is not None and hasattr(be, 'decay_tables_set_prologue_build'):
      be.decay_tables_set_prologue_build(tabs, self._prologue_tables)

  def _enable_lazy_decay(self, dim, grp, st, n_route):
    """Per table group: the last-updated-step array and the buffers of er_emb_route (unique rows of a step)."""
    be = kernels.hip()
    dev = self.device
    lz = {
        'last_step': st['last_step'] if st.get('last_step') is not None
                     else torch.full((st['total_rows'],), -1, dtype=torch.int32, device=dev),
        'ukeys': torch.zeros(max(n_route, 1), dtype=torch.int32, device=dev),
        'n_unique': torch.zeros(1, dtype=torch.int32, device=dev),
    }
    be.emb_group_enable_lazy_decay(grp, lz['last_step'], self._clock[1], self._clock[0])
    if getattr(self, '_decay_tables', None) is not None:
      be.emb_group_set_decay_tables(grp, self._decay_tables)
    return lz

  def _lazy_groups(self):
    """[(C group handle, its lazy-decay state)] of the table groups whose rows carry pending decay."""
    return [(self.emb_groups[dim], lz) for dim, lz in self._lazy.items()]

  def rebind_lr_history(self, lr_hist, decay_tables=None):
    """The estimator re-allocated the per-step lr_t history (it grew): point the table groups at the new buffer (and
    at the closed-form tables re-created for it)."""
    if self._clock is None:
      return
    self._clock = (self._clock[0], lr_hist, self._clock[2])
    self._decay_tables = decay_tables
    self._set_prologue_tables(decay_tables)
    be = kernels.hip()
    for grp, lz in self._lazy_groups():
      be.emb_group_set_decay_tables(grp, None)
      be.emb_group_enable_lazy_decay(grp, lz['last_step'], lr_hist, self._clock[0])
      if decay_tables is not None:
        be.emb_group_set_decay_tables(grp, decay_tables)

  def flush_decay(self):
    """Bring every row current (lazy dense decay): before reading tables out (state_dict, evaluation)."""
    if not self.lazy_decay:
      return
    be = kernels.hip()
    self._join_window_flush()
    for dim, grp in self.emb_groups.items():
      be.emb_flush_decay(grp, self._clock[2])
    self._decay_pending = False

  def begin_inference(self):
    """Lookups that no row update follows (predict / evaluate).  The catch-up kernel leaves `last_step` to the row
    update of the same step, so running it without one would replay the same pending decay on every call: instead every
    row is brought current once (only if a training step ran since the last flush) and the lookups skip the catch-up."""
    if self.lazy_decay and self._decay_pending:
      self.flush_decay()
    self.inference = True

  def end_inference(self):
    self.inference = False

  def mark_restored(self, step):
    """Tables were loaded as of `step` finished steps: no decay is pending on any row."""
    for lz in self._lazy_states():
      lz['last_step'].fill_(int(step) - 1)
    self._decay_pending = False
    if getattr(self, '_prologue_tables', False):  # (the step counter was set: the prologue's copy of it follows)
      kernels.hip().decay_tables_sync(self._decay_tables)

  def _lazy_states(self):
    return list(self._lazy.values())

  @staticmethod
  def record_floats(dim, n_slots, with_step):
    """Floats per row record [var | slot ... | last_step | pad]: a power of two up to 16 floats, then whole 64-byte lines,
    so that a record never straddles more lines than it needs (dim 16 + Adam + last_step: 49 -> 64 floats = 256 bytes;
    dim 1: 4 floats = 16 bytes)."""
    n = (1 + n_slots) * dim + (1 if with_step else 0)
    if n <= 16:
      p = 1
      while p < n:
        p *= 2
      return max(p, 4 if dim % 4 == 0 else 1)
    return (n + 15) // 16 * 16

  def _alloc_storage(self, total, dim, opt_kind, force_bitmap=False):
    """The tables of one embedding dim.  Row records (the default): var, the optimizer slots and - under lazy dense
    decay - the row's last_step lie side by side in ONE [total, ld] buffer, `var` / `m` / `v` / `last_step` being column
    blocks of it, so a touched row is one contiguous HBM access per pass instead of three or four scattered ones
    (er_emb_group_set_row_pitch).  TensorFlow keeps every slot in a variable of its own
    (tf.train.AdamOptimizer._create_slots); state_dict / checkpoints convert at that boundary.  EASYREC_AMD_ROW_RECORDS=0:
    plain [total, dim] arrays."""
    adam = opt_kind in (kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM)
    n_slots = 2 if adam else (1 if opt_kind == kernels.OPT_ADAGRAD else 0)
    with_step = bool(self.lazy_decay) and opt_kind == kernels.OPT_ADAM and not force_bitmap
    st = {'var': None, 'm': None, 'v': None, 'bitmap': None, 'total_rows': total, 'last_step': None, 'rec': None}
    if os.environ.get('EASYREC_AMD_ROW_RECORDS', '1') != '0' and n_slots > 0:
      ld = self.record_floats(dim, n_slots, with_step)
      plain = (1 + n_slots) * dim + (1 if with_step else 0)
      if total * ld * 4 >= (1 << 30):  # (what the padding to whole 64-byte lines costs, where it matters)
        logging.info('easyrec_amd: %d rows of dim %d as %d-float records: %.1f GB (%.1f GB unpadded; EASYREC_AMD_ROW_RECORDS=0 '
                     'for plain arrays)', total, dim, ld, total * ld * 4 / 2 ** 30, total * plain * 4 / 2 ** 30)
      rec = torch.zeros(total, ld, dtype=torch.float32, device=self.device)
      st['rec'] = rec
      st['var'] = rec[:, 0:dim]
      if adam:
        st['m'], st['v'] = rec[:, dim:2 * dim], rec[:, 2 * dim:3 * dim]
      else:
        st['v'] = rec[:, dim:2 * dim]
      if with_step:
        st['last_step'] = rec.view(torch.int32)[:, (1 + n_slots) * dim]
        st['last_step'].fill_(-1)
    else:
      var = torch.empty(total, dim, dtype=torch.float32, device=self.device)
      st['var'] = var
      if adam:
        st['m'] = torch.zeros_like(var)
        st['v'] = torch.zeros_like(var)
      elif opt_kind == kernels.OPT_ADAGRAD:
        st['v'] = torch.zeros_like(var)
    if opt_kind == kernels.OPT_ADAM and (force_bitmap or not self.lazy_decay):
      st['bitmap'] = torch.zeros((total + 31) // 32, dtype=torch.int32, device=self.device)
    return st

  @staticmethod
  def _init_mean_std(t):
    """(mean, stddev) of a hash-table row's initial values: the column's (truncated) normal initializer, else the
    default of an embedding column (feature_column_v2.py:908-912)."""
    init = t['init']
    kind = init.WhichOneof('initializer_oneof') if init is not None else None
    if kind == 'truncated_normal_initializer':
      return init.truncated_normal_initializer.mean, init.truncated_normal_initializer.stddev
    if kind == 'random_normal_initializer':
      return init.random_normal_initializer.mean, init.random_normal_initializer.stddev
    assert kind is None, 'hash-table embedding %s: initializer %s is not supported' % (t['name'], kind)
    return 0.0, 0.01 / math.sqrt(t['dim'])

  def translate_kv_ids(self):
    """ids -> arena rows for every lookup of a hash-table backed table; unseen ids get a row while training and read
    zeros (row -1) in predict() / evaluate() (feature_column_v2.py:3487-3493)."""
    be = kernels.hip()
    insert = self.train_mode and not self.inference
    if self._kv_handle is None:  # every lookup of every hash-table table in one pair of launches
      self._kv_handle = be.kv_jobs_create([(self.kv_tables[job[0]],) + tuple(job[1:]) for job in self.kv_jobs])
    be.kv_translate_multi(self._kv_handle, insert)

  def check_overflow(self):
    """Sticky device flags that void the steps since they were set (read back: a host sync): here the hash-table arenas;
    the sharded engine adds its exchange capacity.  Called every OVERFLOW_CHECK_EVERY steps by the estimator, by
    evaluate(), state_dict() and checkpoint.save()."""
    self.check_kv_overflow()
    if getattr(self, '_prologue_tables', False) and kernels.hip().decay_tables_error(self._decay_tables):
      raise RuntimeError('lazy dense decay: a training lookup found the replay table built for another step (a training '
                         'step ran its prologue without a lookup?): the steps since are void')

  def check_kv_overflow(self):
    for name, kv in self.kv_tables.items():
      if int(kv['overflow'].item()):
        raise RuntimeError('hash-table embedding %s: more than %d distinct ids (raise ev_params.max_capacity)'
                           % (name, kv['capacity']))

  def init_table_values(self, name, view):
    """Fill `view` ([rows, dim]) with the initial values of table `name` (seeded per table name, so the
    single-GPU engine and every rank of the sharded engine draw the same table)."""
    if not view.is_contiguous():  # a column block of the row records: draw into a plain array (the same values), then copy
      tmp = torch.empty(view.shape, dtype=view.dtype, device=view.device)
      self.init_table_values(name, tmp)
      view.copy_(tmp)
      return
    t = self.tables[name]
    init = t['init']
    gen = torch.Generator(device=self.device)
    gen.manual_seed(_stable_seed(name, self.seed))
    if init is not None and init.WhichOneof('initializer_oneof') == 'constant_initializer':
      consts = list(init.constant_initializer.consts)
      vals = torch.tensor(consts, dtype=torch.float32, device=self.device)
      view.copy_(vals.view(-1)[:view.numel()].view_as(view) if vals.numel() >= view.numel() else
                 vals.expand_as(view))
    elif init is not None and init.WhichOneof('initializer_oneof') == 'random_normal_initializer':
      view.normal_(init.random_normal_initializer.mean, init.random_normal_initializer.stddev, generator=gen)
    elif init is not None and init.WhichOneof('initializer_oneof') == 'glorot_normal_initializer':
      std = math.sqrt(2.0 / (t['rows'] + t['dim']))
      torch.nn.init.trunc_normal_(view, 0.0, std, -2 * std, 2 * std, generator=gen)
    else:
      mean, std = 0.0, 0.01 / math.sqrt(t['dim'])  # feature_column_v2.py:908-912
      if init is not None and init.WhichOneof('initializer_oneof') == 'truncated_normal_initializer':
        mean = init.truncated_normal_initializer.mean
        std = init.truncated_normal_initializer.stddev
      torch.nn.init.trunc_normal_(view, mean, std, mean - 2 * std, mean + 2 * std, generator=gen)

  def _ordered_groups(self):
    """regularised groups first so their sum-of-squares partials form a prefix"""
    return sorted(self.groups.items(), key=lambda kv: 0 if kv[1]['reg'] > 0 else 1)

  # -- storage
  def finalize(self, opt_kind):
    """Allocate table groups, initialise tables, create the C handles."""
    assert not self.finalized
    be = kernels.hip()
    for dim, total in self.dim_rows.items():
      self.storage[dim] = self._alloc_storage(total, dim, opt_kind)
    for name, t in self.tables.items():
      if t['kv']:
        # rows are created on first sight by er_kv_translate (a pure function of seed, id and column); the arena
        # starts as zeros (a row that no id owns is never read)
        self.table_view(name).zero_()
        mean, std = self._init_mean_std(t)
        assert t['kv_steps_to_live'] == 0 or self._clock is not None, 'steps_to_live needs the step clock (set_step_clock)'
        self.kv_tables[name] = be.kv_create(self.table_view(name), t['rows'], _stable_seed(name, self.seed), mean, std,
                                            filter_freq=t['kv_filter_freq'], steps_to_live=t['kv_steps_to_live'],
                                            step=self._clock[0] if t['kv_steps_to_live'] > 0 else None)
      else:
        self.init_table_values(name, self.table_view(name))
    # lookup specs: regularised groups first so their sum-of-squares partials form a prefix
    ordered = self._ordered_groups()
    fwd_specs, reg_count = [], 0
    for gkey, g in ordered:
      g['specs'] = []
      for (tname, ids, offsets, weights, col, combiner, n_rows, max_nnz, name) in g['pending']:
        t = self.tables[tname]
        spec = kernels.LookupSpec(
            table=self.table_view(tname), ids=ids, offsets=offsets, weights=weights, out=g['out'],
            out_col=col, rows=t['rows'], key_base=t['key_base'], dim=t['dim'],
            combiner=kernels.COMBINERS[combiner], n_rows=n_rows, max_nnz=max_nnz, name=name)
        g['specs'].append(spec)
        fwd_specs.append(spec)
        if g['reg'] > 0:
          reg_count += 1
    self.fwd_specs = fwd_specs
    self.n_reg_specs = reg_count
    if fwd_specs:
      self.plan = be.emb_plan_create(fwd_specs)
      self.sumsq = torch.zeros(max(self.plan['num_blocks'], 1), dtype=torch.float32, device=self.device)
      self.reg_blocks = _blocks_of(fwd_specs[:reg_count])
      assert len({g['reg'] for g in self.groups.values() if g['reg'] > 0}) <= 1, \
          'one embedding_regularization value per model'
    for dim, st in self.storage.items():
      specs = [s.with_out(self._dout_of(s)) for s in fwd_specs if s.dim == dim]
      if specs:
        self.emb_groups[dim] = be.emb_group_create(specs, dim, st['total_rows'], st['var'], st['m'], st['v'],
                                                   st['bitmap'])
        if self.lazy_decay and opt_kind == kernels.OPT_ADAM:
          self._lazy[dim] = self._enable_lazy_decay(dim, self.emb_groups[dim], st, self.emb_groups[dim]['num_entries'])
    self.lazy_decay = self.lazy_decay and opt_kind == kernels.OPT_ADAM
    # wide and deep groups built from the same columns see the same keys: sort them once per step
    self._sort_leader = {}
    dims = list(self.emb_groups)
    for dim in dims[1:]:
      if self.share_sort and be.emb_group_share_sort(self.emb_groups[dim], self.emb_groups[dims[0]]):
        self._sort_leader[dim] = dims[0]
    self.reg_lambda = max([g['reg'] for g in self.groups.values()] + [0.0])
    self.finalized = True

  def _dout_of(self, spec):
    for g in self.groups.values():
      if g['out'] is spec.out:
        return g['dout']
    raise KeyError('spec output not registered')

  def table_view(self, name):
    t = self.tables[name]
    var = self.storage[t['dim']]['var']
    return var[t['key_base']:t['key_base'] + t['rows']]

  def slot_view(self, name, slot):
    t = self.tables[name]
    buf = self.storage[t['dim']][slot]
    return None if buf is None else buf[t['key_base']:t['key_base'] + t['rows']]

  # -- per-step execution
  def forward(self, version):
    if version == self._ran_version:
      return
    be = kernels.hip()
    for g in self.groups.values():
      g['got_grad'] = False
      g['terms'] = []
    self._join_window_flush()  # (a forward that no row update followed)
    if self.kv_jobs:
      self.translate_kv_ids()
    lazy_lookup = looked_up = False
    if self.lazy_decay and not self.inference and self._use_fused() and \
        self._fused_front(with_lookup=getattr(self, '_prologue_tables', False) and self.plan is not None):
      lazy_lookup = self._front_deferred
      looked_up = self._front_looked_up
      self._start_window_flush()
    elif self.lazy_decay and not self.inference:
      # sort the step's ids once (reused by the backward), bring the rows it touches up to date, then look up
      grps, uks, nus = [], [], []
      for dim, grp in self.emb_groups.items():
        lz = self._lazy[dim]
        if dim in self._sort_leader:
          lz = self._lazy[self._sort_leader[dim]]
          be.emb_route(grp, None, None, None, None)
        else:
          be.emb_route(grp, lz['ukeys'], lz['n_unique'], None, None)
        grps.append(grp)
        uks.append(lz['ukeys'])
        nus.append(lz['n_unique'])
      step = 4
      probe = getattr(self, 'catch_up_probe', None)  # bench.py: (start event, end event) around the catch-up launches
      if probe is not None:
        probe[0].record()
      for i in range(0, len(grps), step):  # the table groups' catch-up kernels side by side in one launch
        be.emb_catch_up_multi(grps[i:i + step], uks[i:i + step], nus[i:i + step], self._clock[2])
      if probe is not None:
        probe[1].record()
      self._start_window_flush()
    if getattr(self, '_prologue_tables', False) and self.train_mode and not self.inference and not lazy_lookup:
      # a training lookup that is not the lazy one: the next prologue's table workgroups read the step from this word
      be.decay_tables_sync(self._decay_tables)
    if self.plan is not None and not looked_up:
      if lazy_lookup:
        be.emb_fwd_lazy(self.plan, list(self.emb_groups.values()), self._clock[2], self.sumsq if self.reg_lambda > 0 else None)
      else:
        be.emb_fwd(self.plan, self.sumsq if self.reg_lambda > 0 else None)
    self._ran_version = version

  def group_tensor(self, gkey, requires_grad=True):
    if requires_grad and torch.is_grad_enabled():
      t = _GroupOutFn.apply(self._anchor, self, gkey)
      t._er_sink = GradSink(self, gkey)
      return t
    return self.groups[gkey]['out']

  def _deposit_grad(self, gkey, g):
    be = kernels.hip()
    grp = self.groups[gkey]
    g2 = g.reshape(grp['dout'].shape)
    if not g2.is_contiguous():
      g2 = g2.contiguous()
    be.axpy2d(g2, 1.0, grp['dout'], accumulate=grp['got_grad'])
    self._after_deposit(gkey)

  def _after_deposit(self, gkey):
    self.groups[gkey]['got_grad'] = True  # (dout holds a base; lambda * out and the deferred terms come at the finish)

  def finish_group_grads(self):
    """Every group's gradient buffer complete, in ONE launch: base (what GEMMs / autograd deposited, else zero) +
    deferred terms (row-sum broadcast, FM) + lambda * out, the gradient of the embedding-output L2
    (layers/input_layer.py:369-375).  Called first thing by the embedding backward."""
    descs = []
    for g in self.groups.values():
      lam = g['reg'] if g['reg'] > 0 else 0.0
      if g['got_grad'] and not g['terms'] and lam == 0.0:
        continue  # already complete
      terms = g['terms']
      while len(terms) > 4:  # (more than 4 deferred terms on one group: finish in rounds)
        descs.append((g['dout'], g['out'], 0.0, g['got_grad'], terms[:4]))
        terms, g['got_grad'] = terms[4:], True
        kernels.hip().group_grad_finish(descs[-1:])
        descs.pop()
      descs.append((g['dout'], g['out'], lam, g['got_grad'], terms))
    if descs:
      kernels.hip().group_grad_finish(descs)
    for g in self.groups.values():
      g['terms'] = []
      g['got_grad'] = True

  def regularization_loss(self, out):
    """out[0] = lambda * 0.5 * sum(out^2) over regularised embedding outputs."""
    if self.reg_lambda > 0 and self.reg_blocks > 0:
      kernels.hip().reduce_sum(self.sumsq[:self.reg_blocks], 0.5 * self.reg_lambda, out, accumulate=False)
    else:
      out.zero_()

  # -- TF-exact Adam: the dense-decay sweep of the untouched rows, overlapped on a second stream
  def start_decay_sweep(self, hyper):
    """Call once the step's ids are on device (after DeviceFeatures.transform()).  Marks the rows the
    step touches (main stream), then forks: the side stream sweeps every OTHER row of every table
    group while the main stream runs forward/backward/row updates.  join_decay_sweep() joins."""
    be = kernels.hip()
    if self._sweep_stream is None:
      self._sweep_stream = torch.cuda.Stream(device=self.device)
    main = torch.cuda.current_stream()
    for grp in self.emb_groups.values():
      be.emb_mark_touched(grp)
    self._sweep_stream.wait_stream(main)
    with torch.cuda.stream(self._sweep_stream):
      for grp in self.emb_groups.values():
        be.emb_sweep_untouched(grp, hyper)
    self._sweep_pending = True

  def join_decay_sweep(self):
    if self._sweep_pending:
      torch.cuda.current_stream().wait_stream(self._sweep_stream)
      self._sweep_pending = False

  # -- the fused single-GPU step
  def _use_fused(self):
    be = kernels.hip()
    # (bench.py's catch-up probe brackets the general path's catch-up launches with events: the fused front is one C call)
    return (self._fused is not False and self.allow_fused and getattr(be, 'fused_emb', False) and self.train_mode and
            getattr(self, 'catch_up_probe', None) is None and
            type(self) is EmbeddingEngine and not self.kv_jobs and len(self.emb_groups) <= 4 and
            not self._sweep_pending and (not self.lazy_decay or self.flush_windows <= 0) and len(self.groups) <= 8 and
            not any(st['bitmap'] is not None for st in self.storage.values()))

  def _fused_front(self, with_lookup=False):
    """er_emb_front over all table groups (leader first); remembers whether the groups are eligible.  with_lookup: the
    step's lookup in the same call (er_emb_front_fwd: one launch with the sort)."""
    be = kernels.hip()
    grps = list(self.emb_groups.values())
    self._front_looked_up = False
    if with_lookup and self._fused is not False:
      ok = be.emb_front_fwd(grps, self.plan, self._clock[2], True, self.sumsq if self.reg_lambda > 0 else None)
      if self._fused is None:
        self._fused = bool(ok)
      else:
        assert ok == self._fused
      self._front_done = self._front_deferred = self._front_looked_up = bool(ok)
      return ok
    # lazy dense decay in closed form: no catch-up launch - the lookup and the row update evaluate a row's pending steps
    # in registers (er_emb_fwd_lazy)
    defer = bool(self.lazy_decay and getattr(be, 'defer_catch_up', False) and getattr(self, '_decay_tables', None) is not None)
    ok = be.emb_front(grps, self._clock[2] if self.lazy_decay else None, True, defer=defer)
    self._front_deferred = bool(ok and defer)
    if self._fused is None:
      self._fused = bool(ok)
      if not ok:
        logging.info('easyrec_amd: the fused embedding step does not cover this model (%s): general path',
                     getattr(be, 'last_error', lambda: '')())
    else:
      assert ok == self._fused
    self._front_done = bool(ok)
    return ok

  def _finish_descs(self):
    """group_grad_finish's descriptors for EVERY group buffer (a complete one: base only)."""
    descs = []
    for g in self.groups.values():
      lam = g['reg'] if g['reg'] > 0 else 0.0
      descs.append((g['dout'], g['out'], lam, g['got_grad'], g['terms']))
    return descs

  def backward_update(self, opt_kind, hyper, pending_wgrads=False, dense_opt=None):
    """pending_wgrads: the dense layers' weight gradients of this backward pass are still queued (model.backward(flush=
    False)): the fused step contracts them in the row update's grid (er_emb_bwd_fused_tail), every other path launches
    them first.  dense_opt: be.dense_opt_step's arguments - the fused tail may run the dense optimizer too; returns True
    when it did."""
    be = kernels.hip()
    # this step's front already ran fused (forward, lazy dense decay), or - optimizers without it - runs now
    if getattr(self, '_front_done', False) or (not self.lazy_decay and self._use_fused() and self._fused_front()):
      self._front_done = False
      for g in self.groups.values():  # (more than 4 deferred terms on one buffer: finish the surplus into it first)
        while len(g['terms']) > 4:
          be.group_grad_finish([(g['dout'], g['out'], 0.0, g['got_grad'], g['terms'][:4])])
          g['terms'], g['got_grad'] = g['terms'][4:], True
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
      if dense_opt is not None and not (wgrads is not None and getattr(be, 'tail_riders', False) and
                                        be.dense_opt_fits_the_tail(wgrads, dense_opt[0], dense_opt[3])):
        dense_opt = None
      # (the narrow groups' long tiles FIRST in the grid: measured, no change - profiles/r05_s14_*)
      ran_opt = be.emb_bwd_fused(list(self.emb_groups.values()), self._finish_descs(), opt_kind, hyper, wgrads=wgrads,
                                 dense_opt=dense_opt)
      for g in self.groups.values():
        g['terms'] = []
        g['got_grad'] = True
      self._roll_flush(hyper)
      self._decay_pending = True
      return bool(ran_opt)
    if pending_wgrads:
      be.flush_wgrads()
    be.flush_loss_tail()
    self.finish_group_grads()
    if opt_kind == kernels.OPT_ADAM and self._sweep_pending:
      # the sweep of the untouched rows is already in flight on the side stream; the touched rows
      # get the same per-row arithmetic as TF's sparse apply (== the lazy row update)
      opt_kind = kernels.OPT_LAZY_ADAM
    grps = list(self.emb_groups.values())
    step = 4
    for i in range(0, len(grps), step):  # one tile launch + one fix launch for (up to 4) table groups
      be.emb_bwd_update_multi(grps[i:i + step], opt_kind, hyper)
    self.join_decay_sweep()
    self._roll_flush(hyper)
    self._decay_pending = True
    return False

  def _roll_flush(self, hyper):
    """After the step's row updates: this step's window of every lazily decaying table group (one launch per 4) - or,
    when the window was started next to the step (_start_window_flush), the join with it."""
    if not self.lazy_decay or self.flush_windows <= 0:
      return
    if self._window_started:  # (this step's window ran next to the step)
      self._join_window_flush()
      self._window_started = False
      return
    lazy = [grp for grp, _ in self._lazy_groups()]
    for i in range(0, len(lazy), 4):
      kernels.hip().emb_flush_window(lazy[i:i + 4], self.flush_windows, hyper)

  def _start_window_flush(self):
    """Right after the step's catch-up: the step's window on the second stream, bringing its rows to the step BEFORE
    this one (lag 1) - exactly where the catch-up left the rows the step touches, so the launch skips those and is
    independent of the lookup, the dense part and the row update it runs next to."""
    if not (self.lazy_decay and self.overlap_flush and self.flush_windows > 0):
      return
    lazy = [grp for grp, _ in self._lazy_groups()]
    if not lazy:
      return
    be, hyper = kernels.hip(), self._clock[2]
    if self.device.type == 'cuda':
      if self._flush_stream is None:
        self._flush_stream = torch.cuda.Stream(device=self.device)
      self._flush_stream.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(self._flush_stream):
        for i in range(0, len(lazy), 4):
          be.emb_flush_window(lazy[i:i + 4], self.flush_windows, hyper, lag=1, max_blocks=self.flush_blocks)
    else:
      for i in range(0, len(lazy), 4):
        be.emb_flush_window(lazy[i:i + 4], self.flush_windows, hyper, lag=1, max_blocks=self.flush_blocks)
    self._window_pending = self._window_started = True

  def _join_window_flush(self):
    if self._window_pending:
      if self.device.type == 'cuda':
        torch.cuda.current_stream().wait_stream(self._flush_stream)
      self._window_pending = False

  # -- gradient clipping by global norm: the reduce and the row update as two steps, the norm in between
  def backward_reduce(self, normsq, weight):
    """First half of backward_update when gradients are clipped by their global norm (compat/optimizers.py:365-376):
    the de-duplicated row sums of every table group go to buffers instead of straight into the optimizer, and
    normsq[0] += weight * sum of their squares (weight = grad_scale^2: `values` of the IndexedSlices after the gradient
    multipliers).  apply_reduced() finishes the step once the multiplier is known."""
    be = kernels.hip()
    self.finish_group_grads()
    self._reduced = []
    for dim, grp in self.emb_groups.items():
      bufs = self._clip_bufs.get(dim) if hasattr(self, '_clip_bufs') else None
      if bufs is None:
        if not hasattr(self, '_clip_bufs'):
          self._clip_bufs = {}
        n = grp['num_entries']
        bufs = self._clip_bufs[dim] = (torch.zeros(n, dtype=torch.int32, device=self.device),
                                       torch.zeros(n, dim, dtype=torch.float32, device=self.device),
                                       torch.zeros(1, dtype=torch.int32, device=self.device))
      if self.lazy_decay:  # this step's er_emb_route (forward) left the sort and the key list
        lz = self._lazy[self._sort_leader.get(dim, dim)]
        keys, n_unique, grads = lz['ukeys'], lz['n_unique'], bufs[1]
        be.emb_bwd_reduce_routed(grp, grads)
      else:
        keys, grads, n_unique = be.emb_bwd_reduce(grp, out=bufs)
      ng = self._norm_group(dim)
      if ng is None:
        be.gradsq_rows(grads, dim, weight, normsq, True, counts=n_unique)
      else:
        # a table read by several lookups: TensorFlow's norm sees each lookup's IndexedSlices on its own (merged within
        # a lookup only: compat/optimizers.py:453-481) - a second reduction over keys that keep the lookups apart
        _, ngrads, nn = be.emb_bwd_reduce(ng['group'], out=ng['bufs'])
        be.gradsq_rows(ngrads, dim, weight, normsq, True, counts=nn)
      self._reduced.append((grp, keys, grads, n_unique))

  def _norm_group(self, dim):
    """None unless a table of this dim is read by more than one lookup; else a reduce-only twin of the table group whose
    lookups own disjoint key ranges (key bases spread), built once."""
    if not hasattr(self, '_norm_groups'):
      self._norm_groups = {}
    if dim not in self._norm_groups:
      specs = [s for s in self.fwd_specs if s.dim == dim]
      bases = [s.key_base for s in specs]
      ng = None
      if len(set(bases)) < len(bases):
        be = kernels.hip()
        spread, base = [], 0
        for s in specs:
          t = s.with_out(self._dout_of(s))
          t.key_base = base
          spread.append(t)
          base += s.rows
        st = self.storage[dim]
        grp = be.emb_group_create(spread, dim, base, st['var'], st['m'], st['v'], None)
        n = grp['num_entries']
        ng = {'group': grp, 'bufs': (torch.zeros(n, dtype=torch.int32, device=self.device),
                                     torch.zeros(n, dim, dtype=torch.float32, device=self.device),
                                     torch.zeros(1, dtype=torch.int32, device=self.device))}
      self._norm_groups[dim] = ng
    return self._norm_groups[dim]

  def apply_reduced(self, opt_kind, hyper):
    be = kernels.hip()
    for grp, keys, grads, n_unique in self._reduced:
      be.emb_apply_unique(grp, keys, grads, n_unique, opt_kind, hyper)
    self._reduced = []
    self._roll_flush(hyper)
    self._decay_pending = True

  # -- host exchange
  def _kv_filtered(self, name):
    t = self.tables[name]
    return t['kv_filter_freq'] > 1 or t['kv_steps_to_live'] > 0

  def evict_stale(self, global_step):
    """ev_params.steps_to_live (GlobalStepEvict): drop the ids whose last training lookup is more than steps_to_live
    steps before `global_step` and compact the arena (var / slots / last_step rows move down, the freed rows are zeroed
    like a fresh arena's).  Called when a checkpoint is written (utils/checkpoint.py save), as DeepRec evicts."""
    be = kernels.hip()
    evicted = {}
    for name, kv in self.kv_tables.items():
      stl = self.tables[name]['kv_steps_to_live']
      if stl <= 0:
        continue
      self.flush_decay()
      keys, rows, freq, version = be.kv_export_all(kv)
      keep = (int(global_step) - version.to(torch.int64)) <= stl
      if bool(keep.all()):
        continue
      evicted[name] = int((~keep).sum().item())
      old_used = int(rows.max().item()) + 1   # (arena rows in use: 0 .. old_used - 1)
      keys, rows, freq, version = keys[keep], rows[keep], freq[keep], version[keep]
      has_row = rows >= 0
      src = rows[has_row]
      order = torch.argsort(src)               # arena order is kept: row i of the compacted arena = i-th surviving row
      new_rows = torch.full_like(rows, -1)
      new_rows[has_row.nonzero().view(-1)[order]] = torch.arange(src.numel(), dtype=rows.dtype, device=rows.device)
      src = src[order].to(self.device)
      n = src.numel()
      t = self.tables[name]
      views = [(self.table_view(name), 0.0)] + [(self.slot_view(name, sl), self.slot_init[sl]) for sl in ('m', 'v')
                                                if self.slot_view(name, sl) is not None]
      # (lazy decay's last_step needs no move: after the flush above every row of the group carries the same stamp)
      for v, fresh in views:
        moved = v[src].clone()
        v[:n] = moved
        v[n:old_used].fill_(fresh)
      be.kv_rebuild(kv, keys, new_rows, freq, version)
    return evicted

  def state_dict(self, slots=False, rows_of=None):
    """rows_of: {table name: int64 ids}: only those rows of the named (dense) tables - and of their slots - are copied to
    the host, in the order given (a 200 M-row table's state is 153 GB; a parity check needs the rows its batches read)."""
    out = OrderedDict()
    self.flush_decay()
    self.check_kv_overflow()
    for name in self.tables:
      rows = None
      if rows_of is not None and name in rows_of and not self.tables[name]['kv']:
        rows = torch.as_tensor(rows_of[name], dtype=torch.int64)
      if self.tables[name]['kv']:
        # the materialised ids in ascending order and their rows (arena positions are run-dependent, keys are not)
        kv = self.kv_tables[name]
        t = self.tables[name]
        if self._kv_filtered(name):
          seen, seen_rows, freq, version = kernels.hip().kv_export_all(kv)
          has_row = seen_rows >= 0
          keys, rows = seen[has_row], seen_rows[has_row]
          # every id the table tracks (the counter filter's candidates included), its count (counting stops at
          # filter_freq) and the step of its last training lookup
          out[name + '/kv_seen_keys'] = seen.cpu().numpy().copy()
          out[name + '/kv_freq'] = np.minimum(freq.cpu().numpy(), max(t['kv_filter_freq'], 1)).astype(np.int32)
          out[name + '/kv_version'] = version.cpu().numpy().copy()
        else:
          keys, rows = kernels.hip().kv_export(kv)
        out[name + '/keys'] = keys.cpu().numpy().copy()
        out[name + '/kv_meta'] = np.array([kv['seed'], kv['mean'], kv['stddev'], kv['capacity'], t['kv_filter_freq'],
                                           t['kv_steps_to_live']], dtype=np.float64)
      pick = (lambda v: v) if rows is None else (lambda v: v[rows.to(v.device)])
      out[name] = pick(self.table_view(name).detach()).cpu().numpy().copy()
      if slots:
        for s in ('m', 'v'):
          sv = self.slot_view(name, s)
          if sv is not None:
            out[name + '/' + s] = pick(sv.detach()).cpu().numpy().copy()
    return out

  def load_kv_table(self, name, keys, values, slot_values, seen=None, freq=None, version=None):
    """The hash-table table becomes exactly the saved one: saved id i (keys ascending) owns arena row i, the ids the
    counter filter was still counting (`seen` minus `keys`) keep their counts, everything else is dropped.
    values / slot_values[s]: [len(keys), dim] rows in the order of `keys`."""
    kv = self.kv_tables[name]
    keys = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.int64))
    n = keys.numel()
    assert n <= kv['capacity'], 'hash-table embedding %s: %d saved ids, capacity %d' % (name, n, kv['capacity'])
    assert n < 2 or bool((keys[1:] > keys[:-1]).all()), 'hash-table embedding %s: saved ids must be ascending' % name
    rows = torch.arange(n, dtype=torch.int64)
    if seen is not None and self._kv_filtered(name):
      seen = torch.from_numpy(np.ascontiguousarray(seen, dtype=np.int64))
      pos = torch.searchsorted(seen, keys)   # (both ascending; every saved id is among the seen ones)
      assert bool((seen[pos.clamp(max=max(seen.numel() - 1, 0))] == keys).all()), name
      rows = torch.full((seen.numel(),), -1, dtype=torch.int64)
      rows[pos] = torch.arange(n, dtype=torch.int64)
      freq = torch.from_numpy(np.ascontiguousarray(freq, dtype=np.int32))
      version = torch.from_numpy(np.ascontiguousarray(version, dtype=np.int32))
      keys = seen
    else:
      version = None
      # (a checkpoint without the filter's state: a row means the id was admitted)
      freq = torch.full((n,), self.tables[name]['kv_filter_freq'], dtype=torch.int32) \
          if self.tables[name]['kv_filter_freq'] > 1 else None
    kernels.hip().kv_rebuild(kv, keys, rows, freq, version)
    self.check_kv_overflow()
    view = self.table_view(name)
    view.zero_()
    view[:n] = torch.as_tensor(np.asarray(values, dtype=np.float32)).to(self.device)
    for sl in ('m', 'v'):
      sv = self.slot_view(name, sl)
      if sv is not None:
        sv.fill_(self.slot_init[sl])
        if sl in slot_values:
          sv[:n] = torch.as_tensor(np.asarray(slot_values[sl], dtype=np.float32)).to(self.device)

  def load_state_dict(self, state):
    for name in self.tables:
      if name not in state:
        continue
      if self.tables[name]['kv']:
        values = state[name]
        slot_values = {sl: state[name + '/' + sl] for sl in ('m', 'v') if (name + '/' + sl) in state}
        self.load_kv_table(name, state[name + '/keys'], values, slot_values, state.get(name + '/kv_seen_keys'),
                           state.get(name + '/kv_freq'), state.get(name + '/kv_version'))
      else:
        self.table_view(name).copy_(torch.from_numpy(np.asarray(state[name], dtype=np.float32)).to(self.device))


def _stable_seed(name, base):
  import zlib
  return (zlib.crc32(name.encode('utf-8')) ^ (base * 2654435761)) & 0x7FFFFFFF


def _blocks_of(specs):
  n = 0
  for s in specs:
    V = 4 if s.dim % 4 == 0 else 1
    G = 1
    while G < s.dim // V:
      G <<= 1
    rpb = 256 // G
    n += (max(s.n_rows, 1) + rpb - 1) // rpb
  return n


class InputLayer(object):
  """Same constructor and call contract as the reference's InputLayer (layers/input_layer.py:33-70)."""

  def __init__(self, feature_configs, feature_groups_config, variational_dropout_config=None,
               wide_output_dim=-1, ev_params=None, embedding_regularizer=None, kernel_regularizer=None,
               is_training=False, is_predicting=False, engine=None):
    self._feature_groups = OrderedDict((x.group_name, FeatureGroup(x)) for x in feature_groups_config)
    self._group_name_to_seq_features = {
        x.group_name: x.sequence_features for x in feature_groups_config if len(x.sequence_features) > 0
    }
    self._fc_parser = FeatureColumnParser(feature_configs, self.get_wide_deep_dict(), wide_output_dim,
                                          ev_params=ev_params)
    self._embedding_regularizer = embedding_regularizer  # lambda (float) or None
    self._kernel_regularizer = kernel_regularizer
    self._is_training = is_training
    self._is_predicting = is_predicting
    if variational_dropout_config is not None:
      raise NotImplementedError('variational dropout is outside the hot-path scope')
    assert engine is not None, 'InputLayer needs the EmbeddingEngine of the model'
    self._engine = engine
    self._group_plan = {}  # group name -> dict(gkey, columns, cols, dims)
    self._seq_plan = {}    # group name -> plan of the un-combined (sequence + plain) form
    self._scope_count = 0
    # target attention over `sequence_features` declared inside feature groups (layers/input_layer.py:96-111)
    self._sequence_feature_layer = None
    if self._group_name_to_seq_features:
      from easyrec_amd.layers.sequence_feature_layer import SequenceFeatureLayer
      self._sequence_feature_layer = SequenceFeatureLayer(
          feature_configs, feature_groups_config, ev_params, embedding_regularizer, kernel_regularizer, is_training,
          is_predicting, engine=engine)

  @property
  def engine(self):
    return self._engine

  def has_group(self, group_name):
    return group_name in self._feature_groups

  def get_wide_deep_dict(self):
    """reference layers/input_layer.py:378-398."""
    d = {}
    for fg in self._feature_groups.values():
      for k, v in fg.wide_and_deep_dict.items():
        if k not in d:
          d[k] = v
        elif d[k] != v:
          d[k] = WideOrDeep.WIDE_AND_DEEP
    return d

  def _next_scope(self):
    s = 'input_layer' if self._scope_count == 0 else 'input_layer_%d' % self._scope_count
    self._scope_count += 1
    return s

  def _declare(self, features, group_name, plain_only=False):
    """First call for a group: create tables + lookups (TF would create the variables here).  plain_only: the
    group's non-sequence columns alone (`get_plain_feature`, the un-combined form)."""
    fg = self._feature_groups[group_name]
    columns, seq_columns = fg.select_columns(self._fc_parser)
    if seq_columns and not plain_only:
      raise NotImplementedError(
          'combining sequence columns over time inside a feature group (sequence_combiner attention / text_cnn) is '
          'outside the hot-path scope; use sequence_features / seq_att_groups (DIN), an input_layer block with '
          'output_seq_and_normal_feature, or TagFeature')
    scope = self._next_scope()
    eng = self._engine
    B = eng.batch_size
    width = sum(c.dimension for c in columns)
    gkey = 'group:' + group_name + (':plain' if plain_only else '')
    eng.declare_group(gkey, width, self._embedding_regularizer)
    col, cols, dims, numeric = 0, [], [], []
    for c in columns:
      if isinstance(c, NumericColumn):
        numeric.append((c, col))
      elif isinstance(c, EmbeddingColumn):
        declare_lookup(eng, features, c, scope, gkey, col, B)
      else:
        raise NotImplementedError('column type %s' % type(c).__name__)
      cols.append(col)
      dims.append(c.dimension)
      col += c.dimension
    self._group_plan[(group_name, plain_only)] = {'gkey': gkey, 'columns': columns, 'cols': cols, 'dims': dims,
                                                  'numeric': numeric}

  def _declare_sequences(self, features, group_name):
    """The sequence columns of a group kept over time (`get_sequence_feature`, layers/input_layer.py:164-200): one
    lookup per column into ONE [B * L, sum(dim)] output; tables `input_layer/<categorical column>/embedding_weights`."""
    fg = self._feature_groups[group_name]
    _, seq_columns = fg.select_columns(self._fc_parser)
    eng = self._engine
    B = eng.batch_size
    assert all(isinstance(c, EmbeddingColumn) for c in seq_columns), 'sequence raw features: outside the hot-path scope'
    lens = {features.seqs[c.raw_name]['ids'].shape[1] for c in seq_columns}
    assert len(lens) == 1, 'the sequence features of group %s differ in max_seq_len' % group_name
    L = lens.pop()
    gkey = 'group:' + group_name + ':seq'
    eng.declare_seq_output(gkey, B * L, sum(c.dimension for c in seq_columns), self._embedding_regularizer)
    col, cols = 0, []
    for c in seq_columns:
      declare_lookup(eng, features, c, 'input_layer', gkey, col, B * L, seq=True,
                     table_name='input_layer/%s/embedding_weights' % c.categorical_column.name)
      cols.append((col, c.dimension, c.raw_name))
      col += c.dimension
    self._seq_plan[group_name] = {'gkey': gkey, 'cols': cols, 'L': L, 'width': col}

  def _uncombined(self, features, group_name):
    """is_combine=False (layers/input_layer.py:268-278): ([(embedding [B, L, dim], length [B]) per sequence column],
    the plain columns' concat (None when the group has none), their per-feature list).  L = the loaded batch's longest
    sequence."""
    eng = self._engine
    fg = self._feature_groups[group_name]
    columns, _ = fg.select_columns(self._fc_parser)
    if group_name not in self._seq_plan:
      self._declare_sequences(features, group_name)  # (reference order: sequence features first, then the plain ones)
    sp = self._seq_plan[group_name]
    B, L = eng.batch_size, sp['L']
    if not eng.finalized:
      seq_out = eng.groups[sp['gkey']]['out']
    else:
      eng.forward(features.version)
      seq_out = eng.group_tensor(sp['gkey'], requires_grad=self._is_training)
    seq3 = seq_out.view(B, L, sp['width'])
    seq_features = []
    for c0, d, name in sp['cols']:
      Lm = features.seq_pad_len(name)
      seq_features.append((seq3[:, :Lm, c0:c0 + d], features.seqs[name]['len']))
    if not columns:
      return seq_features, None, []
    plain, plain_list = self._combined(features, group_name, plain_only=True)
    return seq_features, plain, list(plain_list)

  def __call__(self, features, group_name, is_combine=True, is_dict=False):
    assert group_name in self._feature_groups, 'invalid group_name[%s], list: %s' % (
        group_name, ','.join(self._feature_groups))
    if not is_combine:
      return self._uncombined(features, group_name)
    out, flist, name_to_out = self._combined(features, group_name, is_dict=True)
    if group_name in self._group_name_to_seq_features:
      # target attention over the group's sequence_features; keys are the group's own outputs where it has them
      seq_cfgs = self._group_name_to_seq_features[group_name]
      assert not self._feature_groups[group_name].config.HasField('negative_sampler') or \
          not self._feature_groups[group_name].config.negative_sampler, 'negative sampler: outside the hot-path scope'
      _, all_seq_fea = self._sequence_feature_layer(features, out, seq_cfgs, dict(name_to_out), scope_name=group_name)
      views = list(flist) + list(all_seq_fea)
      for cfg, fea in zip(seq_cfgs, all_seq_fea):
        name_to_out['seq_fea/' + cfg.group_name] = fea
      out = kernels.concat_cols([out] + list(all_seq_fea))
      flist = FeatureList(views)
    if is_dict:
      return out, flist, name_to_out
    return out, flist

  def _combined(self, features, group_name, is_dict=False, plain_only=False):
    eng = self._engine
    if (group_name, plain_only) not in self._group_plan:
      self._declare(features, group_name, plain_only)
    plan = self._group_plan[(group_name, plain_only)]
    if not eng.finalized:
      # build pass: shapes only (tables are allocated by engine.finalize() after all groups exist)
      out = eng.groups[plan['gkey']]['out']
    else:
      eng.forward(features.version)
      g = eng.groups[plan['gkey']]
      for c, col in plan['numeric']:
        src = features.raw(c.key)
        src2 = src.view(-1, 1) if src.dim() == 1 else src
        kernels.hip().axpy2d(src2, 1.0, g['out'][:, col:col + c.dimension], accumulate=False)
      out = eng.group_tensor(plan['gkey'], requires_grad=self._is_training)
    views = [out[:, c0:c0 + d] for c0, d in zip(plan['cols'], plan['dims'])]
    flist = FeatureList(views, base=out, col0=0, dims=plan['dims'])
    if is_dict:
      return out, flist, {c.raw_name: v for c, v in zip(plan['columns'], views)}
    return out, flist


def declare_lookup(eng, features, column, scope, gkey, col, n_out_rows, seq=False, table_name=None):
  """Create the table (or reuse a shared one) and the lookup spec of one embedding column."""
  cat = column.categorical_column
  rows = cat.num_buckets
  if column.shared_name:
    table_name = '%s/embedding_weights' % column.var_scope_name  # shared across scopes/columns
  elif table_name is None:
    table_name = '%s/%s/embedding_weights' % (scope, column.var_scope_name)
  ev = getattr(column, 'ev_params', None)
  kv_capacity = None
  if ev is not None:
    assert cat.kind == 'hash' and not column.shared_name, \
        'ev_params on %s: hash-table embeddings cover hashed Id-, Tag- and SequenceFeatures with a table of their own' % column.raw_name
    kv_capacity = int(ev.max_capacity) if ev.HasField('max_capacity') else int(os.environ.get('EASYREC_AMD_KV_CAPACITY', 1 << 22))
  eng.declare_table(table_name, rows, column.dimension, column.initializer, kv_capacity=kv_capacity,
                    kv_filter_freq=int(ev.filter_freq) if ev is not None else 0,
                    kv_steps_to_live=int(ev.steps_to_live) if ev is not None else 0)
  fname = column.raw_name
  schema = features.schema
  B = features.batch_size
  if cat.weight_key and cat.key.endswith('_raw_proj_id'):
    # RawFeature projection: ids 0..k-1 weighted by the normalised values (input.py:648-673)
    if fname in schema.raw_multi:
      rm = features.raw_multi[fname]
      eng.add_lookup(gkey, table_name, rm['ids'], rm['offsets'], rm['values'].view(-1), col, column.combiner, B,
                     rm['ids'].numel(), fname)
    else:
      eng.add_lookup(gkey, table_name, features.zero_ids, None, features.raw(fname), col, column.combiner, B, B,
                     fname)
    return
  if seq:
    s = features.seqs[fname]
    L = s['ids'].shape[1]
    seq_ids = s['ids'].view(-1)
    if kv_capacity is not None:
      # every position of the [B, L] id buffer is translated each step (padding is -1: no row, a zero embedding)
      rows_buf = torch.full_like(seq_ids, -1)
      eng.kv_jobs.append((table_name, seq_ids, rows_buf))
      seq_ids = rows_buf
    eng.add_lookup(gkey, table_name, seq_ids, None, None, col, 'sum', B * L, B * L, fname)
    return
  if fname in schema.tags:
    t = features.tags[fname]
    tag_ids = t['ids']
    if kv_capacity is not None:
      # the buffer holds offsets[B] ids of this step (the tail is stale): only those are translated
      rows_buf = torch.full_like(tag_ids, -1)
      eng.kv_jobs.append((table_name, tag_ids, rows_buf, t['offsets'][B:B + 1]))
      tag_ids = rows_buf
    eng.add_lookup(gkey, table_name, tag_ids, t['offsets'], t['weights'], col, column.combiner, B,
                   t['ids'].numel(), fname)
    return
  if fname in schema.seqs:
    # sequence feature used as a plain (combined) column: sum/mean over the time axis
    s = features.seqs[fname]
    raise NotImplementedError('sequence feature %s as combined column' % fname)
  ids = features.ids_of(fname)
  if kv_capacity is not None:
    assert fname in schema.hash_single, 'ev_params on %s: hash-table embeddings cover hashed IdFeatures and TagFeatures' % fname
    rows_buf = torch.full((B,), -1, dtype=torch.int64, device=eng.device)  # arena rows of this step's ids
    eng.kv_jobs.append((table_name, ids, rows_buf))
    ids = rows_buf
  eng.add_lookup(gkey, table_name, ids, None, None, col, column.combiner, B, B, fname)

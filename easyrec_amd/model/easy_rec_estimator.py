"""The training step: what `EasyRecEstimator._train_model_fn` + `optimize_loss` do per batch.

Behaviour spec: reference easy_rec/python/model/easy_rec_estimator.py:155-353 and
compat/optimizers.py:280-450:
  total_loss = sum(loss_dict) + add_n(REGULARIZATION_LOSSES)           (estimator :166-184)
    REGULARIZATION_LOSSES = embedding-output L2 (layers/input_layer.py:369-375)
                          + l2 * 0.5*||kernel||^2 per regularised dense kernel (layers/dnn.py:57-62)
  optimizer from train_config.optimizer_config (builders/optimizer_builder.py:28-144); with two
    configs the first drives the embeddings, the second the dense variables (easy_rec_model.py:446-467)
  embedding_learning_rate_multiplier -> gradient multiplier on embedding tables (estimator :308-317)
  global step increments after the apply; the LR schedule sees the pre-increment step.

The TF Estimator machinery (sessions, hooks, savers) is replaced by a plain Python loop around HIP
launches; the whole step (hash -> lookup -> interactions/MLP -> loss -> backward -> sparse + dense
optimizer) can be captured into one hipGraph (`capture()`), because every buffer has a fixed address
and per-step scalars are read from device memory.
"""
import logging
import os
from collections import OrderedDict

import numpy as np
import torch

from easyrec_amd import kernels
from easyrec_amd.builders import optimizer_builder
from easyrec_amd.core import context
from easyrec_amd.core.variables import VarStore
from easyrec_amd.input.features import DeviceFeatures, FeatureSchema
from easyrec_amd.layers.input_layer import EmbeddingEngine
from easyrec_amd.model.easy_rec_model import EasyRecModel
from easyrec_amd.utils import config_util
from easyrec_amd.utils.load_class import import_all_models


class EasyRecEstimator(object):

  HYPER_SLOTS = 4096

  def __init__(self, pipeline_config, device='cuda', batch_size=None, seed=0, schema_kwargs=None,
               is_training=True, overlap_sweep=False, dense_dtype=None, dense_sweep=None):
    import_all_models()
    self.pipeline_config = config_util.get_configs_from_pipeline_file(pipeline_config) \
        if isinstance(pipeline_config, str) else pipeline_config
    self.device = torch.device(device)
    self.seed = seed
    self.is_training = is_training
    # (dense weight gradients + dense optimizer on a second stream alongside the embedding backward were measured in rounds
    # 2 and 4 - 0.609 against 0.550 ms, 0.479 against 0.447 - and removed in round 5: concurrent kernels stretch the
    # latency-bound ones more than they hide, profiles/r04_s1_lines_summary.txt)
    # the id hash as workgroups of the step prologue's launch (one launch less per step) - A/B switch
    self.fused_front = os.environ.get('EASYREC_AMD_FUSED_FRONT', '1') != '0'
    # TF-exact Adam: True = run the dense-decay sweep of the untouched rows on a second stream (see
    # EmbeddingEngine.start_decay_sweep); False (default) = sequential sweep inside er_emb_bwd_update.
    # Measured on MI355X (profiles/r01_overlap_knob.txt): the overlapped step is SLOWER (3.8 ms vs
    # 2.74 ms) - the HBM-saturating sweep stretches the latency-bound forward/backward kernels more
    # than it hides - so the sequential order is the default.
    self.overlap_sweep = overlap_sweep and torch.device(device).type == 'cuda'
    cfg = self.pipeline_config
    self.feature_configs = config_util.get_compatible_feature_configs(cfg)
    if cfg.model_config.HasField('ev_params'):
      # model-level ev_params apply to every feature (model/easy_rec_model.py:64-66): written into the hashed
      # IdFeatures' own configs so that the schema, the generators and the oracle all read one place
      for fc in self.feature_configs:
        if fc.feature_type in (fc.IdFeature, fc.TagFeature) and fc.HasField('hash_bucket_size') and \
            fc.hash_bucket_size > 0 and not fc.HasField('ev_params'):
          fc.ev_params.CopyFrom(cfg.model_config.ev_params)
    self.schema = FeatureSchema(cfg.data_config, self.feature_configs, batch_size=batch_size,
                                **(schema_kwargs or {}))
    self.batch_size = self.schema.batch_size
    self.features = DeviceFeatures(self.schema, self.device)
    self.varstore = VarStore(self.device, seed=seed)
    self.engine = self._make_engine()
    self.engine.train_mode = bool(is_training)
    self.ctx = context.ModelContext(self.varstore, self.engine, is_training=is_training)
    # 'f32' (default: exact-fp32 MFMA, the 1e-4 parity bar) or 'bf16' (BASELINE config 3: bf16 dense, fp32 embeddings)
    self.ctx.dense_dtype = dense_dtype or os.environ.get('ER_DENSE_DTYPE', 'f32')
    assert self.ctx.dense_dtype in ('f32', 'bf16')
    self.global_step = 0
    self.graph = None
    self._graph_signature = None
    self._built = False

    # optimizers (estimator :216-235)
    oc = cfg.train_config.optimizer_config
    assert 1 <= len(oc) <= 2, 'one optimizer, or two (embedding, dense)'
    self.opt_emb = optimizer_builder.build(oc[0])
    self.opt_dense = optimizer_builder.build(oc[1]) if len(oc) == 2 else self.opt_emb
    self.emb_grad_scale = 1.0
    if oc[0].HasField('embedding_learning_rate_multiplier'):
      self.emb_grad_scale = float(oc[0].embedding_learning_rate_multiplier)
    # gradient clipping by global norm (estimator :339-353 -> optimize_loss(clip_gradients=...), compat/optimizers.py:
    # 365-376): 0 = off.  The norm covers every gradient after the multipliers, so the embedding backward runs as
    # reduce -> norm -> apply instead of the fused reduce+apply (layers/input_layer.py backward_reduce).
    self.clip_norm = float(cfg.train_config.gradient_clipping_by_norm) \
        if cfg.train_config.gradient_clipping_by_norm > 0 else 0.0
    if self.clip_norm > 0 or self.overlap_sweep:
      self.engine.allow_fused = False  # (the norm sits between reduce and apply; the overlapped sweep marks rows first)

    labels = OrderedDict((name, self.features.label(name)) for name in self.schema.label_fields)
    with context.use(self.ctx):
      model_cls = EasyRecModel.create_class(cfg.model_config.model_class)
      self.model = model_cls(cfg.model_config, self.feature_configs, self.features, labels,
                             is_training=is_training)

    dev = self.device
    # per-step optimizer scalars: precomputed by the host for the next HYPER_SLOTS steps, selected on
    # the device by a device-resident step counter (no host-written memory inside the captured graph)
    self.hyper = torch.zeros(2, kernels.HYPER_FLOATS, dtype=torch.float32, device=dev)  # [emb, dense]
    self.hyper_table = torch.zeros(self.HYPER_SLOTS, 2, kernels.HYPER_FLOATS, dtype=torch.float32, device=dev)
    self.step_counter = torch.zeros(1, dtype=torch.int64, device=dev)
    self._planned_until = 0
    # TF-exact Adam (`adam_optimizer`): by default the dense decay of untouched rows is applied lazily, bit-
    # identically (er_emb_catch_up); dense_sweep=True (or ER_DENSE_SWEEP=1) streams every row every step instead
    if dense_sweep is None:
      dense_sweep = os.environ.get('ER_DENSE_SWEEP', '0') == '1'
    self.dense_sweep = bool(dense_sweep)
    # per-step lr_t history read by the lazy dense decay's replay: step s lives at lr_hist[s]; train_step / set_global_step
    # refuse to run past its capacity (er_hyper_select stops recording there and a replay would read out of bounds)
    # layout [2 * capacity]: lr_t per step | running maximum of lr_t (the replay's absorbed regime bounds with it)
    n_hist = max(2 * int(cfg.train_config.num_steps or 0), 1 << 20)
    self.lr_hist = torch.zeros(2 * n_hist, dtype=torch.float32, device=dev)
    self.engine.set_step_clock(self.step_counter, self.lr_hist, self.hyper[0],
                               lazy_decay=not self.dense_sweep and not self.overlap_sweep)
    # the replay of the decay-only steps: closed form (csrc/er_decay.h; default) or the bit-exact step-by-step replay
    # with its rolling flush (EASYREC_AMD_EXACT_DECAY=1, and whenever the betas are outside the closed form's range)
    self.decay_tables = None
    if self.engine.lazy_decay and self.opt_emb.kind == kernels.OPT_ADAM and \
        os.environ.get('EASYREC_AMD_EXACT_DECAY', '0') != '1':
      self.decay_tables = kernels.hip().decay_tables_create(self.lr_hist, self.step_counter, self.opt_emb.beta1,
                                                            self.opt_emb.beta2)
      self.engine.set_decay_tables(self.decay_tables)
    self.losses = {
        'regularization_loss': torch.zeros(1, dtype=torch.float32, device=dev),
        'total_loss': torch.zeros(1, dtype=torch.float32, device=dev),
    }
    self._reg_emb = torch.zeros(1, dtype=torch.float32, device=dev)
    self._reg_dense = torch.zeros(1, dtype=torch.float32, device=dev)
    self._normsq = torch.zeros(1, dtype=torch.float32, device=dev)   # clipping: sum of squares of all gradients
    self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)  # clipping: the step's global gradient norm

  # -- hooks overridden by the embedding-parallel estimator (model/embedding_parallel.py)
  def _make_engine(self):
    return EmbeddingEngine(self.device, self.batch_size, seed=self.seed)

  def _dense_grad_scale(self):
    return 1.0

  def _sync_dense_grads(self):
    """Multi-GPU: all-reduce of the flat dense gradient buffer.  Single GPU: nothing."""

  # -- construction
  def build(self):
    """Build pass (creates variables / declares tables), then allocate + pack everything."""
    assert not self._built
    kernels.hip().reserve_scratch(1 << 22)
    kernels.hip().gemm_reserve(1 << 24)  # split-K workspace (64 MB): sized before any hipGraph capture
    if self.overlap_sweep and self.opt_emb.kind == kernels.OPT_ADAM:
      kernels.hip().config_set('sweep_blocks_per_cu', int(os.environ.get('ER_SWEEP_BLOCKS_PER_CU', '3')))
    self.ctx.building = True
    with context.use(self.ctx):
      self.model.begin_step()
      with torch.no_grad():
        self.model.build_predict_graph()
    self.ctx.building = False
    self.engine.finalize(self.opt_emb.kind)
    if self.opt_emb.kind == kernels.OPT_ADAGRAD:
      # (also what a freed / never-owned arena row of a hash-table table goes back to: EmbeddingEngine.slot_init)
      self.engine.slot_init['v'] = float(getattr(self.opt_emb, 'initial_accumulator_value', 0.1))
      for st in self.engine.storage.values():
        st['v'].fill_(self.engine.slot_init['v'])
    self.varstore.pack(self._extra_grad_floats())
    if self.ctx.dense_dtype == 'bf16' and self.device.type == 'cuda':
      self._bf16 = kernels.hip().bf16_enable(self.varstore)  # bf16 weight shadows for er_gemm_bf16_nt
      self.ctx.bf16_state = self._bf16  # (+ the step's registry of producer-written bf16 operands: kernels.Bf16Shadows)
    self._after_pack()
    if self.opt_dense.kind in (kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM):
      self.varstore.slot('m')
      self.varstore.slot('v')
    elif self.opt_dense.kind == kernels.OPT_ADAGRAD:
      self.varstore.slot('v').fill_(getattr(self.opt_dense, 'initial_accumulator_value', 0.1))
    self._built = True
    return self

  def _extra_grad_floats(self):
    return 0

  def _after_pack(self):
    pass

  # -- one step
  def _plan_hyper(self, count):
    """Generate the scalars of the next `count` steps and upload them to their table slots."""
    rows = np.zeros((count, 2, kernels.HYPER_FLOATS), dtype=np.float32)
    for i in range(count):
      step = self._planned_until + i
      rows[i, 0] = self.opt_emb.hyper_row(step, self.emb_grad_scale)
      rows[i, 1] = self.opt_dense.hyper_row(step, self._dense_grad_scale())
      self.opt_emb.finish_step()
      if self.opt_dense is not self.opt_emb:
        self.opt_dense.finish_step()
    slots = (self._planned_until + np.arange(count)) % self.HYPER_SLOTS
    if self.device.type == 'cuda':
      # asynchronous upload from pinned staging buffers: the host never waits for the stream (a pageable .to(device) would
      # drain it - with the 20 ms of Python above, one step in 2048 took 19 ms and the steady-state mean carried 9 us of it)
      if getattr(self, '_hyper_stage', None) is None:
        self._hyper_stage = (torch.empty(self.HYPER_SLOTS, 2, kernels.HYPER_FLOATS, dtype=torch.float32).pin_memory(),
                             torch.empty(self.HYPER_SLOTS, dtype=torch.int64).pin_memory())
        self._hyper_staged = None
      if self._hyper_staged is not None:
        self._hyper_staged.synchronize()  # (the previous upload out of these buffers: half a ring of steps ago)
      srows, sslots = self._hyper_stage
      srows[:count].copy_(torch.from_numpy(rows))
      sslots[:count].copy_(torch.from_numpy(slots))
      self.hyper_table[sslots[:count].to(self.device, non_blocking=True)] = srows[:count].to(self.device, non_blocking=True)
      self._hyper_staged = torch.cuda.Event()
      self._hyper_staged.record()
    else:
      self.hyper_table[torch.from_numpy(slots).to(self.device)] = torch.from_numpy(rows).to(self.device)
    self._planned_until += count

  def _grow_lr_history(self, need, flush=True):
    """Make room for step indices < need in the lr_t history (lazy dense decay).  Outside a captured graph the buffer
    is re-allocated (doubling) and re-registered with the table groups; inside one its address is baked in, so every
    row is brought current first (nothing older than the flush is ever replayed) - then the run must stop.
    flush=False (checkpoint restore): the tables were just loaded and are current by definition - replaying whatever
    decay the PREVIOUS run of this estimator left pending would corrupt the restored rows."""
    cap = self.lr_hist.numel() // 2
    if need <= cap:
      return
    if self.graph is not None or getattr(self, '_graphs', None) is not None:
      raise RuntimeError('easyrec_amd: step %d exceeds the lr_t history capacity %d baked into the captured hipGraph; '
                         'set train_config.num_steps (the history is sized 2x num_steps) before capture()' % (need, cap))
    if self.device.type == 'cuda':
      torch.cuda.synchronize()
    new_cap = max(2 * cap, need)
    grown = torch.zeros(2 * new_cap, dtype=torch.float32, device=self.device)
    grown[:cap].copy_(self.lr_hist[:cap])
    grown[new_cap:new_cap + cap].copy_(self.lr_hist[cap:])
    if self.decay_tables is not None:
      # the per-step table is indexed like the history: every row is brought current first (entries older than a row's
      # last update are never read), then the tables restart on the grown history
      if flush:
        self.engine.flush_decay()
      torch.cuda.synchronize() if self.device.type == 'cuda' else None
      kernels.hip().decay_tables_destroy(self.decay_tables)
      self.decay_tables = kernels.hip().decay_tables_create(grown, self.step_counter, self.opt_emb.beta1,
                                                            self.opt_emb.beta2)
    self.lr_hist = grown
    self.engine.rebind_lr_history(grown, decay_tables=self.decay_tables)

  def _refresh_hyper(self):
    """Keep the device table half a ring ahead of `global_step` (sync only once per half ring)."""
    self._grow_lr_history(self.global_step + 1)
    half = self.HYPER_SLOTS // 2
    if self._planned_until == 0:
      self._plan_hyper(self.HYPER_SLOTS)
    elif self.global_step + half >= self._planned_until:
      # no synchronisation: the upload is ordered on the step's stream behind every step already enqueued, and it rewrites
      # only the half of the ring those steps have consumed (the next half a ring of steps reads the other one)
      self._plan_hyper(half)

  def _device_step(self):
    """Everything that runs on the GPU for one batch (graph-capturable)."""
    be = kernels.hip()
    if hasattr(be, 'discard_loss_tail'):
      be.discard_loss_tail()  # (a step that raised between its loss tail and its tail launch must not poison this one)
    # prologue, one launch: this step's optimizer scalars (device-side step counter) + the flat gradient buffer zeroed
    hash_job = self.features.hash_job() if self.fused_front else None
    be.step_prologue(self.hyper_table, self.step_counter, self.hyper, history=self.lr_hist,
                     zero=self.varstore.flat_grad_all, decay_tables=self.decay_tables, hash_job=hash_job)
    self.features.transform(hashed=hash_job is not None)
    if self.is_training and self.overlap_sweep and self.opt_emb.kind == kernels.OPT_ADAM:
      self.engine.start_decay_sweep(self.hyper[0])
    with context.use(self.ctx):
      self.model.begin_step()
      self.model.build_predict_graph()
      loss_dict = self.model.build_loss_graph()
      # the step's tail (single GPU, no clipping): the weight gradients stay queued and are contracted in the grid of the
      # embedding row update, the scalar loss tail rides in that grid and the dense optimizer in the next one
      # (er_emb_bwd_fused_tail) - nothing between here and the end of the step reads what they write
      tail = self.is_training and self.clip_norm <= 0 and bool(getattr(be, 'fused_tail', False)) and self._tail_fusable()
      riders = tail and bool(getattr(be, 'tail_riders', False))
      self._loss_tail(loss_dict, defer=riders)
      if self.is_training:
        vs = self.varstore
        if self.clip_norm > 0:
          self.model.backward()
          self._sync_dense_grads()
          self._clipped_update()
        else:
          self.model.backward(flush=not tail)
          self._sync_dense_grads()
          opt = (vs.flat, vs.slots.get('m'), vs.slots.get('v'), vs.flat_grad, vs.l2coef if vs.any_l2 else None,
                 self.opt_dense.kind, self.hyper[1], vs.l2_partials)
          done = False
          if tail:
            done = self.engine.backward_update(self.opt_emb.kind, self.hyper[0], pending_wgrads=True,
                                               dense_opt=opt if riders else None)
          else:
            self.engine.backward_update(self.opt_emb.kind, self.hyper[0])
          if riders:
            be.flush_loss_tail()  # (a tail that did not take it)
          if not done:
            be.dense_opt_step(*opt[:7], l2_partials=opt[7])

  def _tail_fusable(self):
    """The queued weight gradients may wait for the embedding backward: one process (no dense all-reduce reads them
    first) and the single-GPU engine, whose backward_update takes them."""
    return type(self.engine) is EmbeddingEngine and type(self)._sync_dense_grads is EasyRecEstimator._sync_dense_grads \
        and self._dense_grad_scale() == 1.0

  def _emb_gradsq_weight(self):
    """What the squared embedding row sums are multiplied by in the norm: grad_scale^2 (fp32, as the kernels apply it)."""
    gs = np.float32(self.emb_grad_scale)
    return float(gs * gs)

  def _clipped_update(self):
    """norm over (dense gradients as the optimizer sees them, de-duplicated embedding row sums) -> multiplier in the
    step's er_opt_hyper records -> both optimizers."""
    be, vs = kernels.hip(), self.varstore
    l2 = vs.l2coef if vs.any_l2 else None
    be.gradsq_dense(vs.flat, vs.flat_grad, l2, self.hyper[1], self._normsq, accumulate=False)
    self.engine.backward_reduce(self._normsq, self._emb_gradsq_weight())
    be.clip_scale(self._normsq, self.clip_norm, self.hyper, self.grad_norm)
    self.engine.apply_reduced(self.opt_emb.kind, self.hyper[0])
    be.dense_opt_step(vs.flat, vs.slots.get('m'), vs.slots.get('v'), vs.flat_grad, l2, self.opt_dense.kind, self.hyper[1],
                      l2_partials=vs.l2_partials)

  def _loss_tail(self, loss_dict, defer=False):
    """regularization_loss = embedding-output L2 + kernel L2, total_loss = that + the task losses (estimator :166-184);
    one launch."""
    be, eng, vs = kernels.hip(), self.engine, self.varstore
    names = list(loss_dict.keys())
    for name in names:
      if name not in self.losses:
        self.losses[name] = torch.zeros(1, dtype=torch.float32, device=self.device)
    partials = eng.sumsq[:eng.reg_blocks] if (eng.reg_lambda > 0 and eng.reg_blocks > 0) else None
    kernels.materialize_pending_heads()  # (a logit head that no loss consumed: its logits are still owed)
    jobs = list(self.ctx.tail_jobs)
    del self.ctx.tail_jobs[:]
    values = [loss_dict[n] for n in names]
    if jobs or any(getattr(v, '_er_partials', None) is not None for v in values):
      # a fused head (builders/loss_builder.py): its loss arrives as per-workgroup partial sums, its dW / db as column-sum jobs
      be.loss_tail(partials, 0.5 * eng.reg_lambda, vs.l2_partials if vs.any_l2 else None, values,
                   [self.losses[n] for n in names], self.losses['regularization_loss'], self.losses['total_loss'], jobs=jobs,
                   defer=defer)
      return
    be.reg_total_loss(partials, 0.5 * eng.reg_lambda, vs.l2_partials if vs.any_l2 else None,
                      [v.reshape(1) for v in values], [self.losses[n] for n in names],
                      self.losses['regularization_loss'], self.losses['total_loss'])

  def train_step(self, batch=None):
    """Load `batch` (optional) and run one optimisation step.  Returns the dict of loss tensors."""
    assert self._built, 'call build() first'
    if batch is not None:
      self.features.load(batch)
    else:
      self.features.version += 1
    self._refresh_hyper()
    if self.graph is not None and self.features.shape_signature() == self._graph_signature:
      self.graph.replay()
    else:
      # eager: no graph yet, or this batch's sequences are padded to another length than the captured one's (the
      # reference pads to the batch's longest sequence: tensor shapes follow the batch)
      self._device_step()
    self.global_step += 1
    # hash-table embeddings: an arena that ran out of rows serves zeros for every new id from then on (a sticky device
    # flag): a periodic blocking read bounds how long that can go unnoticed; evaluate(), state_dict() and
    # checkpoint.save() check it too
    if self.engine.kv_tables and self.global_step % self.OVERFLOW_CHECK_EVERY == 0:
      self.engine.check_overflow()
    return self.losses

  OVERFLOW_CHECK_EVERY = 256

  def predict(self, batch=None):
    assert self._built
    if batch is not None:
      self.features.load(batch)
    self.features.transform()
    # no row update follows these lookups: bring the tables current once, then look up without the catch-up
    # (running it here would re-apply the pending Adam decay on every call)
    self.engine.begin_inference()
    try:
      with context.use(self.ctx), torch.no_grad():
        self.model.begin_step()
        return self.model.build_predict_graph()
    finally:
      self.engine.end_inference()

  def evaluate(self, batches, eval_config=None):
    """The evaluation pass of the reference (`EasyRecEstimator._eval_model_fn` -> `build_metric_graph`,
    model/easy_rec_estimator.py:355-420, model/rank_model.py:334-470) for `metrics_set { auc | gauc | session_auc |
    max_f1 }`.  The model runs with is_training=False (BatchNorm moving statistics, no dropout) over `batches` (host
    batches when a grouped AUC needs its key column); returns {'<metric>'[+ '_<tower>']: value}."""
    from easyrec_amd.core import metrics as metrics_lib
    assert self._built
    if getattr(self.engine, 'kv_tables', None):
      self.engine.check_overflow()
    ec = eval_config if eval_config is not None else self.pipeline_config.eval_config
    from easyrec_amd.input.features import host_key_column
    def specs_of(metrics_set):  # [(output name, metric kind, argument)]
      specs = []
      for m in metrics_set:
        kind = m.WhichOneof('metric')
        if kind == 'auc':
          specs.append(('auc', 'auc', int(m.auc.num_thresholds)))
        elif kind == 'gauc':  # rank_model.py:376-399
          specs.append(('gauc', 'grouped', (m.gauc.uid_field, m.gauc.reduction)))
        elif kind == 'session_auc':  # rank_model.py:401-420
          specs.append(('session_auc', 'grouped', (m.session_auc.session_id_field, m.session_auc.reduction)))
        elif kind == 'max_f1':
          specs.append(('max_f1', 'max_f1', None))
        else:
          raise NotImplementedError('metric %s is outside the hot-path scope (auc, gauc, session_auc, max_f1)' % kind)
      return specs

    towers = getattr(self.model, '_label_name_dict', None)
    if not towers:
      # a rank model: eval_config.metrics_set (rank_model.py build_metric_graph); auc when it is empty (eval.proto)
      heads = [('', self.model._label_name)]
      head_specs = {'': specs_of(ec.metrics_set) or [('auc', 'auc', 200)]}
    else:
      # a multi-task model: every tower's OWN metrics_set (multi_task_model.py:143-158); a passed eval_config overrides
      heads = [('_' + t, l) for t, l in towers.items()]
      by_name = {t.name: t.config.metrics_set for t in self.model._towers}
      head_specs = {'_' + t: specs_of(ec.metrics_set if eval_config is not None else by_name[t]) for t in towers}

    def make(kind, arg):
      if kind == 'auc':
        return metrics_lib.AUC(arg, self.device)
      if kind == 'grouped':
        return metrics_lib.DeviceSeparatedAUC(arg[1], self.device)
      return metrics_lib.DeviceMaxF1(200, self.device)

    acc = {}
    for suf, _ in heads:
      for name, kind, arg in head_specs[suf]:
        acc.setdefault((name, suf), (kind, arg, make(kind, arg)))  # (a metric named twice keeps its first settings)
    was = (self.model._is_training, self.ctx.is_training)
    self.model._is_training, self.ctx.is_training = False, False
    try:
      for batch in batches:
        pred = self.predict(batch)
        for (name, suf), (kind, arg, m) in acc.items():
          label = self.features.label(dict(heads)[suf])
          if label.dtype.is_floating_point:  # tf.to_int64(label) in front of the metrics (rank_model.py:352): truncation
            label = torch.trunc(label)
          if kind == 'auc':  # metrics_tf.auc(label, probs, num_thresholds): no weights (rank_model.py:362)
            m.update(label, pred['probs' + suf], None)
          elif kind == 'grouped':  # labels / predictions stay on the device; only the key column comes from the host batch
            m.update(label, pred['probs' + suf], host_key_column(self.features.schema, batch, arg[0]))
          else:
            # max_f1 is fed the LOGITS (rank_model.py:424-427: `metrics_lib.max_f1(label, prediction_dict['logits'])`),
            # against its 200 thresholds in [0, 1] - the reference's choice, kept so that the numbers agree
            m.update(label, pred['logits' + suf])
    finally:
      self.model._is_training, self.ctx.is_training = was
    return {name + suf: m.result() for (name, suf), (_, _, m) in acc.items()}

  def capture(self, warmup=3):
    """Capture the device part of the step into one hipGraph (replayed by train_step)."""
    assert self._built and self.graph is None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
      for _ in range(warmup):
        self.features.version += 1
        self._refresh_hyper()
        self._device_step()
        self.global_step += 1
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    self.features.version += 1
    self._refresh_hyper()
    with torch.cuda.graph(g):
      self._device_step()
    # the capture itself does not execute: the device step counter is unchanged by it
    self.graph = g
    self._graph_signature = self.features.shape_signature()
    return g

  # -- host exchange (parity tests / checkpoints)
  def state_dict(self, slots=False, rows_of=None):
    sd = self.varstore.state_dict()
    sd.update(self.engine.state_dict(slots=slots, rows_of=rows_of) if rows_of is not None else self.engine.state_dict(slots=slots))
    if slots:
      for name in self.varstore.trainable_names():
        o, n = self.varstore._offsets[name]
        for s in ('m', 'v'):
          if s in self.varstore.slots:
            sd[name + '/' + s] = self.varstore.slots[s][o:o + n].view(
                self.varstore._vars[name]['tensor'].shape).cpu().numpy().copy()
    return sd

  def set_global_step(self, step):
    """Continue from `step` optimisation steps (checkpoint restore): host and device step counters, the optimizers'
    beta powers and the pre-planned per-step scalars restart there; the tables are taken to be current."""
    assert self.graph is None or step == self.global_step, 'restore before capture()'
    if self.device.type == 'cuda':
      torch.cuda.synchronize()
    # counters and "nothing pending" first: growing the history must not replay stale decay onto restored rows
    self.global_step = int(step)
    self.step_counter.fill_(int(step))
    self.engine.mark_restored(int(step))
    self._grow_lr_history(int(step) + 1, flush=False)
    for opt in {id(o): o for o in (self.opt_emb, self.opt_dense)}.values():
      opt.reset_to_step(step)
    self._planned_until = int(step)
    self._plan_hyper(self.HYPER_SLOTS)

  def load_state_dict(self, state):
    self.varstore.load_state_dict(state, strict=False)
    self.engine.load_state_dict(state)

  def loss_values(self):
    torch.cuda.synchronize() if self.device.type == 'cuda' else None
    return {k: float(v.item()) for k, v in self.losses.items()}

"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by easyrec_amd/.

CPU restatement (torch-CPU autograd + numpy fp32) of the reference's whole training step for the
hot-path models, written from the cited reference source (alibaba/EasyRec v0.8.7, paths relative
to /root/reference/easy_rec/python) and TF's documented op semantics:

  preprocessing  input/input.py:537-555 (ids), :557-673 (raw: (x-min)/(max-min)),
                 compat/feature_column/feature_column.py:2599-2643 ('' dropped), v2:3903-3926 (hash)
  input layer    layers/input_layer.py:280-376, compat/feature_column/feature_column.py:384-414,
                 lookup = safe_embedding_lookup_sparse (compat/embedding_ops.py:37-162)
  DeepFM         model/deepfm.py:53-109; FM layers/fm.py:20-26; DNN layers/dnn.py:50-87
  DCN            model/dcn.py:32-70
  MultiTowerDIN  model/multi_tower_din.py:62-130, layers/seq_input_layer.py:34-124
  MMoE           model/mmoe.py:35-70, layers/mmoe.py:62-83, model/multi_task_model.py:33-141
  loss           model/rank_model.py:105-111,213-332, builders/loss_builder.py:35-39
  regularisation model/easy_rec_estimator.py:166-184, compat/regularizers.py:76-108
  optimizer      builders/optimizer_builder.py:61-66 (tf.train.AdamOptimizer, dense-decay sparse
                 apply), compat/adam_s.py:185-213 (lazy), core/learning_schedules.py:25-73

It shares with the product only the config schema (the drop-in boundary) and the host batch dict
format; variables are taken BY NAME from the product's initial state, so a naming or layout
disagreement fails loudly.

Parity status.  PINNED by outputs of the reference's own code (tests/golden/reference_layer_vectors.npz, produced by
tests/golden/make_reference_layer_vectors.py executing the reference's functions unmodified on a numpy stand-in for
the tensorflow module; tests/test_reference_layers.py): layers/fm.py FM, keras FM / Cross (full, diag_scale, low rank,
no bias) / CIN / DotInteraction (layers/keras/interaction.py), layers/dnn.py DNN, model/multi_tower_din.py din(),
keras MLP (layers/keras/blocks.py, configured through the reference's Parameter), keras DIN (layers/keras/din.py),
layers/sequence_feature_layer.py target_attention, layers/mmoe.py MMOE, model/dcn.py _cross_net,
core/learning_schedules.py exponential_decay_with_burnin; hashing by TensorFlow's documented vectors; the embedding
lookup by embed_test's vectors.  The assembly of the model classes (build_predict_graph of eleven classes), the
backbone DAG, keras MMoE / SENet and the FeatureColumnParser are pinned THROUGH THE PRODUCT (its model classes,
backbone, layers and parser are held to the reference's outputs; this oracle is held to the product by
tests/test_host_logic.py).  "Parity unpinned" (the reference's tests hold no numeric expectation, the code is
TensorFlow's own and TensorFlow cannot run here): the loss (tf.losses.sigmoid_cross_entropy), the optimizers' update
rules (tf.train.AdamOptimizer / AdamOptimizerS), the regularisation terms, BatchNorm's moving-average bookkeeping, the
evaluation of the feature columns (compat/feature_column/*).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from oracle import hashing

F32 = np.float32
BN_EPS, BN_MOMENTUM = 1e-3, 0.99


class _TF(object):
  """the tf.* calls config lambdas make, over torch (test infrastructure)"""

  @staticmethod
  def unstack(x, num=None, axis=0):
    return list(torch.unbind(x, dim=axis))

  @staticmethod
  def reshape(x, shape):
    return x.reshape(tuple(int(s) for s in shape))

  @staticmethod
  def reduce_sum(x, axis=None, keepdims=False):
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)

  @staticmethod
  def concat(values, axis=-1):
    return torch.cat(list(values), dim=axis)

  @staticmethod
  def stack(values, axis=0):
    return torch.stack(list(values), dim=axis)

  @staticmethod
  def add_n(values):
    out = values[0]
    for v in values[1:]:
      out = out + v
    return out


tf = _TF


def _fname(fc):
  return fc.feature_name if fc.HasField('feature_name') else fc.input_names[0]


class Vars(object):
  """name -> torch leaf (requires_grad) created from the numpy state on first use."""

  def __init__(self, state, dtype, compact=None, probe=None):
    self.state = state
    self.dtype = dtype
    self.used = OrderedDict()
    self.l2 = {}
    # compact: {table name: ascending ids} - state[name] holds ONLY those rows of the table, lookups map ids to positions
    # (OracleTrainer(compact_ids=...)); probe: {table name: (rows, dim)} - tables stand in as one zero row and the lookups
    # record the ids they see (OracleTrainer.probe_ids)
    self.compact = compact or {}
    self.probe = probe

  def get(self, name, l2=0.0, trainable=True):
    if name not in self.used and self.probe is not None and name in self.probe:
      t = torch.zeros(1, int(self.probe[name][1]), dtype=self.dtype)
      t._probe_name = name
      self.used[name] = t
      self.l2[name] = l2
    if name not in self.used:
      if name not in self.state:
        raise KeyError('oracle: variable %r not found in the product state (have e.g. %s)' %
                       (name, list(self.state)[:5]))
      t = torch.tensor(self.state[name], dtype=self.dtype)
      if trainable:
        t.requires_grad_(True)
      if name in self.compact:
        t._compact_ids = self.compact[name]
      self.used[name] = t
      self.l2[name] = l2
    return self.used[name]


class OracleTrainer(object):

  def __init__(self, cfg, state, batch_size, dtype=torch.float32, compact_ids=None):
    """compact_ids: {embedding table name: ascending int64 ids}: `state[name]` (and the slots handed to resume()) hold
    ONLY those rows - every id the batches of this trainer's life will look up must be among them.  Rows that no lookup
    reads cannot influence a loss, and TF-Adam's every-row decay acts on each row independently, so the losses and the
    listed rows are exactly those of the full tables: what makes the 200 M-row table of BASELINE config 5 (51 GB, 153 GB
    with Adam's slots) checkable on a host (bench.py parity_full_size, tests/test_compact_oracle.py)."""
    self.cfg = cfg
    self.B = batch_size
    self.dtype = dtype
    self.compact_ids = {k: np.asarray(v, dtype=np.int64) for k, v in (compact_ids or {}).items()}
    self._probe = None
    # hash-table (ev_params) tables arrive as (ids ascending, rows, meta): kept here as a dense arena of `capacity` rows
    # + an id -> row dict; rows are created on first sight from the same counter-based generator as the product
    # (oracle/kernel_ref.py kv_init_value), so a row's value does not depend on which arena position it gets
    self.kv = {}
    state = OrderedDict(state)
    for k in [k for k in state if k.endswith('/kv_meta')]:
      name = k[:-len('/kv_meta')]
      meta = [float(x) for x in state.pop(k)]
      seed, mean, std, cap = meta[:4]
      # ev_params.filter_freq / steps_to_live (DeepRec's CounterFilter / GlobalStepEvict, restated from their documented
      # behaviour in easyrec_amd/csrc/er_kv.hip's header): occurrences are counted per id, the id gets its row when the
      # count reaches filter_freq; every training lookup stamps the id with the global step
      filter_freq, steps_to_live = (int(meta[4]), int(meta[5])) if len(meta) >= 6 else (0, 0)
      keys = np.asarray(state.pop(name + '/keys'), dtype=np.int64)
      vals = np.asarray(state[name], dtype=np.float32)
      arena = np.zeros((int(cap), vals.shape[1]), dtype=np.float32)
      arena[:len(keys)] = vals
      state[name] = arena
      kv = {'map': {int(key): i for i, key in enumerate(keys)}, 'n_rows': len(keys), 'seed': int(seed), 'mean': mean,
            'stddev': std, 'filter_freq': filter_freq, 'steps_to_live': steps_to_live, 'freq': {}, 'version': {}}
      if (name + '/kv_seen_keys') in state:
        seen = np.asarray(state.pop(name + '/kv_seen_keys'), dtype=np.int64).tolist()
        kv['freq'] = dict(zip(seen, np.asarray(state.pop(name + '/kv_freq')).tolist()))
        kv['version'] = dict(zip(seen, np.asarray(state.pop(name + '/kv_version')).tolist()))
        for key in seen:
          kv['map'].setdefault(int(key), -1)
      self.kv[name] = kv
    self.state = OrderedDict((k, np.array(v, dtype=np.float32)) for k, v in state.items())
    self.features = list(cfg.feature_configs) if cfg.feature_configs else list(cfg.feature_config.features)
    self.fc_by_name = OrderedDict((_fname(f), f) for f in self.features)
    self.global_step = 0
    oc = cfg.train_config.optimizer_config
    self.opt = [self._opt_cfg(o) for o in oc]
    self.beta_pow = [[F32(o['beta1']), F32(o['beta2'])] for o in self.opt]
    self.slots = {}
    self.emb_mult = float(oc[0].embedding_learning_rate_multiplier) \
        if oc[0].HasField('embedding_learning_rate_multiplier') else 1.0
    self.model_class = cfg.model_config.model_class

  def resume(self, global_step, slots):
    """Continue from a training state taken elsewhere: `global_step` finished steps (LR schedule position, Adam's beta
    powers = beta^(step + 1), computed by repeated fp32 multiplication as TF's update op does) and the optimizer slots
    {'<var>/m', '<var>/v'}."""
    assert not self.kv, 'resume() with hash-table tables is not restated'
    self.global_step = int(global_step)
    self.slots = {k: np.array(v, dtype=np.float32) for k, v in slots.items()}
    for oi, o in enumerate(self.opt):
      b1p, b2p = F32(o['beta1']), F32(o['beta2'])
      for _ in range(self.global_step):
        b1p, b2p = F32(b1p * F32(o['beta1'])), F32(b2p * F32(o['beta2']))
      self.beta_pow[oi] = [b1p, b2p]

  # ------------------------------------------------------------------ optimizer config
  @staticmethod
  def _opt_cfg(o):
    kind = o.WhichOneof('optimizer')
    c = getattr(o, kind)
    d = {'kind': kind, 'beta1': getattr(c, 'beta1', 0.9), 'beta2': getattr(c, 'beta2', 0.999), 'lr_cfg': c.learning_rate,
         'initial_accumulator_value': getattr(c, 'initial_accumulator_value', 0.1)}
    return d

  def _lr(self, lr_cfg, step):
    kind = lr_cfg.WhichOneof('learning_rate')
    if kind == 'constant_learning_rate':
      return F32(lr_cfg.constant_learning_rate.learning_rate)
    if kind == 'exponential_decay_learning_rate':
      c = lr_cfg.exponential_decay_learning_rate
      # core/learning_schedules.py:25-73 + tf.train.exponential_decay (staircase default true)
      p = F32(step - c.burnin_steps) / F32(c.decay_steps)
      if c.staircase:
        p = np.floor(p)
      post = F32(c.initial_learning_rate) * np.power(F32(c.decay_factor), F32(p), dtype=F32)
      if c.burnin_learning_rate == 0:
        burn = F32(c.initial_learning_rate)
      else:
        burn = F32((c.initial_learning_rate - c.burnin_learning_rate) / c.burnin_steps) * F32(step) + \
            F32(c.burnin_learning_rate)
      lr = burn if step < c.burnin_steps else post
      return F32(max(F32(lr), F32(c.min_learning_rate)))
    raise NotImplementedError(kind)

  # ------------------------------------------------------------------ preprocessing
  def _hashed_ids(self, batch):
    """int64 [n_hash_features, B], -1 where the string is '' (dropped before hashing)."""
    # (a ComboFeature with combo_join_sep is ONE hashed column over the inputs' strings joined by the separator,
    # feature_column/feature_column.py:446-455, input/input.py:408-430: the joined string arrives from the input stage)
    names = [n for n, f in self.fc_by_name.items()
             if (f.feature_type == f.IdFeature or (f.feature_type == f.ComboFeature and len(f.combo_join_sep) > 0))
             and f.HasField('hash_bucket_size') and f.hash_bucket_size > 0]
    if not names:
      return {}
    if 'hash_ids' in batch:
      ids = np.asarray(batch['hash_ids'])
    else:
      # (ev_params: the id is the hash into the whole int64 range, feature_column/feature_column.py:19, 222-226)
      nb = np.array([(2**63 - 1) if self.fc_by_name[n].HasField('ev_params') else self.fc_by_name[n].hash_bucket_size
                     for n in names], dtype=np.uint64)
      ids = hashing.hash_bucket_fast(np.asarray(batch['str_bytes']), np.asarray(batch['str_offsets']), self.B, nb,
                                     True).reshape(len(names), self.B)
    return {n: ids[i] for i, n in enumerate(names)}

  def _raw_values(self, batch):
    out, row = {}, 0
    for n, f in self.fc_by_name.items():
      if f.feature_type == f.RawFeature:
        if f.raw_input_dim > 1:
          out[n] = np.asarray(batch['rawm/%s' % n], dtype=np.float32)
        else:
          out[n] = np.asarray(batch['raw'], dtype=np.float32)[row]
          row += 1
    return out

  @staticmethod
  def _bounds(f):
    """feature_column/feature_column.py:365-376"""
    numeric = f.feature_type == f.RawFeature or (f.feature_type == f.SequenceFeature and f.sub_feature_type == f.RawFeature)
    if not numeric or f.raw_input_dim > 1:
      return None
    if len(f.boundaries) > 0:
      return sorted(np.float32(b) for b in f.boundaries)
    if f.num_buckets > 1 and f.max_val > f.min_val:
      return [np.float32(x / float(f.num_buckets)) for x in range(f.num_buckets)]
    return None

  def _int_ids(self, batch):
    out, col = {}, 0
    for n, f in self.fc_by_name.items():
      if f.feature_type == f.IdFeature and not (f.HasField('hash_bucket_size') and f.hash_bucket_size > 0):
        out[n] = np.asarray(batch['int_ids'])[col]
        col += 1
      elif f.feature_type == f.RawFeature and self._bounds(f) is not None:
        col += 1  # the batch carries the bucket index too; the oracle re-derives it from the raw value
      elif f.feature_type == f.ComboFeature and len(f.combo_join_sep) == 0 and not any(len(x) for x in f.combo_input_seps):
        out[n] = np.asarray(batch['int_ids'])[col]  # crossed id from the input stage (oracle/hashing.py pins it)
        col += 1
    return out

  # ------------------------------------------------------------------ embedding columns
  def _column_var_name(self, scope, fc, wide):
    name = _fname(fc)
    if (fc.feature_type == fc.RawFeature or (fc.feature_type == fc.SequenceFeature and fc.sub_feature_type == fc.RawFeature)) \
        and self._bounds(fc) is not None:
      col = '%s_bucketized' % name  # BucketizedColumn.name (feature_column_v2.py:2777-2779; Sequence-: :2934-2936)
    elif fc.feature_type == fc.RawFeature:
      col = '%s_weighted_by_%s_raw_proj_val' % (name, name)
    elif fc.feature_type == fc.TagFeature and (len(fc.input_names) > 1 or fc.HasField('kv_separator')):
      col = '%s_weighted_by_%s_w' % (name, name)
    else:
      col = name
    if fc.HasField('embedding_name') and self._is_shared(fc.embedding_name):
      return '%s%s/embedding_weights' % (fc.embedding_name, '_wide' if wide else '')
    return '%s/%s_embedding/embedding_weights' % (scope, col)

  def _is_shared(self, embedding_name):
    return sum(1 for f in self.features if f.HasField('embedding_name') and f.embedding_name == embedding_name) > 1

  def _touch(self, table, ids):
    key = id(table)
    mask = self._touched.setdefault(key, np.zeros(table.shape[0], dtype=bool))
    ids = np.asarray(ids).reshape(-1)
    ok = (ids >= 0) & (ids < table.shape[0])
    mask[ids[ok]] = True

  def _kv_rows(self, name, ids):
    """ids -> arena rows of a hash-table table in a training lookup: every occurrence counts, an id gets the next row
    (initialised from the generator) on first sight or - filter_freq > 1 - in the lookup that brings its count to
    filter_freq; ids without a row read zeros (-1) and take no update."""
    from oracle.kernel_ref import RefBackend
    kv, arena = self.kv[name], self.state[name]
    ids = np.asarray(ids, dtype=np.int64).tolist()
    counted = kv['filter_freq'] > 1
    # (W workers: the owners see the ids of ALL workers before any row is read - train_step_world runs the workers'
    #  lookups once in 'insert' mode, then forward / backward in 'find' mode)
    for key in (ids if getattr(self, '_kv_mode', 'both') != 'find' else ()):
      if key < 0:
        continue
      known = key in kv['map']
      if not known:
        kv['map'][key] = -1
      if kv['steps_to_live'] > 0:
        kv['version'][key] = self.global_step + 1  # (the global step this training step ends with)
      create = not known
      if counted:
        create = False
        if kv['freq'].get(key, 0) < kv['filter_freq']:
          kv['freq'][key] = kv['freq'].get(key, 0) + 1
          create = kv['freq'][key] == kv['filter_freq']
      if create:
        r = kv['n_rows']
        assert r < arena.shape[0], 'oracle: hash-table %s is full' % name
        kv['n_rows'] += 1
        arena[r] = RefBackend.kv_init_value(kv['seed'], [key], arena.shape[1], kv['mean'], kv['stddev'])[0]
        for slot in ('/m', '/v'):  # (a row that an evicted id owned before: never the case here, rows are not reused)
          assert (name + slot) not in self.slots or not self.slots[name + slot][r].any()
        kv['map'][key] = r
    return np.array([-1 if key < 0 else kv['map'].get(key, -1) for key in ids], dtype=np.int64)

  def kv_evict(self, name):
    """steps_to_live at checkpoint time: ids whose last training lookup is more than steps_to_live steps back are
    dropped (their arena rows are simply never used again here).  -> the number of ids dropped."""
    kv = self.kv[name]
    if kv['steps_to_live'] <= 0:
      return 0
    stale = [k for k in kv['map'] if self.global_step - kv['version'].get(k, 0) > kv['steps_to_live']]
    for k in stale:
      del kv['map'][k]
      kv['freq'].pop(k, None)
      kv['version'].pop(k, None)
    return len(stale)

  def kv_state(self, name, array=None):
    """(ids ascending, their rows of `array` - default the table itself) of a hash-table table's ids that have a row"""
    items = sorted((k, r) for k, r in self.kv[name]['map'].items() if r >= 0)
    src = self.state[name] if array is None else array
    return np.array([k for k, _ in items], dtype=np.int64), np.stack([src[r] for _, r in items]) if items else src[:0]

  def kv_filter_state(self, name):
    """(every tracked id ascending, min(count, filter_freq), last-lookup step)"""
    kv = self.kv[name]
    keys = sorted(kv['map'])
    ff = max(kv['filter_freq'], 1)
    return (np.array(keys, dtype=np.int64), np.array([min(kv['freq'].get(k, 0), ff) for k in keys], dtype=np.int32),
            np.array([kv['version'].get(k, 0) for k in keys], dtype=np.int32))

  def _per_lookup(self, table):
    """When gradients are clipped: the table as THIS lookup sees it (an identity op whose gradient is kept), so that the
    gradient each lookup sends to a shared table can be told apart afterwards - TensorFlow keeps the IndexedSlices of
    the lookups of one variable side by side (merged within a lookup: embedding_lookup_sparse gathers unique ids) and
    `_get_grad_norm` takes l2_loss of their concatenated values (compat/optimizers.py:453-481)."""
    if not getattr(self, '_track_lookups', False) or not table.requires_grad:
      return table
    t = table + 0
    t.retain_grad()
    self._lookup_leaves.setdefault(id(table), []).append(t)
    return t

  def _table_ids(self, table, ids):
    """ids as positions of `table`: unchanged for a full table; for a compact one (Vars.compact) the position of each id
    in its ascending id list (an id that is not listed is a caller error); a probe table records them and reads row 0."""
    ids = np.asarray(ids)
    name = getattr(table, '_probe_name', None)
    if name is not None:
      rows = int(self._probe[name][0])
      ok = (ids >= 0) & (ids < rows)
      self._probe_seen.setdefault(name, []).append(np.unique(ids[ok]).astype(np.int64))
      return np.where(ok, 0, -1).astype(np.int64)
    listed = getattr(table, '_compact_ids', None)
    if listed is None:
      return ids
    pos = np.searchsorted(listed, ids)
    posc = np.minimum(pos, len(listed) - 1)
    hit = (ids >= 0) & (listed[posc] == ids)
    assert bool(np.all(hit | (ids < 0) | (ids >= int(getattr(table, '_full_rows', 1 << 62))))), \
        'compact oracle: a looked-up id is not among compact_ids'
    return np.where(hit, posc, -1).astype(np.int64)

  def _lookup_dense(self, table, ids, weights=None):
    """One id per example; id < 0 -> zero row; optional weight multiplies the row (combiner sum)."""
    ids = self._table_ids(table, ids)
    self._touch(table, ids)
    table = self._per_lookup(table)
    idt = torch.as_tensor(np.asarray(ids), dtype=torch.int64)
    ok = (idt >= 0) & (idt < table.shape[0])
    e = table[torch.where(ok, idt, torch.zeros_like(idt))]
    if weights is not None:
      e = e * torch.as_tensor(np.asarray(weights), dtype=self.dtype)[:, None]
    return e * ok.to(self.dtype)[:, None]

  def _lookup_ragged(self, table, ids, offsets, weights, combiner):
    rows = []
    ids = self._table_ids(table, np.asarray(ids))
    self._touch(table, ids[int(offsets[0]):int(offsets[self.B])])
    table = self._per_lookup(table)
    for r in range(self.B):
      kb, ke = int(offsets[r]), int(offsets[r + 1])
      acc = torch.zeros(table.shape[1], dtype=self.dtype)
      wsum, w2 = 0.0, 0.0
      for k in range(kb, ke):
        i = int(ids[k])
        if i < 0 or i >= table.shape[0]:
          continue
        w = 1.0 if weights is None else float(weights[k])
        if weights is not None and combiner != 'sum' and not (w > 0):
          continue
        acc = acc + table[i] * w if weights is not None else acc + table[i]
        wsum += w
        w2 += w * w
      if combiner == 'mean' and wsum != 0:
        acc = acc / wsum
      elif combiner == 'sqrtn' and wsum != 0:
        acc = acc / math.sqrt(w2)
      rows.append(acc)
    return torch.stack(rows)

  def input_layer(self, V, batch, group_name, scope, wide_dim=None, only=None):
    """Returns (concat [B, sum dim], [per feature tensors]); adds the embedding L2 to self._reg.  only: restrict to
    these features of the group (the plain columns of a group that also holds sequence features)."""
    group = [g for g in self.cfg.model_config.feature_groups if g.group_name == group_name][0]
    wide = (group.wide_deep == 1)
    names = []
    for n in group.feature_names:
      m = __import__('re').match(r'([a-zA-Z_]+)\[([0-9]+)-([0-9]+)\]', n)
      names.extend(['%s%d' % (m.group(1), t) for t in range(int(m.group(2)), int(m.group(3)) + 1)] if m else [n])
    if only is not None:
      names = [n for n in names if n in only]
    hashed, raws, ints = self._cache
    outs = []
    lam = self.cfg.model_config.embedding_regularization
    for n in names:
      fc = self.fc_by_name[n]
      dim = wide_dim if wide else fc.embedding_dim
      if fc.feature_type == fc.RawFeature and self._bounds(fc) is not None:
        # BucketizedColumn: id = number of boundaries <= value (feature_column_v2.py:2762-2916)
        bounds = self._bounds(fc)
        ids = np.array([sum(1 for b in bounds if b <= np.float32(v)) for v in raws[n]], dtype=np.int64)
        outs.append((self._lookup_dense(V.get(self._column_var_name(scope, fc, wide)), ids), True))
      elif fc.feature_type == fc.RawFeature:
        if dim == 0:
          v = torch.as_tensor(raws[n], dtype=self.dtype)
          outs.append((v.reshape(self.B, -1), False))
          continue
        table = V.get(self._column_var_name(scope, fc, wide))
        if fc.raw_input_dim > 1:
          k = fc.raw_input_dim
          ids = np.tile(np.arange(k), self.B)
          e = self._lookup_ragged(table, ids, np.arange(self.B + 1) * k, raws[n].reshape(-1),
                                  'sum' if wide else fc.combiner)
        else:
          e = self._lookup_dense(table, np.zeros(self.B, dtype=np.int64), raws[n])
        outs.append((e, True))
      elif fc.feature_type == fc.IdFeature or (fc.feature_type == fc.ComboFeature and (n in ints or n in hashed)):
        # (crossed ComboFeature: CrossedColumn under an EmbeddingColumn, the id comes from the input stage)
        var_name = self._column_var_name(scope, fc, wide)
        ids = hashed[n] if n in hashed else ints[n]
        if var_name in self.kv:  # (before V.get: new rows are written into the state the leaf is made from)
          ids = self._kv_rows(var_name, ids)
        table = V.get(var_name)
        outs.append((self._lookup_dense(table, ids), True))
      elif fc.feature_type in (fc.TagFeature, fc.LookupFeature) or (
          fc.feature_type == fc.ComboFeature and ('tag/%s/ids' % n) in batch):
        # (LookupFeature: the selected map values arrive from the input stage as a ragged id list, input.py:941-1000)
        var_name = self._column_var_name(scope, fc, wide)
        tag_ids = batch['tag/%s/ids' % n]
        if var_name in self.kv:
          tag_ids = self._kv_rows(var_name, tag_ids)
        table = V.get(var_name)
        w = batch.get('tag/%s/weights' % n)
        e = self._lookup_ragged(table, tag_ids, batch['tag/%s/offsets' % n], w,
                                'sum' if wide else fc.combiner)
        outs.append((e, True))
      else:
        raise NotImplementedError('oracle: feature type of %s' % n)
    if lam > 0:
      for e, is_emb in outs:
        if is_emb:
          self._reg = self._reg + lam * 0.5 * (e * e).sum()  # scale * tf.nn.l2_loss(out_f)
    feats = [e for e, _ in outs]
    concat = torch.cat(feats, dim=1)
    if only is None and len(group.sequence_features) > 0:
      # target attention over the group's sequence_features (layers/input_layer.py:96-111, sequence_feature_layer.py:
      # 215-270): keys = the group's own outputs (allow_key_search false) or tables of their own under the group's
      # name scope; history tables under the group's name scope; embedding L2 on own keys and on the history
      by_name = dict(zip(names, feats))
      l2 = self._l2_of(self.cfg.model_config)
      for sc in group.sequence_features:
        keys, hists, seq_len = [], [], None
        for m in sc.seq_att_map:
          for k in m.key:
            if k in by_name and not sc.allow_key_search:
              keys.append(by_name[k])
            elif k in by_name:
              keys.append(by_name[k])  # (present in the group: reused either way, seq_input_layer.py:66-79)
            else:
              fc = self.fc_by_name[k]
              e = self._id_lookup(V, self._column_var_name(group_name, fc, False), self._categorical_ids(batch, fc, k))
              if lam > 0:
                self._reg = self._reg + lam * 0.5 * (e * e).sum()
              keys.append(e)
          for h in m.hist_seq:
            fc = self.fc_by_name[h]
            e, lens = self._seq_lookup(V, self._column_var_name(group_name, fc, False), batch, h)
            hists.append(e)
            seq_len = lens if seq_len is None else seq_len
        hist = torch.cat(hists, dim=-1)
        if lam > 0:
          self._reg = self._reg + lam * 0.5 * (hist * hist).sum()
        fea = {'key': torch.cat(keys, dim=-1), 'hist_seq_emb': hist, 'hist_seq_len': seq_len}
        if sc.HasField('seq_dnn'):
          dnn_cfg = sc.seq_dnn
        else:
          from easyrec_amd.protos.dnn_pb2 import DNN  # (the config schema is the shared boundary)
          dnn_cfg = DNN()
          dnn_cfg.hidden_units.extend([128, 64, 32, 1])
        att = self._din(V, dnn_cfg, fea, 'seq_dnn' + sc.group_name, l2, sc.allow_key_transform, sc.transform_dnn)
        if not sc.need_key_feature:
          att = att[:, :hist.shape[-1]]
        feats = feats + [att]
        concat = torch.cat([concat, att], dim=-1)
    return concat, feats

  def _categorical_ids(self, batch, fc, name):
    hashed, raws, ints = self._cache
    return hashed[name] if name in hashed else ints[name]

  def _id_lookup(self, V, var_name, ids):
    """one id per example into table `var_name` (hash-table tables: ids -> arena rows first, then the leaf)"""
    if var_name in self.kv:  # (before V.get: new rows are written into the state the leaf is made from)
      ids = self._kv_rows(var_name, ids)
    return self._lookup_dense(V.get(var_name), ids)

  def _seq_lookup(self, V, var_name, batch, name):
    """EmbeddingColumn._get_sequence_dense_tensor (feature_column_v2.py:3616-3640): [B, L, E] with L = the BATCH's
    longest sequence (the sparse tensor's dense shape; compat/feature_column/utils.py:30-54), zero rows for padding."""
    ids = np.asarray(batch['seq/%s/ids' % name])  # [B, max_seq_len], -1 padded
    lens = np.asarray(batch['seq/%s/len' % name]).astype(np.int64)
    L = max(1, int(lens.max())) if getattr(self, 'pad_to_batch_max', True) else ids.shape[1]
    ids = ids[:, :L]
    B = ids.shape[0]
    flat = ids.reshape(-1)
    if var_name in self.kv:  # a hash-table (ev_params) sequence: ids -> arena rows (padding -> -1), before the leaf exists
      flat = self._kv_rows(var_name, flat)
    return self._lookup_dense(V.get(var_name), flat).reshape(B, L, -1), lens

  def seq_input_layer(self, V, batch, group_name):
    """layers/seq_input_layer.py:34-124: keys under variable_scope(group_name), history sequences keep the
    time axis ([B, L, E], zero rows for padding); embedding L2 on key and history outputs
    (:68-70, model/multi_tower_din.py:54-60)."""
    grp = [g for g in self.cfg.model_config.seq_att_groups if g.group_name == group_name][0]
    lam = self.cfg.model_config.embedding_regularization
    keys, hists, seq_len = [], [], None
    for m in grp.seq_att_map:
      for k in m.key:
        fc = self.fc_by_name[k]
        keys.append(self._id_lookup(V, self._column_var_name(group_name, fc, False), self._categorical_ids(batch, fc, k)))
      for h in m.hist_seq:
        fc = self.fc_by_name[h]
        e, lens = self._seq_lookup(V, self._column_var_name(group_name, fc, False), batch, h)
        hists.append(e)
        if seq_len is None:
          seq_len = lens
    key = torch.cat(keys, dim=-1)
    hist = torch.cat(hists, dim=-1)
    if lam > 0:
      for k in keys:
        self._reg = self._reg + lam * 0.5 * (k * k).sum()
      self._reg = self._reg + lam * 0.5 * (hist * hist).sum()
    return {'key': key, 'hist_seq_emb': hist, 'hist_seq_len': seq_len}

  # ------------------------------------------------------------------ dense layers
  def dense(self, V, x, units, name, l2):
    w = V.get(name + '/kernel', l2=l2)
    b = V.get(name + '/bias')
    assert w.shape == (x.shape[-1], units)
    return x @ w + b

  def batch_norm(self, V, x, name, training=True):
    gamma, beta = V.get(name + '/gamma'), V.get(name + '/beta')
    mm, mv = V.get(name + '/moving_mean', trainable=False), V.get(name + '/moving_variance', trainable=False)
    if not training:  # tf.layers.batch_normalization(training=False): the moving statistics, no update op
      return (x - mm) / torch.sqrt(mv + BN_EPS) * gamma + beta
    axes = tuple(range(x.dim() - 1))
    mean = x.mean(dim=axes)
    var = ((x - mean)**2).mean(dim=axes)
    y = (x - mean) / torch.sqrt(var + BN_EPS) * gamma + beta
    with torch.no_grad():  # assign_moving_average, UPDATE_OPS forced before the loss (estimator :202-213)
      self._moving[name + '/moving_mean'] = mm - (mm - mean) * (1 - BN_MOMENTUM)
      self._moving[name + '/moving_variance'] = mv - (mv - var) * (1 - BN_MOMENTUM)
    return y

  def dice(self, V, x, name):
    alpha = V.get('alpha_' + name)
    mm = V.get(name + '/batch_normalization/moving_mean', trainable=False)
    mv = V.get(name + '/batch_normalization/moving_variance', trainable=False)
    axes = tuple(range(x.dim() - 1))
    mean = x.mean(dim=axes)
    var = ((x - mean)**2).mean(dim=axes)
    p = torch.sigmoid((x - mean) / torch.sqrt(var + 1e-9))
    with torch.no_grad():
      self._moving[name + '/batch_normalization/moving_mean'] = mm - (mm - mean) * (1 - BN_MOMENTUM)
      self._moving[name + '/batch_normalization/moving_variance'] = mv - (mv - var) * (1 - BN_MOMENTUM)
    return alpha * (1.0 - p) * x + p * x

  def dnn(self, V, x, dnn_cfg, name, l2, last_no_act=False, last_no_bn=False, training=True):
    """layers/dnn.py:50-87.  training = the layer's is_training (BatchNorm's mode)."""
    n = len(dnn_cfg.hidden_units)
    for i, units in enumerate(dnn_cfg.hidden_units):
      x = self.dense(V, x, units, '%s/dnn_%d' % (name, i), l2)
      if dnn_cfg.use_bn and ((i + 1 < n) or not last_no_bn):
        x = self.batch_norm(V, x, '%s/dnn_%d/bn' % (name, i), training=training)
      if (i + 1 < n) or not last_no_act:
        act = dnn_cfg.activation.lower()
        if act in ('tf.nn.relu', 'relu', 'nn.relu'):
          x = torch.relu(x)
        elif act == 'dice':
          assert training, 'dice inside an is_training=False DNN is not restated'
          x = self.dice(V, x, '%s/dnn_%d/act' % (name, i))
        else:
          raise NotImplementedError(act)
      assert len(dnn_cfg.dropout_ratio) == 0 or all(r == 0 for r in dnn_cfg.dropout_ratio)
    return x

  def _l2_of(self, mc):
    sub = getattr(mc, mc.WhichOneof('model'))
    if hasattr(sub, 'dense_regularization') and sub.HasField('dense_regularization'):
      return sub.dense_regularization
    return getattr(sub, 'l2_regularization', 0.0)

  # ------------------------------------------------------------------ models
  def _deepfm(self, V, batch):
    mc = self.cfg.model_config
    c = mc.deepfm
    l2 = self._l2_of(mc)
    wide, _ = self.input_layer(V, batch, 'wide', 'input_layer', wide_dim=c.wide_output_dim)
    deep, fm_list = self.input_layer(V, batch, 'deep', 'input_layer_1')
    if any(g.group_name == 'fm' for g in mc.feature_groups):
      _, fm_list = self.input_layer(V, batch, 'fm', 'input_layer_2')
    wide_fea = wide.sum(dim=1, keepdim=True)
    fm_feas = torch.stack(fm_list, dim=1)
    fm_fea = 0.5 * (fm_feas.sum(dim=1)**2 - (fm_feas**2).sum(dim=1))
    deep_fea = self.dnn(V, deep, c.dnn, 'deep_feature', l2)
    if len(c.final_dnn.hidden_units) > 0:
      all_fea = torch.cat([wide_fea, fm_fea, deep_fea], dim=1)
      all_fea = self.dnn(V, all_fea, c.final_dnn, 'final_dnn', l2)
      out = self.dense(V, all_fea, mc.num_class, 'output', l2)
    else:
      out = wide_fea + fm_fea.sum(dim=1, keepdim=True) + self.dense(V, deep_fea, mc.num_class, 'deep_logits', l2)
    return {'logits': out.squeeze(1)}

  def _cross_net(self, V, x0, num_layers):
    """model/dcn.py:32-45: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l."""
    x = x0
    for i in range(num_layers):
      w = V.get('cross_layer_%d_w' % i)
      b = V.get('cross_layer_%d_b' % i)
      xw = (x * w).sum(dim=1, keepdim=True)
      x = (x0 * xw + b) + x
    return x

  def _dcn(self, V, batch):
    mc = self.cfg.model_config
    c = mc.dcn
    l2 = self._l2_of(mc)
    feats, _ = self.input_layer(V, batch, 'all', 'input_layer')
    deep = self.dnn(V, feats, c.deep_tower.dnn, 'dnn', l2)
    x = self._cross_net(V, feats, c.cross_tower.cross_num)
    all_fea = torch.cat([deep, x], dim=1)
    all_fea = self.dnn(V, all_fea, c.final_dnn, 'final_dnn', l2)
    out = self.dense(V, all_fea, mc.num_class, 'output', 0.0)  # no kernel_regularizer (dcn.py:66)
    return {'logits': out.squeeze(1)}

  def _wide_and_deep(self, V, batch):
    """easy_rec/python/model/wide_and_deep.py:38-86"""
    mc = self.cfg.model_config
    c = mc.wide_and_deep
    l2 = self._l2_of(mc)
    has_final = len(c.final_dnn.hidden_units) > 0
    wd = c.wide_output_dim if has_final else mc.num_class
    _, wide_list = self.input_layer(V, batch, 'wide', 'input_layer', wide_dim=wd)
    deep, _ = self.input_layer(V, batch, 'deep', 'input_layer_1')
    wide_fea = torch.stack(wide_list, dim=0).sum(dim=0)  # add_n
    deep_fea = self.dnn(V, deep, c.dnn, 'deep_feature', l2)
    if has_final:
      all_fea = self.dnn(V, torch.cat([wide_fea, deep_fea], dim=1), c.final_dnn, 'final_dnn', l2)
      out = self.dense(V, all_fea, mc.num_class, 'output', l2)
    else:
      out = self.dense(V, deep_fea, mc.num_class, 'deep_out', l2) + wide_fea
    return {'logits': out.squeeze(1)}

  def _fm(self, V, batch):
    """easy_rec/python/model/fm.py:35-63"""
    mc = self.cfg.model_config
    wide, _ = self.input_layer(V, batch, 'wide', 'input_layer', wide_dim=mc.num_class)
    _, fm_list = self.input_layer(V, batch, 'deep', 'input_layer_1')
    wide_fea = wide.sum(dim=1, keepdim=True)
    e = torch.stack(fm_list, dim=1)
    fm_fea = 0.5 * (e.sum(dim=1)**2 - (e**2).sum(dim=1))
    assert mc.num_class == 1
    out = (wide_fea + fm_fea.sum(dim=1, keepdim=True)) + V.get('fm_bias')
    return {'logits': out.squeeze(1)}

  def _multi_tower(self, V, batch):
    """easy_rec/python/model/multi_tower.py:37-62"""
    mc = self.cfg.model_config
    c = mc.multi_tower
    l2 = self._l2_of(mc)
    feats = [self.input_layer(V, batch, t.input, 'input_layer' if i == 0 else 'input_layer_%d' % i)[0]
             for i, t in enumerate(c.towers)]
    outs = []
    for t, fea in zip(c.towers, feats):
      fea = self.batch_norm(V, fea, '%s_fea_bn' % t.input)
      outs.append(self.dnn(V, fea, t.dnn, '%s_dnn' % t.input, l2))
    all_fea = self.dnn(V, torch.cat(outs, dim=1), c.final_dnn, 'final_dnn', l2)
    out = self.dense(V, all_fea, mc.num_class, 'output', 0.0)  # no kernel_regularizer (multi_tower.py:58)
    return {'logits': out.squeeze(1)}

  def _dlrm(self, V, batch):
    """easy_rec/python/model/dlrm.py:36-73"""
    mc = self.cfg.model_config
    c = mc.dlrm
    l2 = self._l2_of(mc)
    _, sparse = self.input_layer(V, batch, 'sparse', 'input_layer')
    dense, _ = self.input_layer(V, batch, 'dense', 'input_layer_1')
    dense_fea = self.dnn(V, dense, c.bot_dnn, 'bot_dnn', l2)
    if c.arch_interaction_op == 'cat':
      all_fea = torch.cat([dense_fea] + sparse, dim=1)
    else:
      feas = torch.stack([dense_fea] + sparse, dim=1)
      inter = torch.einsum('bne,bme->bnm', feas, feas)
      off = 0 if c.arch_interaction_itself else 1
      n = feas.shape[1]
      upper = torch.cat([inter[:, i, i + off:n] for i in range(n)], dim=1)
      parts = [upper] + sparse
      if c.arch_with_dense_feature:
        parts.append(dense_fea)
      all_fea = torch.cat(parts, dim=1)
    all_fea = self.dnn(V, all_fea, c.top_dnn, 'top_dnn', l2)
    out = self.dense(V, all_fea, 1, 'output', l2)
    return {'logits': out.squeeze(1)}

  def _din(self, V, dnn_cfg, fea, name, l2, allow_key_transform=False, transform_dnn=False):
    """model/multi_tower_din.py:62-97; layers/sequence_feature_layer.py:123-189 with its key transform (:138-147: a key
    of another width than the history is zero-padded up to it, or - transform_dnn, or a wider key - both go through a
    dense layer of the history's width)."""
    q, h, seq_len = fea['key'], fea['hist_seq_emb'], fea['hist_seq_len']
    B, L, E = h.shape
    if allow_key_transform and q.shape[-1] != E:
      if E > q.shape[-1] and not transform_dnn:
        q = torch.nn.functional.pad(q, (0, E - q.shape[-1]))
      else:
        q = self.dense(V, q, E, 'sequence_key_transform_layer_' + name, 0.0)
        h = self.dense(V, h, E, 'sequence_fea_transform_layer_' + name, 0.0)
    cur = q[:, None, :].expand(B, L, E)
    din_net = torch.cat([cur, h, cur - h, cur * h], dim=-1)
    din_net = self.dnn(V, din_net, dnn_cfg, name, l2, last_no_act=True, last_no_bn=True)
    scores = din_net.reshape(B, 1, L)
    mask = (torch.arange(L)[None, :] < torch.as_tensor(seq_len)[:, None])[:, None, :]
    scores = torch.where(mask, scores, torch.full_like(scores, float(-2**32 + 1)))
    scores = torch.softmax(scores, dim=-1)
    pooled = torch.matmul(scores, h).reshape(B, E)
    return torch.cat([pooled, q], dim=1)

  def _multi_tower_din(self, V, batch):
    mc = self.cfg.model_config
    c = mc.multi_tower
    l2 = self._l2_of(mc)
    feas, scope_id = [], 0
    for tower in c.towers:
      scope = 'input_layer' if scope_id == 0 else 'input_layer_%d' % scope_id
      scope_id += 1
      fea, _ = self.input_layer(V, batch, tower.input, scope)
      feas.append(fea)
    din_feas = [self.seq_input_layer(V, batch, t.input) for t in c.din_towers]
    arr = []
    for tower, fea in zip(c.towers, feas):
      fea = self.batch_norm(V, fea, '%s_fea_bn' % tower.input)
      arr.append(self.dnn(V, fea, tower.dnn, '%s_dnn' % tower.input, l2))
    for tower, fea in zip(c.din_towers, din_feas):
      arr.append(self._din(V, tower.dnn, fea, '%s_dnn' % tower.input, l2))
    all_fea = self.dnn(V, torch.cat(arr, dim=1), c.final_dnn, 'final_dnn', l2)
    out = self.dense(V, all_fea, mc.num_class, 'output', 0.0)
    return {'logits': out.squeeze(1)}

  def _mmoe_layer(self, V, x, expert_cfgs, num_task, l2, name='mmoe', training=False):
    """layers/mmoe.py:62-83: expert DNNs stacked on axis 1, per task a softmax gate over the experts, the mixture.
    The model classes build the layer without is_training (model/mmoe.py:37-47, model/dbmtl.py:66-70; the layer's
    default is False, layers/mmoe.py:20): the experts' BatchNorm normalises with the moving statistics."""
    experts = torch.stack([self.dnn(V, x, cfg, '%s/expert_%d' % (name, i), l2, training=training)
                           for i, cfg in enumerate(expert_cfgs)], dim=1)
    out = []
    for t in range(num_task):
      gate = torch.softmax(self.dense(V, x, len(expert_cfgs), '%s/gate_%d/dnn' % (name, t), l2), dim=1)
      out.append((experts * gate[:, :, None]).sum(dim=1))
    return out

  def _mmoe(self, V, batch):
    """model/mmoe.py:35-70, layers/mmoe.py:62-83."""
    mc = self.cfg.model_config
    c = mc.mmoe
    l2 = self._l2_of(mc)
    x, _ = self.input_layer(V, batch, 'all', 'input_layer')
    if c.HasField('expert_dnn'):
      cfgs = [c.expert_dnn] * c.num_expert
    else:
      cfgs = [e.dnn for e in c.experts]
    task_inputs = self._mmoe_layer(V, x, cfgs, len(c.task_towers), l2)
    pred = {}
    for t, tower in enumerate(c.task_towers):
      task_in = task_inputs[t]
      if tower.HasField('dnn'):
        task_in = self.dnn(V, task_in, tower.dnn, tower.tower_name, l2)
      out = self.dense(V, task_in, tower.num_class, 'dnn_output_%d' % t, l2)
      pred['logits_%s' % tower.tower_name] = out.squeeze(1)
    return pred

  def _simple_multi_task(self, V, batch):
    """model/simple_multi_task.py:36-54: one tower DNN per task on the shared input."""
    mc = self.cfg.model_config
    l2 = self._l2_of(mc)
    x, _ = self.input_layer(V, batch, 'all', 'input_layer')
    pred = {}
    for t, tower in enumerate(mc.simple_multi_task.task_towers):
      h = self.dnn(V, x, tower.dnn, tower.tower_name, l2)
      pred['logits_%s' % tower.tower_name] = self.dense(V, h, tower.num_class, 'dnn_output_%d' % t, l2).squeeze(1)
    return pred

  def _ple(self, V, batch):
    """model/ple.py:36-118: CGC layers (task experts + shared experts, one softmax gate per task and one for the
    shared path except in the last layer), then the towers."""
    mc = self.cfg.model_config
    c = mc.ple
    l2 = self._l2_of(mc)
    x, _ = self.input_layer(V, batch, 'all', 'input_layer')
    T = len(c.task_towers)

    def gate(sel, cands, name):
      g = torch.softmax(self.dense(V, sel, len(cands), name + '_gate/dnn', l2), dim=1)
      return (torch.stack(cands, dim=1) * g[:, :, None]).sum(dim=1)

    task_in, shared_in = [x] * T, x
    nets = list(c.extraction_networks)
    for li, net in enumerate(nets):
      nm = net.network_name
      shared = [self.dnn(V, shared_in, net.share_expert_net, '%s_share/dnn_expert_%d/dnn' % (nm, e), l2)
                for e in range(net.share_num)]
      all_task, outs = [], []
      for t in range(T):
        tn = '%s_task_%d' % (nm, t)
        mine = [self.dnn(V, task_in[t], net.task_expert_net, '%s_expert_%d/dnn' % (tn, e), l2)
                for e in range(net.expert_num_per_task)]
        outs.append(gate(task_in[t], mine + shared, tn))
        all_task.extend(mine)
      new_shared = None if li == len(nets) - 1 else gate(shared_in, all_task + shared, nm + '_share')
      task_in, shared_in = outs, new_shared
    pred = {}
    for t, tower in enumerate(c.task_towers):
      h = self.dnn(V, task_in[t], tower.dnn, tower.tower_name, l2) if tower.HasField('dnn') else task_in[t]
      pred['logits_%s' % tower.tower_name] = self.dense(V, h, tower.num_class, 'dnn_output_%d' % t, l2).squeeze(1)
    return pred

  def _dbmtl(self, V, batch):
    """model/dbmtl.py:46-116: bottom (dnn) -> optional MMoE block -> tower dnn -> relation dnn over
    [own features, earlier towers' relation features] -> output."""
    mc = self.cfg.model_config
    c = mc.dbmtl
    l2 = self._l2_of(mc)
    x, _ = self.input_layer(V, batch, 'all', 'input_layer')
    if c.HasField('bottom_dnn'):
      x = self.dnn(V, x, c.bottom_dnn, 'bottom_dnn', l2)
    if c.HasField('expert_dnn'):
      task_in = self._mmoe_layer(V, x, [c.expert_dnn] * c.num_expert, len(c.task_towers), l2)
    else:
      task_in = [x] * len(c.task_towers)
    rel, pred = {}, {}
    for tower, x_t in zip(c.task_towers, task_in):
      nm = tower.tower_name
      own = self.dnn(V, x_t, tower.dnn, nm + '/dnn', l2) if tower.HasField('dnn') else x_t
      inp = torch.cat([own] + [rel[r] for r in tower.relation_tower_names], dim=-1)
      rel[nm] = self.dnn(V, inp, tower.relation_dnn, nm + '/relation_dnn', l2)
      pred['logits_%s' % nm] = self.dense(V, rel[nm], tower.num_class, nm + '/output', l2).squeeze(1)
    return pred

  # ------------------------------------------------------------------ backbone (RankModel)
  def _keras_mlp(self, V, x, p, name, l2):
    """layers/keras/blocks.py:37-128: Dense(use_bias=False, he_uniform) -> BatchNorm -> activation per layer;
    the LAST layer has BN too (use_final_bn default true) and `final_activation` (the layer's default is none; a
    proto-configured MLP that leaves it unset gets the proto default 'relu', see opt below)."""
    def opt(field, default):
      # Parameter.get_or_default on a proto message (layers/utils.py:222-236): anything with a length - repeated
      # fields AND strings - is taken when non-empty, so an unset string field yields its non-empty PROTO default
      # (final_activation 'relu', protos/dnn.proto:26); other scalars only when set
      value = getattr(p, field)
      if hasattr(value, '__len__'):
        return value if len(value) > 0 else default
      return value if p.HasField(field) else default
    units = list(p.hidden_units)
    use_bn, use_final_bn = opt('use_bn', True), opt('use_final_bn', True)
    use_bias, use_final_bias = opt('use_bias', False), opt('use_final_bias', False)
    act, final_act = opt('activation', 'relu'), opt('final_activation', None)
    for i, u in enumerate(units):
      last = i == len(units) - 1
      lname = '%s/layer_%d' % (name, i)
      w = V.get(lname + '/dense/kernel', l2=l2)
      x = x @ w
      if (use_final_bias if last else use_bias):
        x = x + V.get(lname + '/dense/bias')
      if (use_final_bn if last else use_bn):
        x = self.batch_norm(V, x, lname + '/bn')
      a = final_act if last else act
      if a and a.lower() not in ('linear',):
        if a.lower() == 'dice':
          x = self.dice(V, x, lname + '/act')
        else:
          assert a.lower() in ('relu', 'tf.nn.relu', 'nn.relu'), a
          x = torch.relu(x)
    return x

  def _keras_din(self, V, keys, seq_len, query, cfg, l2, name='din'):
    """layers/keras/din.py:13-67: attention MLP `din_attention` (last layer: bias, no BN, linear) over
    [q, h, q - h, q * h]; -2^32 + 1 on padding; softmax | sigmoid(score / sqrt(E)); scores @ h (+ the target)."""
    import copy
    B, L, E = keys.shape
    qd = query.shape[-1]
    q = query if qd == E else torch.nn.functional.pad(query, (0, E - qd))
    cur = q[:, None, :].expand(B, L, E)
    din_all = torch.cat([cur, keys, cur - keys, cur * keys], dim=-1)
    att = copy.deepcopy(cfg.attention_dnn)
    att.use_final_bn, att.use_final_bias, att.final_activation = False, True, 'linear'
    scores = self._keras_mlp(V, din_all, att, name + '/din_attention', l2).reshape(B, 1, L)
    mask = (torch.arange(L)[None, :] < torch.as_tensor(seq_len)[:, None])[:, None, :]
    scores = torch.where(mask, scores, torch.full_like(scores, float(-2**32 + 1)))
    if cfg.attention_normalizer == 'softmax':
      scores = torch.softmax(scores, dim=-1)
    else:
      assert cfg.attention_normalizer == 'sigmoid'
      scores = torch.sigmoid(scores / (E ** 0.5))
    out = torch.matmul(scores, keys[:, :, :qd] if qd < E else keys).reshape(B, qd)
    return torch.cat([out, q], dim=-1) if cfg.need_target_feature else out  # (q: the PADDED target, din.py:36, 63)

  def _seq_block(self, V, batch, group_name, scope):
    """input_layer { output_seq_and_normal_feature } (layers/common_layers.py:119-131, layers/input_layer.py:164-200):
    sequence columns under `input_layer/<column>` (time axis kept, batch-max padded), then the plain columns through a
    regular input-layer call; embedding L2 on both."""
    g = [x for x in self.cfg.model_config.feature_groups if x.group_name == group_name][0]
    lam = self.cfg.model_config.embedding_regularization
    seqs, seq_len, plain_names = [], None, []
    for n in g.feature_names:
      fc = self.fc_by_name[n]
      if fc.feature_type == fc.SequenceFeature:
        e, lens = self._seq_lookup(V, 'input_layer/%s/embedding_weights' % n, batch, n)
        if lam > 0:
          self._reg = self._reg + lam * 0.5 * (e * e).sum()
        seqs.append(e)
        seq_len = lens if seq_len is None else seq_len
      else:
        plain_names.append(n)
    target = None
    if plain_names:
      target, _ = self.input_layer(V, batch, group_name, scope, only=plain_names)
    return torch.cat(seqs, dim=-1), seq_len, target

  def _keras_cin(self, V, x0, hidden_sizes, name):
    """keras CIN (layers/keras/interaction.py:370-409): per layer the outer product of x_k and x_0 along the field
    axes, weighted by cin_kernel_k [H_k+1, H_k, H_0] and summed over both, + bias, relu; the feature maps summed over the
    embedding axis and concatenated."""
    if isinstance(x0, (list, tuple)):
      x0 = torch.stack(list(x0), dim=1)
    xi, pooled = x0, []
    for k, hk in enumerate(hidden_sizes):
      w = V.get('%s/cin_kernel_%d' % (name, k))
      b = V.get('%s/cin_bias_%d' % (name, k))
      inter = x0[:, None, :, :] * xi[:, :, None, :]                      # [B, H_k, H_0, D]
      fm = (inter[:, None, :, :, :] * w[None, :, :, :, None]).sum(dim=3).sum(dim=2)  # [B, H_k+1, D]
      fm = torch.relu(fm + b[None, :, None])
      pooled.append(fm.sum(dim=-1))
      xi = fm
    return torch.cat(pooled, dim=-1)

  def _keras_cross(self, V, x0, x, st_params, name):
    """layers/keras/interaction.py:249-286: x0 * (W x + b [+ diag_scale x]) + x; W = U V when projection_dim."""
    proj = int(st_params['projection_dim']) if 'projection_dim' in st_params else None
    diag = float(st_params['diag_scale']) if 'diag_scale' in st_params else 0.0
    if proj is None:
      prod = x @ V.get(name + '/dense/kernel') + V.get(name + '/dense/bias')
    else:
      prod = (x @ V.get(name + '/dense_u/kernel')) @ V.get(name + '/dense_v/kernel') + V.get(name + '/dense/bias')
    if diag:
      prod = prod + diag * x
    return x0 * prod + x

  def _keras_mmoe(self, V, x, cfg, name, l2):
    """layers/keras/multi_task.py:18-68: `num_expert` keras MLPs `expert_i` over the input (or, without expert_mlp, the
    first num_expert entries of the input list as experts and the entry after them as the gates' input); per task a
    Dense(num_expert, softmax) gate `gate_t`; the list of the tasks' mixtures."""
    E, T = cfg.num_expert, cfg.num_task
    if E == 0:
      return x
    if cfg.HasField('expert_mlp'):
      experts = [self._keras_mlp(V, x, cfg.expert_mlp, '%s/expert_%d' % (name, i), l2) for i in range(E)]
      gate_in = x
    else:
      experts, gate_in = list(x)[:E], x[E]
    stacked = torch.stack(experts, dim=1)
    out = []
    for t in range(T):
      gate = torch.softmax(self.dense(V, gate_in, E, '%s/gate_%d' % (name, t), l2), dim=-1)
      out.append((stacked * gate[:, :, None]).sum(dim=1))
    return out

  def _standard_keras(self, V, x, kl, name):
    """tensorflow.keras.layers.Dense / Activation / Dropout named directly in a backbone (st_params = constructor
    keyword arguments, backbone.py:381-397)"""
    p = kl.st_params
    act = p['activation'] if 'activation' in p else None
    if kl.class_name == 'Dense':
      x = self.dense(V, x, int(p['units']), name, 0.0)  # (no kernel regulariser: none is passed)
    elif kl.class_name == 'Dropout':
      assert float(p['rate']) == 0.0, 'the oracle has no random dropout'
      return x
    if act in (None, 'linear'):
      return x
    return {'relu': torch.relu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh,
            'softmax': lambda t: torch.softmax(t, dim=-1)}[act](x)

  def _keras_senet(self, V, inputs, cfg, name):
    """layers/keras/fibinet.py:15-92: per field and squeeze group the max and the mean over the group's columns ->
    Dense W1 (relu, with bias) -> Dense W2 (sum of the dims) -> re-weight the concatenated embeddings (+ skip
    connection, + LayerNormalization `output_ln`, epsilon 1e-3)."""
    g = int(cfg.num_squeeze_group)
    inputs = list(inputs)
    sq = []
    for emb in inputs:
      grouped = emb.reshape(emb.shape[0], g, -1)
      sq.append(grouped.max(dim=-1).values)
      sq.append(grouped.mean(dim=-1))
    z = torch.cat(sq, dim=1)
    emb_size = sum(int(e.shape[-1]) for e in inputs)
    red = max(1, len(inputs) * g * 2 // int(cfg.reduction_ratio))
    a1 = torch.relu(self.dense(V, z, red, name + '/W1', 0.0))
    w = self.dense(V, a1, emb_size, name + '/W2', 0.0)
    x = torch.cat(inputs, dim=-1)
    out = x * w
    if cfg.use_skip_connection:
      out = out + x
    if cfg.use_output_layer_norm:
      mean = out.mean(dim=-1, keepdim=True)
      var = ((out - mean) ** 2).mean(dim=-1, keepdim=True)
      out = (out - mean) / torch.sqrt(var + 1e-3) * V.get(name + '/output_ln/gamma') + V.get(name + '/output_ln/beta')
    return out

  def _rank_backbone(self, V, batch):
    """model/rank_model.py:38-55: the backbone's output, top_mlp aside, + the head `output` dense when its width is
    not num_class."""
    mc = self.cfg.model_config
    out = self._backbone(V, batch)
    if out.shape[-1] != mc.num_class:
      out = self.dense(V, out, mc.num_class, 'output', 0.0)
    return {'logits': out.squeeze(1)}

  def _multi_task_backbone(self, V, batch):
    """model/multi_task_model.py:33-100 (model_class MultiTaskModel): the backbone yields one input per tower or one
    shared input; tower DNN `<tower>`, Bayes relation DNN `<tower>/relation_dnn` over [own features, the named
    earlier towers' relation features], `<tower>/output`."""
    mc = self.cfg.model_config
    l2 = mc.model_params.l2_regularization
    towers = list(mc.model_params.task_towers)
    shared = self._backbone(V, batch)
    inputs = list(shared) if isinstance(shared, (list, tuple)) else [shared] * len(towers)
    assert len(inputs) == len(towers), 'The number of backbone outputs and task towers must be equal'
    feats, rel, pred = {}, {}, {}
    for tower, x in zip(towers, inputs):
      feats[tower.tower_name] = self.dnn(V, x, tower.dnn, tower.tower_name, l2) if tower.HasField('dnn') else x
    for tower in towers:
      nm = tower.tower_name
      x = feats[nm]
      if tower.HasField('relation_dnn'):
        x = torch.cat([x] + [rel[r] for r in tower.relation_tower_names], dim=-1)
        x = rel[nm] = self.dnn(V, x, tower.relation_dnn, nm + '/relation_dnn', l2)
      pred['logits_%s' % nm] = self.dense(V, x, tower.num_class, nm + '/output', l2).squeeze(1)
    return pred

  def _backbone(self, V, batch):
    """layers/backbone.py: blocks in config order (the shipped configs list them
    topologically), feature-group inputs -> input layer, `input_fn` lambdas, keras_layer MLP / Cross, recurrent
    with a fixed input, concat_blocks, top_mlp."""
    mc = self.cfg.model_config
    bb = mc.backbone
    l2 = mc.model_params.l2_regularization
    outs, scope_id = {}, 0
    groups = {g.group_name for g in mc.feature_groups}

    def group_out(name):
      nonlocal scope_id
      if name not in outs:
        scope = 'input_layer' if scope_id == 0 else 'input_layer_%d' % scope_id
        scope_id += 1
        outs[name] = self.input_layer(V, batch, name, scope)[0]
      return outs[name]

    # the input-layer calls happen in topological order with implicit group blocks first (backbone.py:160-186)
    for blk in bb.blocks:
      if blk.WhichOneof('layer') == 'input_layer':
        continue
      for node in blk.inputs:
        if node.WhichOneof('name') == 'feature_group_name' and node.feature_group_name in groups:
          group_out(node.feature_group_name)
    for blk in bb.blocks:
      if blk.WhichOneof('layer') == 'input_layer':
        gname = blk.inputs[0].feature_group_name
        if blk.input_layer.output_seq_and_normal_feature:
          has_plain = any(self.fc_by_name[n].feature_type != self.fc_by_name[n].SequenceFeature
                          for g in mc.feature_groups if g.group_name == gname for n in g.feature_names)
          scope = None
          if has_plain:
            scope = 'input_layer' if scope_id == 0 else 'input_layer_%d' % scope_id
            scope_id += 1
          outs[blk.name] = self._seq_block(V, batch, gname, scope)
        else:
          scope = 'input_layer' if scope_id == 0 else 'input_layer_%d' % scope_id
          scope_id += 1
          fea, flist = self.input_layer(V, batch, gname, scope)
          il = blk.input_layer
          outs[blk.name] = flist if il.only_output_feature_list else ((fea, flist) if il.output_2d_tensor_and_feature_list else fea)
        continue
      ins = []
      for node in blk.inputs:
        fea = outs[getattr(node, node.WhichOneof('name'))]
        if node.HasField('input_slice'):
          fea = eval('lambda x: x' + node.input_slice.strip())(fea)
        if node.HasField('input_fn'):
          fea = eval(node.input_fn)(fea)
        ins.append(fea)
      if blk.merge_inputs_into_list:
        x = ins
      elif len(ins) == 1:
        x = ins[0]
      elif any(isinstance(v, list) for v in ins):  # (backbone.py merge_inputs: lists are merged into one list)
        x = [e for v in ins for e in (v if isinstance(v, list) else [v])]
      else:
        x = torch.cat(ins, dim=-1)
      if blk.HasField('extra_input_fn'):  # (backbone.py:424-441: applied to the merged input)
        x = eval(blk.extra_input_fn)(x)
      kind = blk.WhichOneof('layer')
      if kind == 'keras_layer':
        kl = blk.keras_layer
        if kl.class_name == 'MLP':
          x = self._keras_mlp(V, x, kl.mlp, blk.name, l2)
        elif kl.class_name == 'Cross':
          x = self._keras_cross(V, x[0], x[1], kl.st_params, blk.name)
        elif kl.class_name == 'DIN':
          x = self._keras_din(V, x[0], x[1], x[2], kl.din, l2, blk.name)
        elif kl.class_name == 'Add':
          x = _TF.add_n(list(x))
        elif kl.class_name == 'DotInteraction':
          # layers/keras/interaction.py:90-127: lower triangle of the F x F dot products, row by row
          fl = torch.stack(list(x), dim=1)
          inter = fl @ fl.transpose(1, 2)
          self_inter = 'self_interaction' in kl.st_params and bool(kl.st_params['self_interaction'])
          F = fl.shape[1]
          x = torch.stack([inter[:, i, j] for i in range(F) for j in range(i + 1 if self_inter else i)], dim=1)
        elif kl.class_name == 'CIN':
          x = self._keras_cin(V, x, [int(h) for h in kl.cin.hidden_feature_sizes], blk.name)
        elif kl.class_name == 'MMoE':
          x = self._keras_mmoe(V, x, kl.mmoe, blk.name, l2)
        elif kl.class_name in ('Dense', 'Activation', 'Dropout'):
          x = self._standard_keras(V, x, kl, blk.name)
        elif kl.class_name == 'SENet':
          x = self._keras_senet(V, x, kl.senet, blk.name)
        elif kl.class_name == 'FM':
          fl = torch.stack(list(x), dim=1)  # [B, F, D]
          x = 0.5 * (fl.sum(dim=1) ** 2 - (fl ** 2).sum(dim=1))
          use_variant = 'use_variant' in kl.st_params and bool(kl.st_params['use_variant'])
          x = x if use_variant else x.sum(dim=1, keepdim=True)
        else:
          raise NotImplementedError(kl.class_name)
      elif kind == 'lambda':
        tf = _TF
        x = eval(getattr(blk, 'lambda').expression)(x)
      elif kind == 'recurrent':
        rc = blk.recurrent
        assert rc.keras_layer.class_name == 'Cross' and rc.fixed_input_index == 0
        x0, xi = x
        for i in range(rc.num_steps):
          xi = self._keras_cross(V, x0, xi, rc.keras_layer.st_params, '%s_%d' % (blk.name, i))
        x = xi
      elif kind is None:
        pass  # a block without a layer hands its (merged, transformed) input on (backbone.py:262-268)
      else:
        raise NotImplementedError(kind)
      outs[blk.name] = x
    concat = list(bb.concat_blocks)
    if not concat:  # no concat_blocks / output_blocks: every leaf block, in config order (backbone.py:187-196)
      used = {getattr(node, node.WhichOneof('name')) for blk in bb.blocks for node in blk.inputs}
      concat = [blk.name for blk in bb.blocks if blk.name not in used]
    # merge_inputs (backbone.py:532-550): one output is handed on as it is (a list stays a list)
    out = torch.cat([outs[n] for n in concat], dim=-1) if len(concat) > 1 else outs[concat[0]]
    if bb.HasField('top_mlp'):
      out = self._keras_mlp(V, out, bb.top_mlp, 'backbone_top_mlp', l2)
    return out

  # ------------------------------------------------------------------ one step
  def probe_ids(self, batches, table_shapes):
    """{table name: ascending ids} the lookups of `batches` read: the forward pass with every embedding table of
    table_shapes {name: (rows, dim)} standing in as ONE zero row and the lookups recording what they are asked for.  What
    bench.py needs to fetch only those rows of a table too large for the host (compact_ids)."""
    self._probe, self._probe_seen = dict(table_shapes), {}
    try:
      with torch.no_grad():
        for b in batches:
          self.forward(b)
    finally:
      self._probe = None
    return {k: np.unique(np.concatenate(v)) for k, v in self._probe_seen.items()}

  def forward(self, batch):
    V = Vars(self.state, self.dtype, compact=self.compact_ids, probe=self._probe)
    self._reg = torch.zeros((), dtype=self.dtype)
    self._moving = {}
    self._touched = {}
    self._cache = (self._hashed_ids(batch), self._raw_values(batch), self._int_ids(batch))
    labels_np = np.asarray(batch['labels'], dtype=np.float32)
    def ce_of(z, y):
      # tf.nn.sigmoid_cross_entropy_with_logits as TensorFlow writes it - relu and -|z| through `where(z >= 0, ...)` - so
      # that autograd gives sigmoid(z) - y at z == 0 exactly too (clamp / abs have subgradient 1 / 0 there: a logit that is
      # exactly 0 - a row whose last hidden layer is all-dead under a zero bias - would get 0 - y instead of 0.5 - y)
      pos = z >= 0
      relu_z = torch.where(pos, z, torch.zeros_like(z))
      neg_abs = torch.where(pos, -z, z)
      return (relu_z - z * y + torch.log1p(torch.exp(neg_abs))).mean()
    losses = OrderedDict()
    from easyrec_amd.protos.loss_pb2 import LossType  # (the config schema is the shared boundary)

    def one_loss(loss_type, param, z, labels):
      if loss_type == LossType.F1_REWEIGHTED_LOSS:
        # loss/f1_reweight_loss.py:10-39 (beta^2 = 1 without loss_param, builders/loss_builder.py:198-206): negatives
        # weighted by tp / (beta^2 #pos + #neg - tn + 1e-8), tp = sum of probabilities - a function of the logits
        # whose gradient TensorFlow keeps
        beta2 = 1.0 if param is None else float(param.f1_beta_square)
        B = float(labels.shape[0])
        tp = torch.sigmoid(z).sum()
        neg_w = tp / (beta2 * labels.sum() + (B - labels.sum()) - (B - tp) + 1e-8)
        w = torch.where(labels == 1.0, torch.ones_like(z), neg_w.expand_as(z))
        pos = z >= 0
        per = torch.where(pos, z, torch.zeros_like(z)) - z * labels + torch.log1p(torch.exp(torch.where(pos, -z, z)))
        return 'f1_reweighted_loss', (w * per).sum() / (w != 0).sum().clamp(min=1)
      if loss_type == LossType.PAIR_WISE_LOSS:
        # loss/pairwise_loss.py:15-70: sigmoid CE of z_i - z_j - margin against 1 over the pairs label_i > label_j
        margin = 0.0 if param is None else float(param.margin)
        temp = 1.0 if param is None else float(param.temperature)
        zz = z / temp if temp != 1.0 else z
        x = (zz[:, None] - zz[None, :] - margin)[labels[:, None] > labels[None, :]]
        per = torch.where(x >= 0, x, torch.zeros_like(x)) - x + torch.log1p(torch.exp(torch.where(x >= 0, -x, x)))
        return 'pair_wise_loss', per.sum() / max(int(x.numel()), 1)
      if loss_type in (LossType.L2_LOSS, LossType.SIGMOID_L2_LOSS):
        # rank_model.py:123-128 (y = the output column, through a sigmoid for SIGMOID_L2_LOSS) and
        # builders/loss_builder.py:52-56: tf.losses.mean_squared_error of y against the float label
        y = torch.sigmoid(z) if loss_type == LossType.SIGMOID_L2_LOSS else z
        return 'l2_loss', ((y - labels) ** 2).mean()
      assert loss_type in (LossType.CLASSIFICATION, LossType.BINARY_CROSS_ENTROPY_LOSS), loss_type
      # tf.losses.sigmoid_cross_entropy (SUM_BY_NONZERO_WEIGHTS, weights = 1.0)
      return 'cross_entropy_loss', ce_of(z, labels)


    if self.model_class in ('MMoE', 'SimpleMultiTask', 'PLE', 'DBMTL', 'MultiTaskModel'):
      fn, sub = {'MMoE': (self._mmoe, 'mmoe'), 'SimpleMultiTask': (self._simple_multi_task, 'simple_multi_task'),
                 'PLE': (self._ple, 'ple'), 'DBMTL': (self._dbmtl, 'dbmtl'),
                 'MultiTaskModel': (self._multi_task_backbone, 'model_params')}[self.model_class]
      pred = fn(V, batch)
      towers = getattr(self.cfg.model_config, sub).task_towers
      label_fields = list(self.cfg.data_config.label_fields)
      ce = torch.zeros((), dtype=self.dtype)
      for t, tower in enumerate(towers):
        lname = tower.label_name if tower.HasField('label_name') else label_fields[t]
        y = torch.as_tensor(labels_np[label_fields.index(lname)], dtype=self.dtype)
        z = pred['logits_%s' % tower.tower_name]
        pred['probs_%s' % tower.tower_name] = torch.sigmoid(z)
        suffix = '_%s' % tower.tower_name
        if len(tower.losses) == 0:
          # tf.losses.sigmoid_cross_entropy (SUM_BY_NONZERO_WEIGHTS) x task weight (multi_task_model.py:229-240)
          name, li = one_loss(tower.loss_type, None, z, y)
          losses[name + suffix] = li = li * tower.weight
          ce = ce + li
          continue
        # a tower's `losses` list (multi_task_model.py:241-269): every entry times the FIRST entry's weight x the tower
        # weight - the reference indexes the weights by the position inside the one-entry dict its loss builder returns
        first = tower.losses[0].weight * tower.weight
        for entry in tower.losses:
          which = entry.WhichOneof('loss_param')
          name, li = one_loss(entry.loss_type, getattr(entry, which) if which else None, z, y)
          plain = entry.loss_type in (LossType.CLASSIFICATION, LossType.BINARY_CROSS_ENTROPY_LOSS)
          key = (entry.loss_name + ('' if plain else suffix)) if entry.loss_name else name + suffix
          losses[key] = li = li * first
          ce = ce + li
    else:
      if self.model_class == 'DeepFM':
        pred = self._deepfm(V, batch)
      elif self.model_class == 'DCN':
        pred = self._dcn(V, batch)
      elif self.model_class == 'MultiTowerDIN':
        pred = self._multi_tower_din(V, batch)
      elif self.model_class == 'RankModel':
        pred = self._rank_backbone(V, batch)
      elif self.model_class == 'WideAndDeep':
        pred = self._wide_and_deep(V, batch)
      elif self.model_class == 'FM':
        pred = self._fm(V, batch)
      elif self.model_class == 'MultiTower':
        pred = self._multi_tower(V, batch)
      elif self.model_class == 'DLRM':
        pred = self._dlrm(V, batch)
      else:
        raise NotImplementedError('oracle: model_class %s' % self.model_class)
      labels = torch.as_tensor(labels_np[0], dtype=self.dtype)
      z = pred['logits']
      mc = self.cfg.model_config
      if len(mc.losses):  # rank_model.py:269-300, Fixed strategy: each loss times its weight
        ce = torch.zeros((), dtype=self.dtype)
        for entry in mc.losses:
          which = entry.WhichOneof('loss_param')
          name, value = one_loss(entry.loss_type, getattr(entry, which) if which else None, z, labels)
          value = value * entry.weight
          losses[entry.loss_name or name] = value
          ce = ce + value
      else:
        name, ce = one_loss(mc.loss_type, None, z, labels)
        losses[name] = ce
      pred['probs'] = torch.sigmoid(z)
    reg = self._reg
    for name, t in V.used.items():
      if V.l2[name] > 0:
        reg = reg + V.l2[name] * 0.5 * (t * t).sum()
    losses['regularization_loss'] = reg
    losses['total_loss'] = ce + reg
    return V, pred, losses

  def train_step(self, batch):
    return self.train_step_world([batch])[0]

  def train_step_world(self, batches):
    """One optimisation step of W = len(batches) data-parallel workers with row-sharded embedding tables, the
    reference's EmbeddingParallelStrategy (compat/optimizers.py:285-345): every worker runs forward / backward on ITS
    batch (BatchNorm statistics per worker: the reference does not synchronise them), dense gradients are averaged
    (hvd.allreduce(op=Average), :328-331), the row gradients of all workers meet at the rows' owners through the
    all-to-all's backward, summed, and are divided by the world size (:315-316) - i.e. every gradient is the mean over
    the workers - and ONE optimizer step is applied to the (logically single) set of variables.  Returns the list of
    per-worker loss dicts.  W = 1 is plain single-GPU training."""
    W = len(batches)
    per_rank, losses_out, touched = [], [], {}
    self.rank_moving = []
    clip_on = float(self.cfg.train_config.gradient_clipping_by_norm) > 0
    lookup_sq = {}  # W == 1: variable name -> sum over its lookups of the squared per-lookup gradient (shared tables)
    if self.kv and W > 1:
      self._kv_mode, self._track_lookups = 'insert', False
      with torch.no_grad():
        for batch in batches:
          self.forward(batch)
      self._kv_mode = 'find'
    for batch in batches:
      self._track_lookups, self._lookup_leaves = clip_on and W == 1, {}
      V, pred, losses = self.forward(batch)
      losses['total_loss'].backward()
      for name, t in V.used.items():
        leaves = self._lookup_leaves.get(id(t), [])
        if len(leaves) > 1:
          lookup_sq[name] = sum(float((lf.grad.double() ** 2).sum()) for lf in leaves if lf.grad is not None)
      self._track_lookups = False
      gr = OrderedDict()
      for name, t in V.used.items():
        if t.requires_grad:
          gr[name] = np.zeros(t.shape, dtype=np.float32) if t.grad is None else t.grad.numpy().astype(np.float32)
      per_rank.append(gr)
      for name, t in V.used.items():
        mask = self._touched.get(id(t))
        if mask is not None:
          touched[name] = mask if name not in touched else (touched[name] | mask)
      self.rank_moving.append({k: val.numpy().astype(np.float32) for k, val in self._moving.items()})
      losses_out.append({k: float(v.detach()) for k, v in losses.items()})
      self.last_pred = {k: v.detach().numpy() for k, v in pred.items()}
    self._kv_mode = 'both'
    names = list(per_rank[0].keys())
    step = self.global_step
    # gradients after the multipliers (compat/optimizers.py:347-356); mean over the workers
    grads = OrderedDict()
    for name in names:
      g = per_rank[0][name]
      for r in range(1, W):
        g = (g + per_rank[r][name]).astype(np.float32)
      if W > 1:
        g = (g * F32(1.0 / W)).astype(np.float32)
      if name.endswith('/embedding_weights') and self.emb_mult != 1.0:
        g = (g * F32(self.emb_mult)).astype(np.float32)
      grads[name] = g
    self.last_grads = OrderedDict((n, g.copy()) for n, g in grads.items())
    # test hook (tests/test_chaos_bars.py): a model of ANOTHER fp32 summation order - callable (name, g) -> g' applied to
    # every gradient before the optimizer sees it; None in every other use
    if getattr(self, 'grad_noise', None) is not None:
      grads = OrderedDict((n, self.grad_noise(n, g).astype(np.float32)) for n, g in grads.items())
    # clip_by_global_norm (:365-376, 453-481): norm = sqrt(2 * sum of tf.nn.l2_loss(g)); a table's IndexedSlices carry
    # one row per distinct id of a lookup, so (one lookup per table) the dense gradient has the same sum of squares; with
    # sharded tables the rows of different workers stay separate rows of `values` (each divided by W), their l2 sums are
    # all-reduced.  Tables shared by several lookups keep per-lookup rows in TF (W == 1: `lookup_sq`, the sum over the
    # lookups of their own squared gradients; under embedding parallelism the ids of all features of a table are
    # de-duplicated together before the exchange, feature_column.py:259-289, so a rank's rows ARE merged).
    clip = float(self.cfg.train_config.gradient_clipping_by_norm)
    self.last_grad_norm = None
    if clip > 0:
      sq = 0.0
      for name in names:
        if name.endswith('/embedding_weights') and W > 1:
          for r in range(W):
            gr = per_rank[r][name].astype(np.float64) * (self.emb_mult / W)
            sq += float((gr ** 2).sum())
        elif name in lookup_sq:
          sq += lookup_sq[name] * float(self.emb_mult) ** 2
        else:
          sq += float((grads[name].astype(np.float64) ** 2).sum())
      norm = F32(np.sqrt(sq))
      with np.errstate(divide='ignore'):
        scale = F32(clip) * min(F32(1.0) / norm, F32(1.0) / F32(clip))
      grads = OrderedDict((n, (g * scale).astype(np.float32)) for n, g in grads.items())
      self.last_grad_norm = float(norm)
    for name in names:
      is_emb = name.endswith('/embedding_weights')
      oi = 0 if (is_emb or len(self.opt) == 1) else 1
      o = self.opt[oi]
      g = grads[name]
      lr = self._lr(o['lr_cfg'], step)
      var = self.state[name]
      if o['kind'] in ('adam_optimizer', 'lazy_adam_optimizer'):
        b1, b2 = F32(o['beta1']), F32(o['beta2'])
        b1p, b2p = self.beta_pow[oi]
        one = F32(1.0)
        lr_t = F32(lr * np.sqrt(one - b2p) / (one - b1p))
        m = self.slots.setdefault(name + '/m', np.zeros_like(var))
        v = self.slots.setdefault(name + '/v', np.zeros_like(var))
        eps = F32(1e-8)
        if is_emb:
          # python-graph sparse apply (IndexedSlices): tf AdamOptimizer decays every row;
          # AdamOptimizerS (lazy) only the rows present in the gradient.
          if o['kind'] == 'lazy_adam_optimizer':
            rows = np.flatnonzero(touched.get(name, np.zeros(var.shape[0], dtype=bool)))
            m_t = m[rows] * b1 + g[rows] * (one - b1)
            v_t = v[rows] * b2 + (g[rows] * g[rows]) * (one - b2)
            var[rows] = var[rows] - (lr_t * m_t) / (np.sqrt(v_t, dtype=np.float32) + eps)
            m[rows], v[rows] = m_t, v_t
          else:
            # every row: the same fp32 operations in the same order, as in-place multi-threaded torch-CPU
            # passes over the numpy buffers (this is the timed CPU baseline of bench.py)
            tm, tv, tvar, tg = (torch.from_numpy(a) for a in (m, v, var, g))
            tm.mul_(float(b1)).add_(tg * float(one - b1))
            tv.mul_(float(b2)).add_((tg * tg) * float(one - b2))
            tvar.sub_((tm * float(lr_t)) / (tv.sqrt() + float(eps)))
        else:
          # training_ops.apply_adam
          m[:] = m + (g - m) * (one - b1)
          v[:] = v + (g * g - v) * (one - b2)
          var[:] = var - (m * lr_t) / (np.sqrt(v, dtype=np.float32) + eps)
      elif o['kind'] == 'adagrad_optimizer':
        # tf.train.AdagradOptimizer (builders/optimizer_builder.py:110-116): accumulator starts at
        # initial_accumulator_value; accum += g^2; var -= lr * g / sqrt(accum).  A table's sparse apply touches the rows
        # of the gradient only - the same arithmetic, since untouched rows have g = 0.
        acc = self.slots.setdefault(name + '/v', np.full_like(var, F32(o['initial_accumulator_value'])))
        acc[:] = acc + g * g
        var[:] = var - (F32(lr) * g) / np.sqrt(acc, dtype=np.float32)
      else:
        raise NotImplementedError(o['kind'])
    for oi, o in enumerate(self.opt):
      if o['kind'] in ('adam_optimizer', 'lazy_adam_optimizer'):
        self.beta_pow[oi][0] = F32(self.beta_pow[oi][0] * F32(o['beta1']))
        self.beta_pow[oi][1] = F32(self.beta_pow[oi][1] * F32(o['beta2']))
    # BatchNorm moving statistics live per worker; `state` follows worker 0 (rank_moving has all of them)
    for k, val in self.rank_moving[0].items():
      self.state[k] = val
    self.global_step += 1
    return losses_out

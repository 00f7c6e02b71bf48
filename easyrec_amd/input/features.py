"""Device-resident feature "placeholders" and the host->device batch layout.

The reference's model constructors take a `features` dict of TF tensors produced by
`Input._preprocess` (easy_rec/python/input/input.py:806-939).  Here `features` is a
`DeviceFeatures`: persistent HBM buffers with fixed addresses (so the whole training step can be
captured in a hipGraph) that are refilled for every batch by `load()`.  The per-feature
representation follows the reference's parsed dict:

  IdFeature + hash_bucket_size : packed utf-8 bytes + offsets  -> hashed ON DEVICE (er_hash_bucket_fast)
                                 or pre-hashed int64 ids (host hashing in data-loader threads)
  IdFeature + num_buckets/vocab: int64 ids, out-of-range / OOV already mapped to default 0
  RawFeature                   : fp32, min/max-normalised (input.py:638-640); with embedding_dim>0 the
                                 value is the weight of projection id 0..raw_input_dim-1 (input.py:648-673)
  TagFeature                   : CSR ids (+ weights)            (input.py:432-505)
  SequenceFeature              : [B, L] ids padded with -1 + lengths (input.py:677-804)
"""
from collections import OrderedDict

import numpy as np
import torch

from easyrec_amd.protos.feature_config_pb2 import FeatureConfig


def feature_name_of(fc):
  return fc.feature_name if fc.HasField('feature_name') else fc.input_names[0]


def raw_boundaries(fc):
  """Sorted float32 bucket boundaries of a RawFeature, or None (reference feature_column/feature_column.py:365-376:
  explicit `boundaries`, or `num_buckets` equal-width buckets of the [0, 1]-normalised value)."""
  if fc.raw_input_dim > 1:
    return None
  if len(fc.boundaries) > 0:
    return np.sort(np.asarray(list(fc.boundaries), dtype=np.float32))
  if fc.num_buckets > 1 and fc.max_val > fc.min_val:
    return np.asarray([x / float(fc.num_buckets) for x in range(0, fc.num_buckets)], dtype=np.float32)
  return None


def bucketize(x, bounds):
  """tf bucketize (BucketizedColumn, feature_column_v2.py:2762-2916): index = number of boundaries <= x."""
  return np.searchsorted(bounds, np.asarray(x, dtype=np.float32), side='right').astype(np.int64)


MAX_HASH_BUCKET_SIZE = 9223372036854775807  # (feature_column/feature_column.py:19)


class FeatureSchema(object):
  """Static description of what a batch contains, derived from the config."""

  def __init__(self, data_config, feature_configs, batch_size=None, max_tag_len=16,
               max_seq_len=50, max_str_bytes=32):
    self.batch_size = int(batch_size or data_config.batch_size)
    self.label_fields = list(data_config.label_fields)
    self.sample_weight = data_config.sample_weight if data_config.HasField('sample_weight') else None
    self.raw = OrderedDict()      # name -> dict(dim, row)  rows of the raw block
    self.raw_multi = OrderedDict()  # name -> k (raw_input_dim > 1)
    self.hash_single = OrderedDict()   # name -> dict(buckets, col)
    self.int_single = OrderedDict()    # name -> dict(col, num_buckets)
    self.tags = OrderedDict()     # name -> dict(cap, weighted, hash_buckets | None)
    self.seqs = OrderedDict()     # name -> dict(max_len, hash_buckets | None)
    self.feature_configs = {}
    self.max_str_bytes = max_str_bytes
    n_raw_rows = 0
    for fc in feature_configs:
      name = feature_name_of(fc)
      self.feature_configs[name] = fc
      ft = fc.feature_type
      if ft == FeatureConfig.RawFeature:
        if fc.raw_input_dim > 1:
          self.raw_multi[name] = int(fc.raw_input_dim)  # stored example-major [B, k]
        else:
          self.raw[name] = {'dim': 1, 'row': n_raw_rows}
          n_raw_rows += 1
          bounds = raw_boundaries(fc)
          if bounds is not None:
            # bucketized raw feature (feature_column.py:365-386): the value stays in the raw block, the bucket index
            # it falls into is an id column like any other
            self.int_single[name] = {'col': len(self.int_single), 'num_buckets': len(bounds) + 1, 'bounds': bounds}
      elif ft == FeatureConfig.IdFeature:
        if fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0:
          # `ev_params` (hash-table embedding): the id is the hash into the whole int64 range, not into a bucket count
          # (feature_column.py:222-226: MAX_HASH_BUCKET_SIZE), the table holds a row per id actually seen
          buckets = MAX_HASH_BUCKET_SIZE if fc.HasField('ev_params') else int(fc.hash_bucket_size)
          self.hash_single[name] = {'buckets': buckets, 'col': len(self.hash_single)}
        else:
          assert not fc.HasField('ev_params'), 'ev_params on %s: only hashed IdFeatures are hash-table backed here' % name
          nb = len(fc.vocab_list) if fc.vocab_list else int(fc.num_buckets)
          self.int_single[name] = {'col': len(self.int_single), 'num_buckets': nb}
      elif ft == FeatureConfig.TagFeature:
        hb = int(fc.hash_bucket_size) if fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0 else None
        if fc.HasField('ev_params'):
          assert hb is not None, 'ev_params on %s: only hashed TagFeatures are hash-table backed here' % name
          hb = MAX_HASH_BUCKET_SIZE
        weighted = len(fc.input_names) > 1 or fc.HasField('kv_separator')
        self.tags[name] = {'cap': self.batch_size * max_tag_len, 'weighted': weighted, 'hash_buckets': hb}
      elif ft == FeatureConfig.SequenceFeature:
        hb = int(fc.hash_bucket_size) if fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0 else None
        if fc.HasField('ev_params'):
          # (the sequence's embedding column creates its variable like any other: feature_column_v2.py:3616-3640 ->
          # _old_get_dense_tensor_internal -> :3478-3513 get_embedding_variable; the id is the hash into the whole range)
          assert hb is not None, 'ev_params on %s: only hashed SequenceFeatures are hash-table backed here' % name
          hb = MAX_HASH_BUCKET_SIZE
        ml = int(fc.max_seq_len) if fc.HasField('max_seq_len') and fc.max_seq_len > 0 else max_seq_len
        self.seqs[name] = {'max_len': ml, 'hash_buckets': hb}
        if fc.sub_feature_type == FeatureConfig.RawFeature:
          # a sequence of numbers (feature_column.py:521-545): with `boundaries` / `num_buckets` every element is
          # bucketized into an id (sequence_numeric_column_with_bucketized_column) and embedded like an id sequence
          bounds = raw_boundaries(fc)
          assert bounds is not None and hb is None, \
              'SequenceFeature %s: numeric sequences without boundaries / num_buckets are outside the hot-path scope' % name
          self.seqs[name]['bounds'] = bounds
      elif ft == FeatureConfig.ComboFeature and len(fc.combo_join_sep) == 0 and any(len(x) > 0 for x in fc.combo_input_seps):
        # crossed_column over multi-valued inputs (input.py:400-405: tf.string_split by combo_input_seps[i]): every
        # combination of one token per input is an id - a ragged lookup like a TagFeature, ids from the host
        assert fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0, 'ComboFeature %s needs hash_bucket_size' % name
        self.tags[name] = {'cap': self.batch_size * max_tag_len, 'weighted': False, 'hash_buckets': None,
                           'num_buckets': int(fc.hash_bucket_size), 'cross': True}
      elif ft == FeatureConfig.ComboFeature and len(fc.combo_join_sep) == 0:
        # crossed_column (feature_column.py:434-445): the id is computed on the host (sparse_cross_hashed of the
        # inputs' strings, er_sparse_cross_hashed_host) and looked up like any identity column
        assert fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0, 'ComboFeature %s needs hash_bucket_size' % name
        self.int_single[name] = {'col': len(self.int_single), 'num_buckets': int(fc.hash_bucket_size), 'cross': True}
      elif ft == FeatureConfig.LookupFeature:
        # the values of the row's map whose key equals the row's key (input.py:941-1000): a ragged lookup of hashed
        # strings, at most lookup_max_sel_elem_num per row - the TagFeature machinery
        assert fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0, 'LookupFeature %s needs hash_bucket_size' % name
        self.tags[name] = {'cap': self.batch_size * max(int(fc.lookup_max_sel_elem_num), 1), 'weighted': False,
                           'hash_buckets': int(fc.hash_bucket_size)}
      elif ft == FeatureConfig.ComboFeature:
        if fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0:
          self.hash_single[name] = {'buckets': int(fc.hash_bucket_size), 'col': len(self.hash_single)}
      # Expr / PassThrough: handled by the Input class as raw values when used
    self.n_raw_rows = n_raw_rows

  @property
  def hash_buckets_array(self):
    return np.array([v['buckets'] for v in self.hash_single.values()], dtype=np.uint64)


def host_key_column(schema, batch, name):
  """The values of input feature `name` in a HOST batch (numpy arrays, before pack()), one per example, as grouping
  keys for gAUC / session AUC (the reference groups by `feature_dict[uid_field]`, model/rank_model.py:380): the raw
  strings of a hashed id feature, the integers of an identity one."""
  B = schema.batch_size
  if name in schema.hash_single:
    j = schema.hash_single[name]['col']
    if 'str_bytes' in batch:
      data = np.asarray(batch['str_bytes'], dtype=np.uint8).tobytes()
      off = np.asarray(batch['str_offsets']).astype(np.int64)
      return np.array([data[off[j * B + r]:off[j * B + r + 1]] for r in range(B)], dtype=object)
    if 'hash_ids' in batch:  # already hashed on the host: the bucket is all that is left of the value
      return np.asarray(batch['hash_ids'])[j].astype(np.int64)
  if name in schema.int_single and 'int_ids' in batch:
    return np.asarray(batch['int_ids'])[schema.int_single[name]['col']].astype(np.int64)
  raise KeyError('no per-example key column for feature %r in this batch (need a host batch with str_bytes / hash_ids / '
                 'int_ids; a packed batch has left the host)' % name)


class DeviceFeatures(object):
  """Persistent device buffers for one batch; dict-like access by feature name."""

  def __init__(self, schema, device, backend=None):
    self.schema = schema
    self.device = torch.device(device)
    self.version = 0
    self._backend = backend
    B = schema.batch_size
    dev = self.device
    nh = len(schema.hash_single)
    # the fixed-size inputs live back to back in ONE arena so that a packed batch (pack()) is loaded with a single
    # copy: labels | raw | int ids | hash ids | string offsets | string bytes, each section 256-byte aligned
    sections = [('labels', (max(len(schema.label_fields), 1), B), torch.float32),
                ('raw_block', (max(schema.n_raw_rows, 1), B), torch.float32),
                ('int_ids', (max(len(schema.int_single), 1), B), torch.int64),
                ('hash_ids', (max(nh, 1), B), torch.int64),
                ('str_offsets', (nh * B + 1,), torch.int64),
                ('str_bytes', (max(nh * B * schema.max_str_bytes, 16),), torch.uint8)]
    self._layout, off = {}, 0
    for name, shape, dt in sections:
      nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
      self._layout[name] = (off, nbytes, shape, dt)
      off += (nbytes + 255) // 256 * 256
    self.arena = torch.zeros(off, dtype=torch.uint8, device=dev)
    for name, (o, nbytes, shape, dt) in self._layout.items():
      setattr(self, name, self.arena[o:o + nbytes].view(dt).view(shape))
    self.hash_ids.fill_(-1)
    self.sample_weight = None
    self.hash_buckets = torch.from_numpy(schema.hash_buckets_array.astype(np.int64)).to(dev) if nh else None
    self.zero_ids = torch.zeros(B, dtype=torch.int64, device=dev)  # projection id 0 of raw features
    self.raw_multi = {}
    for name, k in schema.raw_multi.items():
      self.raw_multi[name] = {
          'values': torch.zeros(B, k, dtype=torch.float32, device=dev),
          'ids': torch.arange(k, dtype=torch.int64, device=dev).repeat(B),
          'offsets': (torch.arange(B + 1, dtype=torch.int32, device=dev) * k),
      }
    self.tags = {}
    for name, t in schema.tags.items():
      self.tags[name] = {
          'ids': torch.full((t['cap'],), -1, dtype=torch.int64, device=dev),
          'offsets': torch.zeros(B + 1, dtype=torch.int32, device=dev),
          'weights': torch.zeros(t['cap'], dtype=torch.float32, device=dev) if t['weighted'] else None,
      }
    self.seqs = {}
    for name, s in schema.seqs.items():
      self.seqs[name] = {
          'ids': torch.full((B, s['max_len']), -1, dtype=torch.int64, device=dev),
          'len': torch.zeros(B, dtype=torch.int32, device=dev),
      }
    # longest sequence of the loaded batch, per sequence feature: the reference pads a batch's sequences to ITS longest
    # one (the sparse tensor's dense shape), and BatchNorm inside an attention MLP sees exactly those positions
    # (SURVEY.md App. B.2); the buffers here have the static max_seq_len, consumers slice [:, :seq_pad_len(name)]
    self.seq_batch_max = {name: s['max_len'] for name, s in schema.seqs.items()}
    self.pad_to_batch_max = True
    self._use_device_hash = False

  @property
  def batch_size(self):
    return self.schema.batch_size

  # -- dict-like views (the reference's parsed feature dict)
  def raw(self, name):
    if name in self.raw_multi:
      return self.raw_multi[name]['values']  # [B, k]
    if name not in self.schema.raw:
      # (ExprFeature columns: the reference evaluates the expression in the input pipeline, input.py:507-530)
      raise NotImplementedError('feature %s: ExprFeature / PassThroughFeature values are outside the hot-path scope '
                                '(no numeric input of that name in the batch)' % name)
    r = self.schema.raw[name]
    blk = self.raw_block[r['row']:r['row'] + r['dim']]
    return blk[0] if r['dim'] == 1 else blk  # [B] or [dim, B]

  def ids_of(self, name):
    if name in self.schema.hash_single:
      return self.hash_ids[self.schema.hash_single[name]['col']]
    if name in self.schema.int_single:
      return self.int_ids[self.schema.int_single[name]['col']]
    raise KeyError(name)

  def label(self, name):
    return self.labels[self.schema.label_fields.index(name)]

  def seq_pad_len(self, name):
    """Time steps of sequence feature `name` the model sees this step: the loaded batch's longest sequence (at least
    1), or the static max_seq_len when pad_to_batch_max is off."""
    L = self.schema.seqs[name]['max_len']
    return max(1, min(L, self.seq_batch_max.get(name, L))) if self.pad_to_batch_max else L

  def shape_signature(self):
    """What a captured hipGraph depends on besides buffer addresses: the padded lengths of the sequence features."""
    return tuple(self.seq_pad_len(n) for n in self.schema.seqs)

  def __contains__(self, name):
    s = self.schema
    return (name in s.raw or name in s.raw_multi or name in s.hash_single or name in s.int_single or name in s.tags or
            name in s.seqs)

  # -- batch loading
  def load(self, batch, non_blocking=True):
    """Copy one batch (dict of numpy arrays or device tensors; see input/input.py) into the buffers.

    Device-side transforms that belong to the training step (string hashing) are NOT run here;
    call `transform()` inside the (possibly graph-captured) step.
    """

    # device-resident parts (a batch ring in HBM) are gathered and copied by ONE launch at the end (er_copy_multi)
    dev_pairs = []
    multi = self.device.type == 'cuda'

    def put(dst, src):
      if src is None:
        return
      if isinstance(src, np.ndarray):
        src = torch.from_numpy(src)
      # every path below copies src.numel() elements into the FRONT of dst: a source larger than the buffer must fail
      # here (the one-launch device copy moves bytes and would overrun the buffer silently)
      if src.numel() > dst.numel():
        raise ValueError('batch array of %d elements exceeds the device buffer of %d' % (src.numel(), dst.numel()))
      if multi and src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous():
        if src.numel():
          dev_pairs.append((dst.view(-1)[:src.numel()], src.reshape(-1)))
        return
      if src.device != dst.device and src.device.type == 'cpu' and non_blocking and not src.is_pinned():
        src = src.pin_memory() if torch.cuda.is_available() else src
      dst.view(-1)[:src.numel()].copy_(src.reshape(-1), non_blocking=non_blocking)

    if 'packed' in batch:  # pack(): one copy for labels, raw values, ids and strings
      src = batch['packed']
      put(self.arena, src)
      if batch.get('packed_slot') is not None and self.device.type == 'cuda':  # pack()'s ring: the slot is free once copied
        ev = torch.cuda.Event()
        ev.record()
        self._pin_ring['events'][batch['packed_slot']] = ev
      self._use_device_hash = bool(batch['packed_has_strings'])
    put(self.labels, batch.get('labels'))
    put(self.raw_block, batch.get('raw'))
    if 'str_bytes' in batch:
      nb = batch['str_bytes']
      n = nb.numel() if torch.is_tensor(nb) else nb.size
      assert n <= self.str_bytes.numel(), 'string block of %d bytes exceeds capacity %d' % (
          n, self.str_bytes.numel())
      put(self.str_bytes, nb)
      put(self.str_offsets, batch['str_offsets'])
      self._use_device_hash = True
    elif 'hash_ids' in batch:
      put(self.hash_ids, batch['hash_ids'])
      self._use_device_hash = False
    put(self.int_ids, batch.get('int_ids'))
    for name, bufs in self.raw_multi.items():
      put(bufs['values'], batch.get('rawm/%s' % name))
    if 'sample_weight' in batch:
      if self.sample_weight is None:
        self.sample_weight = torch.ones(self.batch_size, dtype=torch.float32, device=self.device)
      put(self.sample_weight, batch['sample_weight'])
    for name, bufs in self.tags.items():
      ids = batch.get('tag/%s/ids' % name)
      if ids is None:
        continue
      n = ids.numel() if torch.is_tensor(ids) else ids.size
      assert n <= bufs['ids'].numel(), 'tag feature %s: %d ids exceed capacity %d' % (
          name, n, bufs['ids'].numel())
      put(bufs['ids'], ids)
      put(bufs['offsets'], batch['tag/%s/offsets' % name])
      if bufs['weights'] is not None:
        put(bufs['weights'], batch['tag/%s/weights' % name])
    for name, bufs in self.seqs.items():
      ids = batch.get('seq/%s/ids' % name)
      if ids is None:
        continue
      put(bufs['ids'], ids)
      lens = batch['seq/%s/len' % name]
      put(bufs['len'], lens)
      mx = batch.get('seq/%s/max' % name)  # (pack() leaves it: asking a device tensor for it would synchronise)
      self.seq_batch_max[name] = int(mx) if mx is not None else (int(lens.max()) if len(lens) else 0)
    if dev_pairs:
      from easyrec_amd import kernels
      if len(dev_pairs) == 1:
        dev_pairs[0][0].copy_(dev_pairs[0][1], non_blocking=non_blocking)
      else:
        (self._backend or kernels.hip()).copy_multi(dev_pairs)
    self.version += 1

  _PACKED_KEYS = {'labels': 'labels', 'raw': 'raw_block', 'int_ids': 'int_ids', 'hash_ids': 'hash_ids',
                  'str_offsets': 'str_offsets', 'str_bytes': 'str_bytes'}

  def pack(self, batch, device=None):
    """Batch dict (input/input.py) -> the same batch with its fixed-size parts laid out as ONE byte image of the
    input arena ('packed'): what a native loader would hand over, and a single host-to-device / device-to-device copy
    per step instead of one per array.  Ragged parts (tags, sequences, multi-valued raws) stay separate entries."""
    has_str = 'str_bytes' in batch
    last = 'str_bytes' if has_str else 'hash_ids'
    end = self._layout[last][0] + self._layout[last][1]
    used = end  # bytes of the image that carry the batch (the string section is cut at its last byte)
    if has_str:
      sb = batch['str_bytes']
      used = self._layout['str_bytes'][0] + int(sb.numel() if torch.is_tensor(sb) else sb.size)
    slot = None
    if device is None and self.device.type == 'cuda':
      # a ring of page-locked images: load() copies straight from them (no pin_memory() allocation + copy per step on the
      # consumer's side: 2.5 - 13 ms of host time per train_step, profiles/r06_s9_from_file_stages.txt) and records an
      # event; a slot is refilled only after its copy has left
      ring = self.__dict__.setdefault('_pin_ring', {'bufs': [], 'events': [], 'next': 0})
      cap = self._layout['str_bytes'][0] + self._layout['str_bytes'][1]
      if not ring['bufs']:
        ring['bufs'] = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(8)]
        ring['events'] = [None] * 8
      slot = ring['next']
      ring['next'] = (slot + 1) % len(ring['bufs'])
      if ring['events'][slot] is not None:
        ring['events'][slot].synchronize()
      img = ring['bufs'][slot].numpy()
      img[:self._layout['str_offsets'][0]] = 0
    else:
      img = np.zeros(end, dtype=np.uint8)
    out = {}
    for key, val in batch.items():
      name = self._PACKED_KEYS.get(key)
      if name is None:
        out[key] = val
        continue
      o, nbytes, shape, dt = self._layout[name]
      a = np.ascontiguousarray(val.cpu().numpy() if torch.is_tensor(val) else val)
      a = a.astype({torch.float32: np.float32, torch.int64: np.int64, torch.uint8: np.uint8}[dt], copy=False).reshape(-1)
      assert a.nbytes <= nbytes, '%s: %d bytes exceed the capacity %d' % (key, a.nbytes, nbytes)
      img[o:o + a.nbytes] = a.view(np.uint8)
    for name in self.schema.seqs:  # the longest sequence of the batch, while the lengths are still on the host (load())
      lens = batch.get('seq/%s/len' % name)
      if lens is not None and 'seq/%s/max' % name not in batch:
        out['seq/%s/max' % name] = int(lens.max()) if len(lens) else 0
    if not has_str and 'hash_ids' not in batch:
      o, nbytes, _, _ = self._layout['hash_ids']
      img[o:o + nbytes] = 0xFF  # -1: missing
    if slot is not None:
      out['packed'] = self._pin_ring['bufs'][slot][:used]
      out['packed_slot'] = slot
    else:
      t = torch.from_numpy(img[:used])
      out['packed'] = t.to(device) if device is not None else t
    out['packed_has_strings'] = has_str
    return out

  def hash_job(self):
    """hash_bucket_fast's arguments for the loaded batch's id strings, or None when the ids arrived hashed: the step
    prologue runs the hash as part of its own launch (kernels.HipBackend.step_prologue) and transform(done=True) follows."""
    if self._use_device_hash and self.hash_buckets is not None:
      return (self.str_bytes, self.str_offsets, self.batch_size, self.hash_buckets, True, self.hash_ids)
    return None

  def transform(self, hashed=False):
    """Device-side part of `_preprocess`: hash the packed id strings (K1).  hashed: the step prologue already did."""
    if hashed:
      return
    if self._use_device_hash and self.hash_buckets is not None:
      from easyrec_amd import kernels
      be = self._backend or kernels.hip()
      be.hash_bucket_fast(self.str_bytes, self.str_offsets, self.batch_size, self.hash_buckets, True,
                          out=self.hash_ids)

"""TensorFlow tensor-bundle files (`<prefix>.index` + `<prefix>.data-00000-of-00001`) for the dense variables.

The reference saves and restores everything that is not a sharded embedding through `tf.train.Saver`
(easy_rec/python/model/easy_rec_estimator.py:240-300, model/easy_rec_model.py:219-351 `restore` reads them back with
`tf.train.NewCheckpointReader` and matches variable names and shapes): kernels, biases, BatchNorm statistics, the
optimizer's slot variables (`<var>/Adam`, `<var>/Adam_1`, `<var>/Adagrad`) and `global_step`.  A TF Saver (V2) writes a
*tensor bundle*; this module writes and reads that format so that a checkpoint of this package names and stores its dense
variables the way the reference's own checkpoints do.

Format (restated from TensorFlow's published sources - tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc},
tensorflow/core/protobuf/tensor_bundle.proto, tensorflow/core/lib/io/{table_builder,block_builder,format}.cc, which are
LevelDB's table format; TensorFlow is not installable here, so the files are NOT checked against a TF install - the tests
hold the writer to the reader, to the format's invariants (footer magic, block checksums, sorted keys) and to known-answer
CRC-32C values):
  * data file: the tensors' little-endian bytes back to back, in key order;
  * index file: an sstable key -> serialized proto.  Key "" -> BundleHeaderProto{num_shards = 1, endianness = LITTLE,
    version{producer = 1}}; key <variable name> -> BundleEntryProto{dtype, shape, shard_id = 0, offset, size, crc32c}
    (crc32c = masked CRC-32C of the tensor's bytes);
  * sstable: data blocks of prefix-compressed entries (varint shared / non_shared / value_len, restart points every 16
    entries, then the restart array and its length), each followed by a 1-byte compression type (0) and the masked CRC-32C
    of block + type; an empty metaindex block; an index block (last key of each data block -> BlockHandle); a 48-byte
    footer (metaindex handle, index handle, padding, magic 0xdb4775248b80fb57).
"""
import ctypes
import os
import struct
from collections import OrderedDict

import numpy as np

from easyrec_amd import kernels

_MAGIC = 0xdb4775248b80fb57
_BLOCK_SIZE = 256 * 1024   # table::Options default block_size
_RESTART_INTERVAL = 16
# tensorflow/core/framework/types.proto
_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9,
       np.dtype(np.bool_): 10, np.dtype(np.uint8): 4}
_DT_INV = {v: k for k, v in _DT.items()}

_lib = None


def _crc32c(data):
  global _lib
  if _lib is None:
    _lib = ctypes.CDLL(kernels.LIB_PATH)
    _lib.er_crc32c.restype = ctypes.c_uint32
    _lib.er_crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int64]
  return _lib.er_crc32c(0, data, len(data))


def _mask(crc):
  """crc32c::Mask: rotate right by 15 bits, add a constant (stored CRCs of data that itself contains CRCs)."""
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _unmask(masked):
  rot = (masked - 0xa282ead8) & 0xFFFFFFFF
  return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def _varint(n):
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _read_varint(buf, pos):
  shift = result = 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7F) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


# -- the three protos, by hand (proto3 wire format; default-valued fields are omitted as protobuf omits them)
def _field(num, wire, payload):
  return _varint((num << 3) | wire) + payload


def _header_proto():
  version = _field(1, 0, _varint(1))                       # VersionDef.producer = 1 (kTensorBundleVersion)
  return _field(1, 0, _varint(1)) + _field(3, 2, _varint(len(version)) + version)  # num_shards = 1; endianness LITTLE = 0


def _shape_proto(shape):
  out = b''
  for d in shape:
    dim = _field(1, 0, _varint(int(d))) if int(d) != 0 else b''
    out += _field(2, 2, _varint(len(dim)) + dim)
  return out


def _entry_proto(dtype, shape, offset, size, crc):
  out = _field(1, 0, _varint(dtype))
  sp = _shape_proto(shape)
  out += _field(2, 2, _varint(len(sp)) + sp)
  if offset:
    out += _field(4, 0, _varint(offset))
  if size:
    out += _field(5, 0, _varint(size))
  out += _field(6, 5, struct.pack('<I', crc))
  return out


def _parse_fields(buf):
  pos, fields = 0, []
  while pos < len(buf):
    tag, pos = _read_varint(buf, pos)
    num, wire = tag >> 3, tag & 7
    if wire == 0:
      val, pos = _read_varint(buf, pos)
    elif wire == 2:
      n, pos = _read_varint(buf, pos)
      val = bytes(buf[pos:pos + n])
      pos += n
    elif wire == 5:
      val = struct.unpack('<I', bytes(buf[pos:pos + 4]))[0]
      pos += 4
    elif wire == 1:
      val = struct.unpack('<Q', bytes(buf[pos:pos + 8]))[0]
      pos += 8
    else:
      raise ValueError('unsupported wire type %d' % wire)
    fields.append((num, val))
  return fields


# -- sstable
class _BlockBuilder(object):

  def __init__(self):
    self.buf = bytearray()
    self.restarts = [0]
    self.counter = 0
    self.last_key = b''

  def add(self, key, value):
    shared = 0
    if self.counter < _RESTART_INTERVAL:
      n = min(len(self.last_key), len(key))
      while shared < n and self.last_key[shared] == key[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.counter = 0
    self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
    self.last_key = key
    self.counter += 1

  def size(self):
    return len(self.buf) + 4 * (len(self.restarts) + 1)

  def finish(self):
    return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _write_block(f, contents):
  """-> BlockHandle (offset, size); the 5-byte trailer (type 0 = no compression, masked crc of contents + type) follows."""
  offset = f.tell()
  f.write(contents)
  trailer_type = b'\x00'
  f.write(trailer_type + struct.pack('<I', _mask(_crc32c(contents + trailer_type))))
  return offset, len(contents)


def _write_table(path, items):
  """items: [(key bytes, value bytes)] in ascending key order."""
  with open(path, 'wb') as f:
    index = _BlockBuilder()
    block = _BlockBuilder()
    prev = None
    for key, value in items:
      assert prev is None or key > prev, 'keys must be sorted and unique'
      prev = key
      block.add(key, value)
      if block.size() >= _BLOCK_SIZE:
        handle = _write_block(f, block.finish())
        index.add(block.last_key, _varint(handle[0]) + _varint(handle[1]))
        block = _BlockBuilder()
    if block.counter > 0 or len(block.buf) > 0:
      handle = _write_block(f, block.finish())
      index.add(block.last_key, _varint(handle[0]) + _varint(handle[1]))
    meta = _write_block(f, _BlockBuilder().finish())
    idx = _write_block(f, index.finish())
    footer = _varint(meta[0]) + _varint(meta[1]) + _varint(idx[0]) + _varint(idx[1])
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    f.write(footer)


def _read_block(buf, offset, size):
  contents = bytes(buf[offset:offset + size])
  ctype = buf[offset + size]
  stored = struct.unpack('<I', bytes(buf[offset + size + 1:offset + size + 5]))[0]
  if ctype != 0:
    raise ValueError('compressed table blocks are not supported')
  if _unmask(stored) != _crc32c(contents + bytes([ctype])):
    raise ValueError('table block checksum mismatch at offset %d' % offset)
  n_restarts = struct.unpack('<I', contents[-4:])[0]
  end = len(contents) - 4 * (n_restarts + 1)
  pos, key, out = 0, b'', []
  while pos < end:
    shared, pos = _read_varint(contents, pos)
    non_shared, pos = _read_varint(contents, pos)
    vlen, pos = _read_varint(contents, pos)
    key = key[:shared] + contents[pos:pos + non_shared]
    pos += non_shared
    out.append((key, contents[pos:pos + vlen]))
    pos += vlen
  return out


def _read_table(path):
  buf = open(path, 'rb').read()
  if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != _MAGIC:
    raise ValueError('%s is not an sstable (bad magic)' % path)
  footer = buf[-48:]
  pos = 0
  _, pos = _read_varint(footer, pos)
  _, pos = _read_varint(footer, pos)
  ioff, pos = _read_varint(footer, pos)
  isize, pos = _read_varint(footer, pos)
  items = []
  for _, handle in _read_block(buf, ioff, isize):
    off, p = _read_varint(handle, 0)
    size, _ = _read_varint(handle, p)
    items.extend(_read_block(buf, off, size))
  return items


# -- the bundle
def data_file(prefix):
  return prefix + '.data-00000-of-00001'


def write_bundle(prefix, tensors):
  """tensors: {variable name: numpy array}.  Writes <prefix>.index and <prefix>.data-00000-of-00001."""
  names = sorted(tensors, key=lambda s: s.encode('utf-8'))
  items = [(b'', _header_proto())]
  offset = 0
  # (both files are written under temporary names, flushed to disk and renamed when complete - the data file first, the
  # index last.  A crash before the renames leaves the previous bundle of this prefix untouched; a crash BETWEEN the two
  # renames while re-saving an existing prefix leaves new data under the old index, which read_bundle's per-tensor CRCs
  # reject - that bundle is then lost.  EmbeddingCheckpoint saves under a fresh step-numbered prefix and publishes it with
  # the atomically renamed `checkpoint` state file, so only a re-save of the same step is exposed to that window.)
  tmp_data, tmp_index = data_file(prefix) + '.tmp', prefix + '.index.tmp'
  with open(tmp_data, 'wb') as f:
    for name in names:
      a = np.asarray(tensors[name])
      a = a if a.ndim == 0 else np.ascontiguousarray(a)  # (ascontiguousarray would turn a scalar into shape [1])
      if a.dtype not in _DT:
        raise TypeError('%s: dtype %s has no tensor-bundle encoding here' % (name, a.dtype))
      raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
      f.write(raw)
      items.append((name.encode('utf-8'), _entry_proto(_DT[a.dtype], a.shape, offset, len(raw), _mask(_crc32c(raw)))))
      offset += len(raw)
    f.flush()
    os.fsync(f.fileno())
  _write_table(tmp_index, items)
  with open(tmp_index, 'rb') as f:
    os.fsync(f.fileno())
  os.replace(tmp_data, data_file(prefix))
  os.replace(tmp_index, prefix + '.index')
  try:  # the renames themselves: the directory entry
    dfd = os.open(os.path.dirname(os.path.abspath(prefix)) or '.', os.O_RDONLY)
    try:
      os.fsync(dfd)
    finally:
      os.close(dfd)
  except OSError:
    pass


def read_bundle(prefix):
  """-> OrderedDict {variable name: numpy array}; checks the table's block checksums and every tensor's CRC."""
  items = _read_table(prefix + '.index')
  assert items and items[0][0] == b'', 'tensor bundle without a header entry'
  header = dict(_parse_fields(items[0][1]))
  if header.get(1, 0) != 1 or header.get(2, 0) != 0:
    raise ValueError('only single-shard little-endian bundles are supported')
  data = open(data_file(prefix), 'rb').read()
  out = OrderedDict()
  for key, value in items[1:]:
    f = _parse_fields(value)
    d = dict(f)
    shape = []
    for num, dim in _parse_fields(d.get(2, b'')):
      if num == 2:
        shape.append(dict(_parse_fields(dim)).get(1, 0))
    off, size = d.get(4, 0), d.get(5, 0)
    raw = data[off:off + size]
    if _unmask(d[6]) != _crc32c(raw):
      raise ValueError('tensor %s: checksum mismatch' % key.decode('utf-8'))
    out[key.decode('utf-8')] = np.frombuffer(raw, dtype=_DT_INV[d.get(1, 0)].newbyteorder('<')).reshape(shape).copy()
  return out


def write_checkpoint_state(ckpt_path):
  """The `checkpoint` text proto tf.train.latest_checkpoint reads (CheckpointState: model_checkpoint_path + the history
  all_model_checkpoint_paths, oldest first, which tf.train.Saver keeps and which its max_to_keep clean-up walks).  Call it
  AFTER the bundle files exist: written to a temporary file and renamed, so the state never names a checkpoint that is not
  complete.  Earlier entries whose files are gone are dropped, as the Saver's own recovery does."""
  folder, name = os.path.split(os.path.abspath(ckpt_path))
  path = os.path.join(folder, 'checkpoint')
  history = []
  if os.path.exists(path):
    import re
    for line in open(path):
      m = re.match(r'\s*all_model_checkpoint_paths:\s*"(.*)"\s*$', line)
      if m and m.group(1) != name and m.group(1) not in history:
        old = m.group(1) if os.path.isabs(m.group(1)) else os.path.join(folder, m.group(1))
        if os.path.exists(old + '.index'):
          history.append(m.group(1))
  history.append(name)
  with open(path + '.tmp', 'w') as f:
    f.write('model_checkpoint_path: "%s"\n' % name)
    for h in history:
      f.write('all_model_checkpoint_paths: "%s"\n' % h)
    f.flush()
    os.fsync(f.fileno())
  os.replace(path + '.tmp', path)

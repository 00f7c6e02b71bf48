"""Independent readers of the checkpoint formats, for the tests only (nothing here imports easyrec_amd.utils.checkpoint or
tensor_bundle): what a reference-side consumer would do with the files a run of this package writes.

  * ref_load_embed: the reference's python re-shard `_load_embed` (compat/embedding_parallel_saver.py:140-168) restated;
    tests/test_checkpoint_pins.py holds it to the outputs of the reference's own function (tests/golden/
    checkpoint_vectors.npz), so the -m gpu checkpoint test can read a GPU run's part files with it.
  * read_bundle: a TensorFlow tensor-bundle reader written from the published format - LevelDB table (footer, index block,
    data blocks with prefix-compressed keys and restart arrays, masked CRC-32C trailers), BundleHeaderProto /
    BundleEntryProto decoded by the REAL protobuf runtime from descriptors built here, CRC-32C in pure Python.
"""
import glob
import os
import struct

import numpy as np


def ref_load_embed(folder, var_name, embed_dim, embed_part_size, part_id, part_num):
  files = glob.glob(glob.escape(os.path.join(folder, var_name)) + '-part-*.bin')
  files.sort(key=lambda p: int(p.split('-')[-1].replace('.bin', '')))
  out = np.zeros([embed_part_size, embed_dim], dtype=np.float32)
  for f in files:
    part_id_o = int(f.split('-')[-1].replace('.bin', ''))
    val = np.frombuffer(open(f, 'rb').read(), np.float32).reshape([-1, embed_dim])
    ids_o = part_id_o + np.arange(len(val)) * len(files)
    sel = np.where(np.logical_and((ids_o % part_num) == part_id, ids_o < embed_part_size * part_num))[0]
    out[np.array(ids_o[sel] / part_num, dtype=np.int64)] = val[sel]
  return out


# ---- CRC-32C (Castagnoli), table driven, and TensorFlow's mask
_CRC_TABLE = []
for _i in range(256):
  _c = _i
  for _ in range(8):
    _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
  _CRC_TABLE.append(_c)


def crc32c(data):
  c = 0xFFFFFFFF
  for b in bytes(data):
    c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def crc32c_np(data):
  """the same over a large buffer: slicing-by-1 with numpy state would be slow in pure python; 64 KiB chunks of table
  lookups keep a few-MB data file under a second"""
  c = 0xFFFFFFFF
  tab = _CRC_TABLE
  for b in memoryview(bytes(data)):
    c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def unmask(m):
  rot = (m - 0xa282ead8) & 0xFFFFFFFF
  return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def _varint(buf, pos):
  shift = res = 0
  while True:
    b = buf[pos]
    pos += 1
    res |= (b & 0x7F) << shift
    if not b & 0x80:
      return res, pos
    shift += 7


def _protos():
  """BundleHeaderProto / BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto) + what they embed, as message classes
  of the real protobuf runtime"""
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto()
  fd.name = 'test_tensor_bundle.proto'
  fd.package = 'tbtest'
  fd.syntax = 'proto3'
  T = descriptor_pb2.FieldDescriptorProto

  def msg(name, fields):
    m = fd.message_type.add()
    m.name = name
    for fname, num, ftype, label, tname in fields:
      f = m.field.add()
      f.name, f.number, f.type, f.label = fname, num, ftype, label
      if tname:
        f.type_name = '.tbtest.' + tname
    return m

  O, R = T.LABEL_OPTIONAL, T.LABEL_REPEATED
  msg('VersionDef', [('producer', 1, T.TYPE_INT32, O, None), ('min_consumer', 2, T.TYPE_INT32, O, None)])
  msg('Dim', [('size', 1, T.TYPE_INT64, O, None), ('name', 2, T.TYPE_STRING, O, None)])
  msg('TensorShapeProto', [('dim', 2, T.TYPE_MESSAGE, R, 'Dim'), ('unknown_rank', 3, T.TYPE_BOOL, O, None)])
  msg('BundleHeaderProto', [('num_shards', 1, T.TYPE_INT32, O, None), ('endianness', 2, T.TYPE_INT32, O, None),
                            ('version', 3, T.TYPE_MESSAGE, O, 'VersionDef')])
  msg('BundleEntryProto', [('dtype', 1, T.TYPE_INT32, O, None), ('shape', 2, T.TYPE_MESSAGE, O, 'TensorShapeProto'),
                           ('shard_id', 3, T.TYPE_INT32, O, None), ('offset', 4, T.TYPE_INT64, O, None),
                           ('size', 5, T.TYPE_INT64, O, None), ('crc32c', 6, T.TYPE_FIXED32, O, None)])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = getattr(message_factory, 'GetMessageClass', None)
  if get is None:
    fac = message_factory.MessageFactory(pool)
    get = fac.GetPrototype
  return (get(pool.FindMessageTypeByName('tbtest.BundleHeaderProto')),
          get(pool.FindMessageTypeByName('tbtest.BundleEntryProto')))


def _block(buf, offset, size):
  """contents of the block at (offset, size): checks the 5-byte trailer (compression type 0 + masked CRC-32C of block + type)"""
  body = buf[offset:offset + size]
  ctype = buf[offset + size]
  stored = struct.unpack('<I', buf[offset + size + 1:offset + size + 5])[0]
  assert ctype == 0, 'compressed block'
  assert unmask(stored) == crc32c(body + bytes([ctype])), 'block checksum'
  return body


def _entries(block):
  """(key, value) pairs of a table block: entries up to the restart array (uint32 offsets + their count at the very end)"""
  n_restarts = struct.unpack('<I', block[-4:])[0]
  limit = len(block) - 4 - 4 * n_restarts
  restarts = struct.unpack('<%dI' % n_restarts, block[limit:len(block) - 4])
  assert restarts[0] == 0 and list(restarts) == sorted(restarts)
  pos, key, out = 0, b'', []
  while pos < limit:
    at = pos
    shared, pos = _varint(block, pos)
    non_shared, pos = _varint(block, pos)
    vlen, pos = _varint(block, pos)
    if at in restarts:
      assert shared == 0, 'a restart point stores its key whole'
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    out.append((key, block[pos:pos + vlen]))
    pos += vlen
  assert pos == limit
  return out


_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 9: np.int64, 10: np.bool_}


def read_bundle(prefix):
  """{variable name: array} of `<prefix>.index` + `<prefix>.data-00000-of-00001`, every checksum verified"""
  Header, Entry = _protos()
  idx = open(prefix + '.index', 'rb').read()
  assert len(idx) >= 48
  footer = idx[-48:]
  assert struct.unpack('<Q', footer[40:])[0] == 0xdb4775248b80fb57, 'table magic'
  pos = 0
  meta_off, pos = _varint(footer, pos)
  meta_size, pos = _varint(footer, pos)
  index_off, pos = _varint(footer, pos)
  index_size, pos = _varint(footer, pos)
  _block(idx, meta_off, meta_size)  # (metaindex: present, checksummed, unused)
  pairs = []
  for last_key, handle in _entries(_block(idx, index_off, index_size)):
    off, p = _varint(handle, 0)
    size, p = _varint(handle, p)
    ents = _entries(_block(idx, off, size))
    assert ents and ents[-1][0] <= last_key
    pairs.extend(ents)
  keys = [k for k, _ in pairs]
  assert keys == sorted(keys) and len(set(keys)) == len(keys), 'keys must be sorted and unique'
  assert keys[0] == b'', 'the header entry has the empty key'
  hdr = Header()
  hdr.ParseFromString(pairs[0][1])
  assert hdr.num_shards == 1 and hdr.endianness == 0 and hdr.version.producer == 1
  data = open(prefix + '.data-00000-of-00001', 'rb').read()
  out, end = {}, 0
  for k, v in pairs[1:]:
    e = Entry()
    e.ParseFromString(v)
    assert e.shard_id == 0
    raw = data[e.offset:e.offset + e.size]
    assert len(raw) == e.size and unmask(e.crc32c) == crc32c_np(raw), k
    shape = [d.size for d in e.shape.dim]
    arr = np.frombuffer(raw, dtype=_DTYPES[e.dtype])
    assert arr.size == int(np.prod(shape, dtype=np.int64)), k
    out[k.decode()] = arr.reshape(shape)
    end = max(end, e.offset + e.size)
  assert end == len(data), 'the data file holds the tensors back to back and nothing else'
  return out

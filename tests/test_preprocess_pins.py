"""easyrec_amd/input/input.py against the REFERENCE'S OWN Input._parse_* (tests/golden/make_preprocess_vectors.py, run
where /root/reference exists, on a stand-in for the dozen TensorFlow string / sparse ops they call): hashed and
num_buckets IdFeatures from string / integer columns, RawFeatures (string and numeric columns, min / max normalisation, a
multi-dimensional one with missing trailing values), a bucketized one, TagFeatures (a separator SET, kv weights, a second
weight column, integer ids) and SequenceFeatures (hashed, integer, bucketized numbers) - empty cells, doubled
separators and non-ASCII text included.  Hashed columns are compared through the oracle's pinned Fingerprint64."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import preprocess_cases as pc  # noqa: E402

with open(os.path.join(HERE, 'golden', 'preprocess_vectors.json')) as f:
  REF = json.load(f)['parsed']


def _hash(strings, buckets):
  """Fingerprint64 % buckets of each string ('' -> -1: dropped), by the oracle's pinned restatement"""
  from oracle import hashing
  data = b''.join(s.encode('utf-8') for s in strings)
  offsets = np.zeros(len(strings) + 1, dtype=np.int64)
  offsets[1:] = np.cumsum([len(s.encode('utf-8')) for s in strings])
  return hashing.hash_bucket_fast(np.frombuffer(data, dtype=np.uint8), offsets, len(strings), np.array([buckets], dtype=np.int64),
                                  True).reshape(-1)


def _batch():
  from google.protobuf import text_format

  from easyrec_amd.input.input import Input
  from easyrec_amd.protos import dataset_pb2, feature_config_pb2
  dc = dataset_pb2.DatasetConfig()
  text_format.Merge(pc.DATA_CONFIG, dc)
  fcs = feature_config_pb2.FeatureConfigV2()
  text_format.Merge(pc.FEATURES, fcs)
  inp = Input(dc, list(fcs.features), batch_size=6, hash_on_host=True)
  cols = {}
  kinds = {f.input_name: f.input_type for f in dc.input_fields}
  for name, col in pc.COLUMNS.items():
    t = kinds[name]
    cols[name] = list(col) if t == dataset_pb2.DatasetConfig.STRING else np.asarray(
        col, dtype={dataset_pb2.DatasetConfig.INT32: np.int32, dataset_pb2.DatasetConfig.INT64: np.int64,
                    dataset_pb2.DatasetConfig.FLOAT: np.float32, dataset_pb2.DatasetConfig.DOUBLE: np.float64}[t])
  return inp, inp.preprocess(cols), {(f.feature_name or f.input_names[0]): f for f in fcs.features}


def _ragged(batch, kind, name):
  if kind == 'tag':
    ids, offs = np.asarray(batch['tag/%s/ids' % name]), np.asarray(batch['tag/%s/offsets' % name])
    w = np.asarray(batch['tag/%s/weights' % name]) if 'tag/%s/weights' % name in batch else None
    rows = [ids[offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]
    return rows, None if w is None else [w[offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]
  ids, lens = np.asarray(batch['seq/%s/ids' % name]), np.asarray(batch['seq/%s/len' % name])
  return [ids[i, :lens[i]].tolist() for i in range(len(lens))], None


def test_single_valued_features(ref_backend):
  inp, batch, fcs = _batch()
  sch = inp.schema
  for name in ('uid', 'item'):  # hashed: the reference hands these STRINGS to the hashed column
    want = _hash(REF[name]['strings'], fcs[name].hash_bucket_size)
    assert np.array_equal(np.asarray(batch['hash_ids'])[sch.hash_single[name]['col']], want), name
  for name in ('city', 'level'):  # num_buckets: integers (strings converted by string_to_number)
    assert np.asarray(batch['int_ids'])[sch.int_single[name]['col']].tolist() == REF[name]['values'], name
  for name in ('price', 'ctr', 'age'):
    got = np.asarray(batch['raw'])[sch.raw[name]['row']]
    assert np.allclose(got, np.asarray(REF[name]['values'], dtype=np.float32), rtol=1e-6, atol=0), name
  assert np.allclose(np.asarray(batch['rawm/vec']), np.asarray(REF['vec']['values'], dtype=np.float32).reshape(6, 3))
  # the bucketized raw feature: its bucket index from the (unnormalised) value and the sorted boundaries
  from easyrec_amd.input.features import bucketize
  assert np.asarray(batch['int_ids'])[sch.int_single['age']['col']].tolist() == \
      bucketize(np.asarray(REF['age']['values'], dtype=np.float32), np.asarray([18, 30, 45], dtype=np.float32)).tolist()


def test_tag_features(ref_backend):
  inp, batch, fcs = _batch()
  rows, _ = _ragged(batch, 'tag', 'tags')
  assert rows == [_hash(r, 100).tolist() if r else [] for r in REF['tags']['sparse_rows']]
  rows, w = _ragged(batch, 'tag', 'tags_kv')
  assert rows == [_hash(r, 100).tolist() for r in REF['tags_kv']['sparse_rows']]
  assert np.allclose(np.concatenate(w), np.concatenate(REF['tags_kv_w']['sparse_rows']))
  rows, w = _ragged(batch, 'tag', 'tag_ids')
  assert rows == REF['tag_ids']['sparse_rows']
  assert np.allclose(np.concatenate(w), np.concatenate(REF['tag_ids_w']['sparse_rows']))


def test_sequence_features(ref_backend):
  inp, batch, fcs = _batch()
  rows, _ = _ragged(batch, 'seq', 'clicks')
  # tf.strings.split keeps empty tokens ('c1||c2' -> 3 positions, an empty cell -> 1) and the hashed column takes the
  # sparse result as it is: the empty string is hashed like any token (only DENSE inputs have '' dropped)
  from oracle import hashing
  empty_id = int(hashing.fingerprint64(b'') % 200)
  want = [[(int(_hash([t], 200)[0]) if t != '' else empty_id) for t in r] for r in REF['clicks']['sparse_rows']]
  assert REF['clicks']['sparse_rows'][3] == [''] and rows == want
  assert _ragged(batch, 'seq', 'cates')[0] == REF['cates']['sparse_rows']
  from easyrec_amd.input.features import bucketize
  got = _ragged(batch, 'seq', 'prices')[0]
  assert got == [bucketize(np.asarray(r, dtype=np.float32), np.asarray([1, 5, 10], dtype=np.float32)).tolist()
                 for r in REF['prices']['sparse_rows']]


def test_lookup_and_combo_features(ref_backend):
  from oracle import hashing
  inp, batch, fcs = _batch()
  sch = inp.schema
  # LookupFeature: the values of the map's pairs whose key is the row's key, in map order, hashed
  rows, _ = _ragged(batch, 'tag', 'city_value')
  assert rows == [_hash(r, 300).tolist() if r else [] for r in REF['city_value']['sparse_rows']]
  # ComboFeature with combo_join_sep: ONE hashed string per row, the inputs joined
  assert np.array_equal(np.asarray(batch['hash_ids'])[sch.hash_single['city_level']['col']],
                        _hash(REF['city_level']['strings'], 400))
  # ComboFeature without it: crossed_column over the inputs as strings (integers through as_string) - the input stage
  # computes the crossed id; the strings the reference hands to the cross are what was crossed
  a, b = REF['uid_x_level']['strings'], REF['uid_x_level_1']['strings']
  want = [int(hashing.sparse_cross_hashed([x.encode('utf-8'), y.encode('utf-8')], 400)) for x, y in zip(a, b)]
  assert np.asarray(batch['int_ids'])[sch.int_single['uid_x_level']['col']].tolist() == want

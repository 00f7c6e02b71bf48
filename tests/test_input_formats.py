"""Packed input formats against vectors produced by the reference's own loader code
(tests/golden/make_input_vectors.py ran easy_rec/python/input/load_parquet.py and criteo_binary_reader.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import make_input_vectors as mv  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402

GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'input_vectors.npz'))


def _gold(prefix):
  n = 0
  while '%s%d/label' % (prefix, n) in GOLD:
    n += 1
  return n


def _parquet_reader(tmp_path):
  pytest.importorskip('pyarrow')
  from easyrec_amd.input.input import Input
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_parquet_small.config'))
  paths = mv.write_parquet_files(str(tmp_path))
  reader = Input.create(cfg.data_config, cfg.feature_config.features, ','.join(paths))
  assert type(reader).__name__ == 'ParquetInput'
  return cfg, reader


@pytest.mark.parametrize('drop', [True, False])
def test_parquet_packed_batches_equal_the_reference_loader(tmp_path, drop):
  """lens / vals / dense_fea / label of every batch, remainders carried across the three files."""
  cfg, reader = _parquet_reader(tmp_path)
  got = list(reader.reference_batches(drop_remainder=drop))
  pre = 'parquet/drop%d/' % int(drop)
  assert len(got) == _gold(pre) == (6 if drop else 7)
  for i, d in enumerate(got):
    lens, vals = d['sparse_fea']
    assert lens.dtype == np.int32 and np.array_equal(lens, GOLD['%s%d/lens' % (pre, i)])
    assert np.array_equal(vals, GOLD['%s%d/vals' % (pre, i)])
    assert d['dense_fea'].dtype == np.float32 and np.array_equal(d['dense_fea'], GOLD['%s%d/dense' % (pre, i)])
    assert np.array_equal(d['label'], GOLD['%s%d/label' % (pre, i)])


def test_parquet_batches_feed_the_device_buffers(tmp_path):
  """The engine-format batch of the same rows: ids % num_buckets (parquet_input.py:222), ragged tag offsets, raw
  values un-normalised (the reference bypasses Input._preprocess for this input), and it loads into DeviceFeatures."""
  from easyrec_amd.input.features import DeviceFeatures
  cfg, reader = _parquet_reader(tmp_path)
  B = cfg.data_config.batch_size
  feats = DeviceFeatures(reader.schema, 'cpu')
  sch = reader.schema
  for i, b in enumerate(reader.batches()):
    pre = 'parquet/drop1/%d/' % i
    lens, vals = GOLD[pre + 'lens'], GOLD[pre + 'vals'] % 1000
    assert np.array_equal(b['int_ids'][sch.int_single['s1']['col']], vals[:B])
    assert np.array_equal(b['int_ids'][sch.int_single['s2']['col']], vals[B:2 * B])
    assert np.array_equal(b['tag/t1/ids'], vals[2 * B:])
    assert np.array_equal(np.diff(b['tag/t1/offsets']), lens[2 * B:])
    assert np.array_equal(b['raw'][sch.raw['d1']['row']], GOLD[pre + 'dense'][:, 0])
    assert np.array_equal(b['raw'][sch.raw['d2']['row']], GOLD[pre + 'dense'][:, 1])
    assert np.array_equal(b['labels'][0], GOLD[pre + 'label'].astype(np.float32))
    feats.load(feats.pack(b))
    assert np.array_equal(feats.int_ids[sch.int_single['s2']['col']].numpy(), vals[B:2 * B])
    n = int(lens[2 * B:].sum())
    assert np.array_equal(feats.tags['t1']['ids'][:n].numpy(), vals[2 * B:])
  assert i == 5


@pytest.mark.parametrize('rank,size', [(0, 1), (0, 2), (1, 2)])
@pytest.mark.parametrize('drop', [True, False])
def test_criteo_binary_batches_equal_the_reference_reader(tmp_path, rank, size, drop):
  from easyrec_amd.input.criteo_input import BinaryDataset
  files = mv.write_criteo_files(str(tmp_path))
  ds = BinaryDataset(*files, batch_size=mv.BATCH, drop_last=drop, global_rank=rank, global_size=size)
  pre = 'criteo/r%d_of_%d/drop%d/' % (rank, size, int(drop))
  assert len(ds) == _gold(pre)
  n_straddle = 0
  for i in range(len(ds)):
    dense, cat, lbl = ds[i]
    assert dense.dtype == np.float32 and cat.dtype == np.uint32 and lbl.dtype == np.int32
    gd, gc, gl = (GOLD['%s%d/%s' % (pre, i, k)] for k in ('dense', 'category', 'label'))
    if len(gd) > mv.BATCH:
      # a batch that straddles two parts: the reference returns too many rows (see the deviation note in
      # easyrec_amd/input/criteo_input.py); ours is exactly the first batch_size of them
      n_straddle += 1
      gd, gc, gl = gd[:mv.BATCH], gc[:mv.BATCH], gl[:mv.BATCH]
    assert np.array_equal(dense, gd)
    assert np.array_equal(cat, gc)
    assert np.array_equal(lbl, gl)
  assert n_straddle >= 1
  with pytest.raises(IndexError):
    ds[len(ds)]


def test_criteo_input_columns(tmp_path):
  """CriteoInput: f1..f13 / c1..c26 / label columns (criteo_input.py:75-85) through the ordinary preprocessing of
  the DeepFM-Criteo feature config: categorical ints are stringified and hashed like the CSV path's cells."""
  from easyrec_amd.input.criteo_input import CriteoInput
  from easyrec_amd.input.csv_input import CSVInput
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
  names = [x.input_name for x in cfg.data_config.input_fields]
  lbl, dense, cat = mv.write_criteo_files(str(tmp_path))
  B = 16
  reader = CriteoInput(cfg.data_config, cfg.feature_config.features, {'label_path': lbl[:1], 'dense_path': dense[:1],
                                                                      'category_path': cat[:1]}, batch_size=B,
                       hash_on_host=False)
  batches = list(reader.batches(num_epochs=1))
  assert len(batches) == 40 // B
  # the same rows written as the TSV the CSV reader parses give the same batch
  d = np.fromfile(dense[0], dtype=np.float32).reshape(-1, 13)
  c = np.fromfile(cat[0], dtype=np.uint32).reshape(-1, 26)
  y = np.fromfile(lbl[0], dtype=np.int32)
  assert names[0] == 'label' and names[1].lower() == 'f1' and names[14].lower() == 'c1', names[:15]
  path = os.path.join(str(tmp_path), 'rows.tsv')
  with open(path, 'w') as f:
    for r in range(2 * B):
      f.write('\t'.join([str(int(y[r]))] + [repr(float(v)) for v in d[r]] + [str(int(v)) for v in c[r]]) + '\n')
  csv = CSVInput(cfg.data_config, cfg.feature_config.features, path, batch_size=B)
  for a, b in zip(batches, csv.batches(num_epochs=1)):
    assert set(a) == set(b)
    for k in a:
      assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_model_trains_from_parquet_batches(tmp_path, ref_backend):
  """End to end on the host path: ParquetInput batches drive two optimisation steps of the DeepFM built from the same
  config, and the losses equal the model-level oracle's on the same batches."""
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg, reader = _parquet_reader(tmp_path)
  B = cfg.data_config.batch_size
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=2).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  n = 0
  for b in reader.batches():
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(got[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, got[k], exp[k])
    n += 1
    if n == 2:
      break
  assert n == 2


def test_reference_ep_scenario_parquet_shared_table_lazy_adam(tmp_path, ref_backend):
  """The shape of the reference's own embedding-parallel config (samples/model_config/
  dlrm_on_criteo_parquet_ep_v2.config): ParquetInput, ONE table behind every id feature (`embedding_name`),
  `lazy_adam_optimizer`, EmbeddingParallelStrategy - here on the small parquet fixture: the embedding-parallel
  estimator (world 1, fixed-capacity exchange after the device-wide sort) must follow the plain one exactly."""
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.model.embedding_parallel import EmbeddingParallelEstimator
  cfg, reader = _parquet_reader(tmp_path)
  for fc in cfg.feature_config.features:
    if fc.num_buckets > 0:
      fc.embedding_name = 'embedding'
  oc = cfg.train_config.optimizer_config[0]
  if oc.WhichOneof('optimizer') == 'adam_optimizer':
    oc.lazy_adam_optimizer.learning_rate.CopyFrom(oc.adam_optimizer.learning_rate)
  B = cfg.data_config.batch_size
  batches = []
  for b in reader.batches():
    batches.append(b)
    if len(batches) == 3:
      break
  ref = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=2).build()
  est = EmbeddingParallelEstimator(cfg, device='cpu', batch_size=B, seed=2, rank=0, world=1, replicate_bytes=64).build()
  assert est.engine.padded and any(p[0] == 'shard' for p in est.engine.placement.values())
  shared = [n for n in est.engine.tables if n.startswith('embedding')]
  assert shared and len(est.engine.tables) < len(list(cfg.feature_config.features)) + 2  # one table for the id features
  for b in batches:
    ref.train_step(b)
    est.train_step(b)
    a, c = ref.loss_values(), est.loss_values()
    for k in a:
      assert abs(a[k] - c[k]) <= 1e-6 * max(1.0, abs(a[k])), (k, a[k], c[k])
  sa, sc = ref.state_dict(slots=True), est.state_dict(slots=True)
  assert set(sa) == set(sc)
  for k in sa:
    assert np.allclose(sa[k], sc[k], rtol=1e-6, atol=1e-8), k


@pytest.mark.parametrize('drop', [True, False])
def test_native_csv_decode_equals_the_line_by_line_path(tmp_path, built_lib, drop, monkeypatch):
  """er_decode_csv_host (one pass per batch, string cells as views of the file bytes) against the Python
  line-by-line decode: CRLF and blank lines, empty cells -> defaults, negative / exponent numbers, a batch that
  straddles two files, the padded remainder."""
  import time
  from easyrec_amd.input.csv_input import CSVInput
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
  feats = list(cfg.feature_config.features)
  rng = np.random.default_rng(8)
  B = 64

  def line(i):
    f = [rng.choice(['%d' % rng.integers(-5, 500), '%.3f' % rng.normal(), '1e2', '']) for _ in range(13)]
    c = ['%08x' % rng.integers(0, 2**32) if rng.random() > 0.2 else '' for _ in range(26)]
    return '\t'.join(['%d' % (i % 2)] + list(f) + c)

  paths = []
  for k, (n, eol) in enumerate(((150, '\n'), (77, '\r\n'))):
    p = tmp_path / ('part%d.tsv' % k)
    rows = [line(i) for i in range(n)]
    rows.insert(10, '')  # a blank line
    p.write_text(eol.join(rows) + (eol if k == 0 else ''))  # the second file does not end with a newline
    paths.append(str(p))
  got = {}
  for native in ('1', '0'):
    monkeypatch.setenv('EASYREC_AMD_NATIVE_CSV', native)
    inp = CSVInput(cfg.data_config, feats, ','.join(paths), batch_size=B, hash_on_host=False)
    assert inp._native_ok() == (native == '1')
    t0 = time.perf_counter()
    got[native] = list(inp.batches(drop_remainder=drop))
    dt = time.perf_counter() - t0
    print('native=%s: %d batches of %d in %.1f ms' % (native, len(got[native]), B, dt * 1e3))
  assert len(got['1']) == len(got['0']) == (3 if drop else 4)
  for a, b in zip(got['1'], got['0']):
    assert set(a) == set(b)
    for k in a:
      assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def _random_cells(rng, kind):
  """what a cell of a numeric / string column may hold: plain and signed decimals, long mantissas, exponents, leading
  zeros and dots (the fast path's edges and what it must hand to strtoll / strtod), empty cells"""
  if kind == 0:
    return rng.choice(['', 'abc', '%08x' % rng.integers(0, 2**32), 'x' * int(rng.integers(1, 40))])
  if kind == 1:
    return rng.choice(['', '0', '-0', '+7', '%d' % rng.integers(-10**6, 10**6), '%d' % rng.integers(-2**62, 2**62),
                       '000123', '999999999999999999', '-999999999999999999', '1234567890123456789'])
  return rng.choice(['', '0', '-0.0', '.5', '5.', '+.25', '%.6f' % rng.normal(), '%d' % rng.integers(-500, 500), '1e2', '-3.5E-3',
                     '123456789012345', '1234567890123456', '0.1234567890123456789', '%.15g' % rng.normal(),
                     '%.17g' % rng.normal(), '0000.5000', '3.', 'inf', '1' + '0' * 25, '0.' + '0' * 23 + '1'])


@pytest.mark.parametrize('threads', [0, 2, 5])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_threaded_csv_decode_equals_the_single_pass_decoder(built_lib, threads, seed):
  """er_decode_csv_host_mt (lines found in one pass, row ranges parsed by host threads, inline fast paths for plain
  decimals) against er_decode_csv_host (one thread, strtoll / strtod on every cell): every output array bit for bit -
  doubles compared as bit patterns -, the row count, the consumed bytes; CRLF, blank lines, an unterminated last line,
  max_rows below and above the number of lines.  Host code: no device needed."""
  from easyrec_amd import kernels
  be = kernels.HipBackend.__new__(kernels.HipBackend)  # (host entry points only: no device is touched)
  import ctypes
  be.lib = ctypes.CDLL(built_lib)
  be.lib.er_last_error.restype = ctypes.c_char_p
  rng = np.random.default_rng(seed)
  kinds = [int(k) for k in rng.integers(0, 3, size=9)]
  n_lines = 1500
  rows = []
  for i in range(n_lines):
    rows.append('\t'.join(str(_random_cells(rng, k)) for k in kinds))
    if rng.random() < 0.03:
      rows.append('')  # blank line
  eol = '\r\n' if seed == 1 else '\n'
  body = eol.join(rows) + (eol if seed != 2 else '')  # seed 2: the last line is not terminated
  # (a line made of empty cells only is not blank: it has separators)
  text = np.frombuffer(body.encode('utf-8'), dtype=np.uint8)
  n_terminated = sum(1 for r in (rows if seed != 2 else rows[:-1]) if r != '')
  for max_rows in (100, 700, 5000):
    one = be.decode_csv_host(text, '\t', kinds, max_rows, threads=1)
    many = be.decode_csv_host(text, '\t', kinds, max_rows, threads=threads)
    n = one[0]
    assert n == many[0] == min(max_rows, n_terminated) and one[1] == many[1]
    for a, b, what in zip(one[2:], many[2:], ('ints', 'floats', 'empty', 'begin', 'length')):
      a, b = a[:, :n], b[:, :n]
      if what == 'floats':
        a, b = a.view(np.int64), b.view(np.int64)
      assert np.array_equal(a, b), (what, max_rows)


def test_threaded_csv_decode_reports_the_first_bad_line(built_lib):
  """a line with too few fields / a cell that is not a number: the same error, naming the SMALLEST failing line, whatever
  thread met it"""
  from easyrec_amd import kernels
  import ctypes
  be = kernels.HipBackend.__new__(kernels.HipBackend)
  be.lib = ctypes.CDLL(built_lib)
  be.lib.er_last_error.restype = ctypes.c_char_p
  good = '1\t2.5\tabc'
  for bad, what in (('1\t2.5', 'fewer than 3 fields'), ('1\t2.5\tabc\tx', 'more than 3 fields'), ('1\tzz\tabc', "'zz' is not a number")):
    rows = [good] * 2000
    rows[1500] = bad
    rows[700] = bad
    text = np.frombuffer(('\n'.join(rows) + '\n').encode('utf-8'), dtype=np.uint8)
    msgs = []
    for threads in (1, 4):
      with pytest.raises(RuntimeError) as e:
        be.decode_csv_host(text, '\t', [1, 2, 0], 4096, threads=threads)
      msgs.append(str(e.value).split('failed')[-1])
    assert 'line 700' in msgs[0] and what in msgs[0]
    assert msgs[0].split(':', 1)[-1].strip() == msgs[1].split(':', 1)[-1].strip()


def test_integer_columns_are_hashed_as_their_decimal_strings_by_one_native_pass(built_lib):
  """er_pack_int_decimal_host: the packed decimal strings of an int64 array are Python's str(int) of every value (the
  reference's `_as_string` of an integer column, input.py:356-376) - digits two at a time, 32-bit arithmetic once they
  fit: the boundaries of both, negative values, the int64 extremes."""
  import ctypes
  from easyrec_amd import kernels
  be = kernels.HipBackend.__new__(kernels.HipBackend)
  be.lib = ctypes.CDLL(built_lib)
  be.lib.er_last_error.restype = ctypes.c_char_p
  rng = np.random.default_rng(3)
  edge = [0, -1, 9, 10, 99, 100, 101, 999, 1000, 2**32 - 1, 2**32, 2**32 + 1, 4294967295000, 10**18, -10**18, 2**63 - 1, -2**63]
  vals = np.concatenate([rng.integers(0, 2**32, size=5000), rng.integers(-2**63, 2**63 - 1, size=2000, dtype=np.int64),
                         np.array(edge, dtype=np.int64)]).astype(np.int64)
  data, offs = be.pack_int_decimal_host(vals)
  raw = data.tobytes()
  assert offs[0] == 0 and offs[-1] == len(raw)
  for i, v in enumerate(vals):
    assert raw[offs[i]:offs[i + 1]] == str(int(v)).encode('ascii'), (i, int(v))
  empty, eo = be.pack_int_decimal_host(np.zeros(0, dtype=np.int64))
  assert empty.size == 0 and list(eo) == [0]


@pytest.mark.parametrize('config', ['mmoe_taobao_small.config', 'din_taobao_small.config'])
@pytest.mark.parametrize('packed', [False, True])
def test_native_tag_and_sequence_split_equals_the_per_row_path(built_lib, config, packed, monkeypatch):
  """er_split_cells_host (TagFeature: tf.string_split, every separator byte a delimiter, empty tokens skipped;
  SequenceFeature: tf.strings.split, empty tokens kept, an empty cell = one empty token, max_seq_len cut) + ONE hash call
  per feature against the per-row Python code: every array of the batch dict, for list-of-str columns and for PackedCol
  views of a text buffer; empty cells, leading / trailing / doubled separators, multi-byte UTF-8 tokens, sequences longer
  than max_seq_len.  Host code only."""
  from easyrec_amd import kernels
  from easyrec_amd.input.input import Input, PackedCol, pack_strings
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
  monkeypatch.setattr(kernels, '_BACKEND', None)
  be = kernels.HipBackend.__new__(kernels.HipBackend)  # (host entry points only: no device is touched)
  import ctypes
  be.lib = ctypes.CDLL(built_lib)
  be.lib.er_last_error.restype = ctypes.c_char_p
  monkeypatch.setattr(kernels, '_BACKEND', be)
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
  feats = list(cfg.feature_config.features)
  B = 64
  inp = Input(cfg.data_config, feats, batch_size=B, hash_on_host=True)
  rng = np.random.default_rng(5)
  words = ['a', 'bb', '1234', 'x' * 20, 'é', '日本', '0', '']
  split_feats = [f for f in feats if f.feature_type in (FeatureConfig.TagFeature, FeatureConfig.SequenceFeature)]
  assert split_feats

  def cell(sep, n_max):
    k = int(rng.integers(0, n_max))
    toks = [str(rng.choice(words)) for _ in range(k)]
    c = sep.join(toks)
    r = rng.random()
    return sep + c if r < 0.1 else c + sep if r < 0.2 else c.replace(sep, sep + sep, 1) if r < 0.3 else c

  gen = SyntheticBatches(cfg.data_config, feats, batch_size=B, seed=3)
  cols = {k: list(v) if not isinstance(v, np.ndarray) else v for k, v in gen.raw_columns().items()} \
      if hasattr(gen, 'raw_columns') else None
  if cols is None:  # build the columns by hand: every input field a plausible value, the split columns the edge cases
    cols = {}
    for f in cfg.data_config.input_fields:
      cols[f.input_name] = ['%d' % rng.integers(0, 50) for _ in range(B)]
  for f in split_feats:
    n_max = 70 if f.feature_type == FeatureConfig.SequenceFeature else 6
    cols[f.input_names[0]] = [cell(f.separator or '|', n_max) for _ in range(B)]
    cols[f.input_names[0]][0] = ''
  if packed:  # the columns as PackedCol views of one text buffer, as the native CSV decoder hands them over
    for f in split_feats:
      data, offs = pack_strings(cols[f.input_names[0]])
      cols[f.input_names[0]] = PackedCol(data, offs[:-1].copy(), np.diff(offs).astype(np.int32))
  outs, calls = [], []
  real_split = be.split_cells_host
  be.split_cells_host = lambda *a, **kw: (calls.append(1), real_split(*a, **kw))[1]
  for native in (True, False):
    inp.native_split = native
    outs.append(inp.preprocess({k: v for k, v in cols.items()}))
    if native:
      assert len(calls) == len(split_feats)  # (every tag / sequence column took the native pass)
  assert len(calls) == len(split_feats)
  a, b = outs
  assert set(a) == set(b)
  checked = 0
  for k in a:
    assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    checked += k.startswith('tag/') or k.startswith('seq/')
  assert checked >= 2 * len(split_feats)


@pytest.mark.parametrize('file_shard', [False, True])
def test_csv_workers_read_disjoint_parts_of_the_data(tmp_path, built_lib, file_shard, monkeypatch):
  """One process per GPU: worker r of W takes line k of the data set when k % W == r (reference
  `_safe_shard`, input/input.py:1018-1023 after csv_input.py:126-127), or whole files with data_config.file_shard
  (:109-110).  Both decode paths; the workers' rows together are the data set, in order, without overlap."""
  from easyrec_amd.input.csv_input import CSVInput
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
  cfg.data_config.file_shard = file_shard
  feats = list(cfg.feature_config.features)
  W, B, n_files, per_file = 3, 8, 3, 40
  rng = np.random.default_rng(1)
  paths, labels_of_file = [], []
  for k in range(n_files):
    rows, labs = [], []
    for i in range(per_file):
      lab = int(rng.integers(0, 1000))  # the label column doubles as a row id
      labs.append(lab)
      f = ['%d' % rng.integers(0, 50) for _ in range(13)]
      c = ['%04x' % rng.integers(0, 2**16) for _ in range(26)]
      rows.append('\t'.join(['%d' % lab] + f + c))
    p = tmp_path / ('p%d.tsv' % k)
    p.write_text('\n'.join(rows) + '\n')
    paths.append(str(p))
    labels_of_file.append(labs)
  whole = [lab for labs in labels_of_file for lab in labs]
  for native in ('1', '0'):
    monkeypatch.setenv('EASYREC_AMD_NATIVE_CSV', native)
    seen = []
    for r in range(W):
      inp = CSVInput(cfg.data_config, feats, ','.join(paths), batch_size=B, hash_on_host=True, task_index=r, task_num=W)
      got = [int(x) for b in inp.batches(drop_remainder=True) for x in b['labels'][0]]
      if file_shard:
        exp = [lab for k in range(n_files) if k % W == r for lab in labels_of_file[k]]
      else:
        exp = whole[r::W]
      assert got == exp[:len(exp) // B * B], (native, r)
      seen.append(got)
    assert sum(len(g) for g in seen) == sum(len(whole[r::W]) // B * B for r in range(W)) or file_shard


def test_prefetcher_keeps_order_and_hands_over_errors():
  import threading
  import time
  from easyrec_amd.input.prefetch import Prefetcher
  assert list(Prefetcher(range(50), depth=3)) == list(range(50))
  assert list(Prefetcher(range(5), depth=1, transform=lambda x: x * x)) == [0, 1, 4, 9, 16]

  def bad():
    yield 1
    yield 2
    raise ValueError('boom')

  p = Prefetcher(bad(), depth=2)
  assert next(p) == 1 and next(p) == 2
  with pytest.raises(ValueError):
    next(p)
  # the producer runs ahead of a slow consumer, but no further than `depth`
  produced = []

  def src():
    for i in range(10):
      produced.append(i)
      yield i

  p = Prefetcher(src(), depth=2)
  time.sleep(0.3)
  assert 2 <= len(produced) <= 4  # depth items queued + one in hand (+ one being offered)
  assert next(p) == 0
  p.close()
  time.sleep(0.3)
  assert len(produced) < 10 and threading.active_count() < 10

"""easyrec_amd's FeatureColumnParser against the REFERENCE'S OWN parser (feature_column/feature_column.py:41-664), whose
output on a recording stand-in for the feature-column constructor API is stored in
tests/golden/feature_column_vectors.json (generator: tests/golden/make_feature_column_vectors.py, run where
/root/reference exists): per feature the table shape, id source, weight input, boundaries, combiner, shared-embedding
name, max_seq_length, sequence_combiner - for a zoo of feature configs covering the parser's branches, with and
without ev_params, and for every fixture under configs/."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'feature_column_vectors.json')) as f:
  CASES = {c['tag']: c for c in json.load(f)['cases']}


def _describe(col):
  from easyrec_amd.feature_column.feature_column import EmbeddingColumn, NumericColumn, SequenceNumericColumn
  if isinstance(col, NumericColumn):
    return {'type': 'numeric', 'key': col.key, 'shape': int(col.shape)}
  if isinstance(col, SequenceNumericColumn):
    return {'type': 'seq_numeric', 'key': col.key, 'sequence_length': int(col.sequence_length)}
  assert isinstance(col, EmbeddingColumn), col
  c = col.categorical_column
  if isinstance(c, NumericColumn):
    cat = {'kind': 'numeric', 'key': c.key}
  else:
    cat = {'kind': c.kind, 'key': c.key, 'num_buckets': int(c.num_buckets), 'weight_key': c.weight_key,
           'boundaries': [float(b) for b in (c.boundaries or [])], 'is_sequence': bool(c.is_sequence),
           'vocabulary': list(c.vocabulary or [])}
  return {'type': 'embedding', 'column': cat, 'dimension': int(col.dimension), 'combiner': col.combiner,
          'shared_name': col.shared_name, 'max_seq_length': int(col.max_seq_length),
          'sequence_combiner': None if col.sequence_combiner is None else str(col.sequence_combiner).strip(),
          'has_initializer': col.initializer is not None, 'partitioned': col.max_partitions > 1}


@pytest.mark.parametrize('tag', sorted(CASES))
def test_parser_builds_what_the_reference_parser_builds(tag):
  from google.protobuf import text_format

  from easyrec_amd.feature_column.feature_column import FeatureColumnParser
  from easyrec_amd.protos import feature_config_pb2
  case = CASES[tag]
  fcfg = feature_config_pb2.FeatureConfigV2()
  text_format.Merge(case['features'], fcfg)
  ev = None
  if case['ev_params']:
    ev = feature_config_pb2.EVParams()
    text_format.Merge(case['ev_params'], ev)
  wd = {k: getattr(feature_config_pb2.WideOrDeep, v) for k, v in case['wide_deep'].items()}
  parser = FeatureColumnParser(list(fcfg.features), wd, case['wide_output_dim'], ev_params=ev)
  for part, cols in (('wide', parser.wide_columns), ('deep', parser.deep_columns), ('sequence', parser.sequence_columns)):
    want = case[part]
    assert list(cols) == list(want) or sorted(cols) == sorted(want), (part, sorted(cols), sorted(want))
    for name in want:
      assert _describe(cols[name]) == want[name], (part, name, _describe(cols[name]), want[name])
  from easyrec_amd.feature_column.feature_group import FeatureGroup
  by_id = {id(v): k for d in (parser.wide_columns, parser.deep_columns, parser.sequence_columns) for k, v in d.items()}
  for gname, want in case['groups'].items():  # the columns a group selects, in the order of its output
    gcfg = feature_config_pb2.FeatureGroupConfig()
    text_format.Merge(want['text'], gcfg)
    plain, seqs = FeatureGroup(gcfg).select_columns(parser)
    assert [by_id[id(c)] for c in plain] == want['plain'] and [by_id[id(c)] for c in seqs] == want['sequence'], gname
  for name, n in case['vocab_size'].items():
    assert parser.get_feature_vocab_size(name) == n, (name, parser.get_feature_vocab_size(name), n)


def test_every_small_fixture_config_has_a_pinned_case():
  """A config added under configs/ without re-running tests/golden/make_feature_column_vectors.py (where /root/reference
  exists) would be a fixture the reference's parser never saw: every `*_small.config` must have its case (the full-size twins
  differ only in table sizes and are not cases of their own)."""
  import glob
  root = os.path.dirname(HERE)
  names = sorted(os.path.basename(p)[:-len('.config')] for p in glob.glob(os.path.join(root, 'configs', '*_small.config')))
  assert names, 'no fixture configs found'
  missing = [n for n in names if n not in CASES]
  assert not missing, 'configs without a feature-column case (regenerate the fixture): %s' % missing

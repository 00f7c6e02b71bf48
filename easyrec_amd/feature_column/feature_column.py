"""FeatureConfig -> column objects (the static lookup plan of the embedding stage).

Host-side mirror of reference easy_rec/python/feature_column/feature_column.py:41-664
(`FeatureColumnParser`) and of the column classes it instantiates from the forked TF feature-column
library (compat/feature_column/feature_column_v2.py: HashedCategoricalColumn :3870-3988,
IdentityCategoricalColumn :4226-4350, WeightedCategoricalColumn :4353-4478, BucketizedColumn :2762,
EmbeddingColumn :3364-3664, SharedEmbeddingColumn :3723, SequenceCategoricalColumn :4926-5058).
Only the *description* lives here (table shape, combiner, id source); the arithmetic is in the HIP
kernels (easyrec_amd/csrc/er_embedding.hip) driven by layers/input_layer.py.
"""
import logging
import sys
from dataclasses import dataclass, field
from typing import List, Optional

from easyrec_amd.protos.feature_config_pb2 import FeatureConfig, WideOrDeep

MAX_HASH_BUCKET_SIZE = 9223372036854775807


class FeatureKeyError(KeyError):
  pass


@dataclass
class NumericColumn:
  """tf numeric_column: the (normalised) value itself is the dense feature."""
  key: str
  shape: int
  feature_name: str

  @property
  def name(self):
    return self.feature_name if self.feature_name else self.key

  @property
  def raw_name(self):
    return self.feature_name if self.feature_name else self.key

  @property
  def dimension(self):
    return self.shape


@dataclass
class CategoricalColumn:
  """kind: 'hash' (string -> Fingerprint64 % buckets), 'identity' (int, out of range -> default),
  'vocab' (string -> index in vocabulary, OOV -> default), 'bucketized' (float -> bucket index)."""
  key: str
  kind: str
  num_buckets: int
  feature_name: str
  default_value: int = 0
  vocabulary: Optional[List[str]] = None
  boundaries: Optional[List[float]] = None
  weight_key: Optional[str] = None
  is_sequence: bool = False
  source_key: Optional[str] = None  # bucketized: the numeric source

  @property
  def base_name(self):
    if self.kind == 'bucketized':
      return '%s_bucketized' % (self.feature_name if self.feature_name else self.source_key)
    return self.feature_name if self.feature_name else self.key

  @property
  def name(self):
    if self.weight_key:
      return '%s_weighted_by_%s' % (self.base_name, self.weight_key)
    return self.base_name


@dataclass
class EmbeddingColumn:
  categorical_column: CategoricalColumn
  dimension: int
  combiner: str
  raw_name: str
  initializer: object = None
  shared_name: Optional[str] = None  # shared_embedding_collection_name
  max_seq_length: int = -1
  sequence_combiner: object = None
  ev_params: object = None
  max_partitions: int = 1

  @property
  def name(self):
    if self.shared_name:
      return '%s_shared_embedding' % self.categorical_column.name
    return '%s_embedding' % self.categorical_column.name

  @property
  def var_scope_name(self):
    return self.shared_name if self.shared_name else self.name

  @property
  def num_buckets(self):
    return self.categorical_column.num_buckets

  @property
  def is_sequence(self):
    return self.categorical_column.is_sequence


@dataclass
class SequenceNumericColumn:
  """sequence raw feature without embedding (sequence_numeric_column_with_raw_column)."""
  key: str
  feature_name: str
  sequence_length: int

  @property
  def name(self):
    return self.feature_name if self.feature_name else self.key

  @property
  def raw_name(self):
    return self.name


def is_embedding_column(fc):
  return isinstance(fc, EmbeddingColumn)


class FeatureColumnParser(object):
  """Parse FeatureConfigs into wide / deep / sequence columns."""

  def __init__(self, feature_configs, wide_deep_dict={}, wide_output_dim=-1, ev_params=None):
    self._feature_configs = feature_configs
    self._wide_output_dim = wide_output_dim
    self._wide_deep_dict = wide_deep_dict
    self._deep_columns = {}
    self._wide_columns = {}
    self._sequence_columns = {}
    self._feature_vocab_size = {}
    self._global_ev_params = ev_params

    # shared embeddings: configs with the same embedding_name used more than once
    # (reference feature_column.py:72-117)
    share_count, share_info = {}, {}
    for config in self._feature_configs:
      if not config.HasField('embedding_name'):
        continue
      name = config.embedding_name
      if name in share_count:
        a, b = config, share_info[name]
        assert (a.embedding_dim == b.embedding_dim and a.combiner == b.combiner and
                a.initializer == b.initializer and a.max_partitions == b.max_partitions), \
            'shared embed info of [%s] is not matched' % name
        share_count[name] += 1
        if config.feature_type == FeatureConfig.SequenceFeature:
          share_info[name] = config
      else:
        share_count[name] = 1
        share_info[name] = config
    self._share_embed_infos = {k: v for k, v in share_info.items() if share_count[k] > 1}
    logging.info('shared embeddings[num=%d]' % len(self._share_embed_infos))

    for config in self._feature_configs:
      assert isinstance(config, FeatureConfig)
      try:
        if config.feature_type == config.IdFeature:
          self.parse_id_feature(config)
        elif config.feature_type == config.TagFeature:
          self.parse_tag_feature(config)
        elif config.feature_type == config.RawFeature:
          self.parse_raw_feature(config)
        elif config.feature_type == config.ComboFeature:
          self.parse_combo_feature(config)
        elif config.feature_type == config.LookupFeature:
          self.parse_lookup_feature(config)
        elif config.feature_type == config.SequenceFeature:
          self.parse_sequence_feature(config)
        elif config.feature_type == config.ExprFeature:
          self.parse_expr_feature(config)
        elif config.feature_type != config.PassThroughFeature:
          assert False, 'invalid feature type: %s' % config.feature_type
      except FeatureKeyError:
        pass

  # -- accessors (same names as the reference)
  @property
  def wide_columns(self):
    return self._wide_columns

  @property
  def deep_columns(self):
    return self._deep_columns

  @property
  def sequence_columns(self):
    return self._sequence_columns

  def get_feature_vocab_size(self, feature):
    return self._feature_vocab_size.get(feature, 1)

  @staticmethod
  def _feature_name(config):
    return config.feature_name if config.HasField('feature_name') else config.input_names[0]

  def is_wide(self, config):
    name = self._feature_name(config)
    if name not in self._wide_deep_dict:
      raise FeatureKeyError(name)
    return self._wide_deep_dict[name] in [WideOrDeep.WIDE, WideOrDeep.WIDE_AND_DEEP]

  def is_deep(self, config):
    name = self._feature_name(config)
    if name not in self._wide_deep_dict:
      raise FeatureKeyError(name)
    return self._wide_deep_dict[name] in [WideOrDeep.DEEP, WideOrDeep.WIDE_AND_DEEP]

  def _get_hash_bucket_size(self, config):
    if not config.HasField('hash_bucket_size'):
      return -1
    if self._global_ev_params is not None or config.HasField('ev_params'):
      return MAX_HASH_BUCKET_SIZE
    return config.hash_bucket_size

  def _categorical(self, config, feature_name, is_sequence=False):
    """hash / vocab / identity categorical column (reference feature_column.py:259-294)."""
    hash_bucket_size = self._get_hash_bucket_size(config)
    if hash_bucket_size > 0:
      return CategoricalColumn(feature_name, 'hash', hash_bucket_size, feature_name,
                               is_sequence=is_sequence)
    if config.vocab_list:
      return CategoricalColumn(feature_name, 'vocab', len(config.vocab_list), feature_name,
                               vocabulary=list(config.vocab_list), is_sequence=is_sequence)
    if config.vocab_file:
      with open(config.vocab_file, 'r') as fin:
        vocab = [line.rstrip('\n') for line in fin]
      return CategoricalColumn(feature_name, 'vocab', len(vocab), feature_name, vocabulary=vocab,
                               is_sequence=is_sequence)
    use_ev = self._global_ev_params is not None or config.HasField('ev_params')
    num_buckets = sys.maxsize if use_ev else config.num_buckets
    return CategoricalColumn(feature_name, 'identity', num_buckets, feature_name,
                             is_sequence=is_sequence)

  def parse_id_feature(self, config):
    feature_name = self._feature_name(config)
    fc = self._categorical(config, feature_name)
    if self.is_wide(config):
      self._add_wide_embedding_column(fc, config)
    if self.is_deep(config):
      self._add_deep_embedding_column(fc, config)

  def parse_tag_feature(self, config):
    """reference feature_column.py:301-349."""
    feature_name = self._feature_name(config)
    fc = self._categorical(config, feature_name)
    if len(config.input_names) > 1 or config.HasField('kv_separator'):
      fc.weight_key = feature_name + '_w'
    if self.is_wide(config):
      self._add_wide_embedding_column(fc, config)
    if self.is_deep(config):
      self._add_deep_embedding_column(fc, config)

  def parse_raw_feature(self, config):
    """reference feature_column.py:351-405."""
    feature_name = self._feature_name(config)
    bounds = None
    if config.boundaries:
      bounds = sorted(config.boundaries)
    elif config.num_buckets > 1 and config.max_val > config.min_val:
      bounds = [x / float(config.num_buckets) for x in range(0, config.num_buckets)]
    if bounds:
      fc = CategoricalColumn(feature_name, 'bucketized', len(bounds) + 1, feature_name,
                             boundaries=bounds, source_key=feature_name)
      if self.is_wide(config):
        self._add_wide_embedding_column(fc, config)
      if self.is_deep(config):
        self._add_deep_embedding_column(fc, config)
      return
    # value-weighted projection: ids 0..raw_input_dim-1 weighted by the values
    wgt_fc = CategoricalColumn(feature_name + '_raw_proj_id', 'identity', config.raw_input_dim,
                               feature_name, weight_key=feature_name + '_raw_proj_val')
    if self.is_wide(config):
      self._add_wide_embedding_column(wgt_fc, config)
    if self.is_deep(config):
      if config.embedding_dim > 0:
        self._add_deep_embedding_column(wgt_fc, config)
      else:
        self._deep_columns[feature_name] = NumericColumn(feature_name, config.raw_input_dim,
                                                         feature_name)

  def parse_expr_feature(self, config):
    feature_name = self._feature_name(config)
    fc = NumericColumn(feature_name, 1, feature_name)
    if self.is_wide(config):
      self._add_wide_embedding_column(fc, config)
    if self.is_deep(config):
      self._deep_columns[feature_name] = fc

  def parse_combo_feature(self, config):
    """reference feature_column.py:424-455: crossed_column, or with combo_join_sep one hashed column."""
    feature_name = config.feature_name if config.HasField('feature_name') else None
    assert len(config.input_names) >= 2
    if len(config.combo_join_sep) == 0:
      # crossed_column(input names, hash_bucket_size, hash_key=None, feature_name) (:434-445): the crossed id comes
      # out of the input pipeline (input/input.py, er_sparse_cross_hashed_host); from here on an identity column
      # named feature_name (CrossedColumn.name, compat/feature_column/feature_column_v2.py:4501-4504)
      assert feature_name, 'ComboFeature needs feature_name'
      fc = CategoricalColumn(feature_name, 'identity', self._get_hash_bucket_size(config), feature_name)
    else:
      fc = CategoricalColumn(feature_name, 'hash', self._get_hash_bucket_size(config), feature_name)
    if self.is_wide(config):
      self._add_wide_embedding_column(fc, config)
    if self.is_deep(config):
      self._add_deep_embedding_column(fc, config)

  def parse_lookup_feature(self, config):
    feature_name = self._feature_name(config)
    assert config.HasField('hash_bucket_size')
    fc = CategoricalColumn(feature_name, 'hash', self._get_hash_bucket_size(config), feature_name)
    if self.is_wide(config):
      self._add_wide_embedding_column(fc, config)
    if self.is_deep(config):
      self._add_deep_embedding_column(fc, config)

  def parse_sequence_feature(self, config):
    """reference feature_column.py:478-564."""
    feature_name = self._feature_name(config)
    sub = config.sub_feature_type
    assert sub in [config.IdFeature, config.RawFeature], \
        'Current sub_feature_type only support IdFeature and RawFeature.'
    if sub == config.IdFeature:
      fc = self._categorical(config, feature_name, is_sequence=True)
    else:
      bounds = None
      if config.boundaries:
        bounds = sorted(config.boundaries)
      elif config.num_buckets > 1 and config.max_val > config.min_val:
        bounds = [x / float(config.num_buckets) for x in range(0, config.num_buckets)]
      if bounds:
        fc = CategoricalColumn(feature_name, 'bucketized', len(bounds) + 1, feature_name,
                               boundaries=bounds, source_key=feature_name, is_sequence=True)
      elif config.embedding_dim > 0:
        fc = CategoricalColumn(feature_name + '_raw_proj_id', 'identity', config.raw_input_dim,
                               feature_name, weight_key=feature_name + '_raw_proj_val',
                               is_sequence=True)
      else:
        fc = SequenceNumericColumn(feature_name, feature_name, config.sequence_length)
    if config.embedding_dim > 0:
      self._add_deep_embedding_column(fc, config)
    else:
      self._sequence_columns[feature_name] = fc

  def _ev_params_of(self, config):
    """The feature's own `ev_params`.  Model-level ev_params (feature_column.py:212-219) have been written into the
    hashed IdFeatures' configs by the estimator - the columns they turn into hash tables here; the other column kinds
    (raw-feature projections, vocab / identity ids, tags, sequences) keep dense tables."""
    return config.ev_params if config.HasField('ev_params') else None

  def _add_wide_embedding_column(self, fc, config):
    """Wide column = dim-`wide_output_dim` embedding with `sum` combiner (feature_column.py:596-623)."""
    feature_name = self._feature_name(config)
    assert self._wide_output_dim > 0, 'wide_output_dim is not set'
    shared, max_seq_length = None, -1
    if config.embedding_name in self._share_embed_infos:
      shared = config.embedding_name + '_wide'
      info = self._share_embed_infos[config.embedding_name]  # (the shared columns all carry the shared info's max_seq_len)
      max_seq_length = info.max_seq_len if info.HasField('max_seq_len') else -1
    self._wide_columns[feature_name] = EmbeddingColumn(
        fc, self._wide_output_dim, 'sum', feature_name,
        initializer=config.initializer if config.HasField('initializer') else None,
        shared_name=shared, max_seq_length=max_seq_length, max_partitions=config.max_partitions,
        ev_params=self._ev_params_of(config))

  def _add_deep_embedding_column(self, fc, config):
    """reference feature_column.py:625-656."""
    feature_name = self._feature_name(config)
    assert config.embedding_dim > 0, 'embedding_dim is not set for %s' % feature_name
    self._feature_vocab_size[feature_name] = fc.num_buckets
    if config.embedding_name in self._share_embed_infos:
      info = self._share_embed_infos[config.embedding_name]
      col = EmbeddingColumn(
          fc, info.embedding_dim, info.combiner, feature_name,
          initializer=info.initializer if info.HasField('initializer') else None,
          shared_name=config.embedding_name,
          max_seq_length=info.max_seq_len if info.HasField('max_seq_len') else -1,
          max_partitions=info.max_partitions)
    else:
      col = EmbeddingColumn(
          fc, config.embedding_dim, config.combiner, feature_name,
          initializer=config.initializer if config.HasField('initializer') else None,
          max_seq_length=config.max_seq_len if config.HasField('max_seq_len') else -1,
          max_partitions=config.max_partitions, ev_params=self._ev_params_of(config))
    if config.feature_type != config.SequenceFeature:
      self._deep_columns[feature_name] = col
    else:
      if config.HasField('sequence_combiner'):
        col.sequence_combiner = config.sequence_combiner
      self._sequence_columns[feature_name] = col

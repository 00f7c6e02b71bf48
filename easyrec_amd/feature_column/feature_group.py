"""Feature groups: ordered column selection (reference easy_rec/python/feature_column/feature_group.py:9-60)."""
import re

from easyrec_amd.protos.feature_config_pb2 import FeatureGroupConfig, WideOrDeep


class FeatureGroup(object):

  def __init__(self, feature_group_config):
    self._config = feature_group_config
    assert isinstance(self._config, FeatureGroupConfig)
    assert self._config.wide_deep in [WideOrDeep.WIDE, WideOrDeep.DEEP]
    self._feature_names = self._auto_expand_feature_name()

  @property
  def group_name(self):
    return self._config.group_name

  @property
  def config(self):
    return self._config

  @property
  def wide_and_deep_dict(self):
    return {name: self._config.wide_deep for name in self._feature_names}

  @property
  def feature_names(self):
    return self._feature_names

  def select_columns(self, fc):
    """Config order is the output order (feature_group.py:32-44)."""
    if self._config.wide_deep == WideOrDeep.WIDE:
      return [fc.wide_columns[x] for x in self._feature_names], []
    sequence_columns, deep_columns = [], []
    for x in self._feature_names:
      if x in fc.sequence_columns:
        sequence_columns.append(fc.sequence_columns[x])
      else:
        deep_columns.append(fc.deep_columns[x])
    return deep_columns, sequence_columns

  def _auto_expand_feature_name(self):
    """`F[1-13]` -> F1..F13 (feature_group.py:46-60)."""
    names = []
    for feature in self._config.feature_names:
      m = re.match(r'([a-zA-Z_]+)\[([0-9]+)-([0-9]+)\]', feature)
      if m:
        prefix, sid, eid = m.group(1), int(m.group(2)), int(m.group(3)) + 1
        names.extend('%s%d' % (prefix, t) for t in range(sid, eid))
      else:
        names.append(feature)
    return names

"""Feature groups: ordered column selection (reference easy_rec/python/feature_column/feature_group.py:9-60)."""
import re

from easyrec_amd.protos.feature_config_pb2 import FeatureGroupConfig, WideOrDeep

_RANGE = re.compile(r'([a-zA-Z_]+)\[([0-9]+)-([0-9]+)\]')  # `F[1-13]` stands for F1 .. F13 (feature_group.py:46-60)


def _expand(names):
  out = []
  for name in names:
    m = _RANGE.match(name)
    out.extend(['%s%d' % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)] if m else [name])
  return out


class FeatureGroup(object):

  def __init__(self, feature_group_config):
    if not isinstance(feature_group_config, FeatureGroupConfig):
      raise AssertionError('FeatureGroupConfig expected')
    if feature_group_config.wide_deep not in (WideOrDeep.WIDE, WideOrDeep.DEEP):
      raise AssertionError('a feature group is WIDE or DEEP')
    self._config = feature_group_config
    self._feature_names = _expand(feature_group_config.feature_names)

  group_name = property(lambda self: self._config.group_name)
  config = property(lambda self: self._config)
  feature_names = property(lambda self: self._feature_names)

  @property
  def wide_and_deep_dict(self):
    return dict.fromkeys(self._feature_names, self._config.wide_deep)

  def select_columns(self, fc):
    """(plain columns, sequence columns) of the group in CONFIG order - the order of the group's output
    (feature_group.py:32-44); a wide group has no sequence columns."""
    if self._config.wide_deep == WideOrDeep.WIDE:
      return [fc.wide_columns[n] for n in self._feature_names], []
    sequences = fc.sequence_columns
    plain = [fc.deep_columns[n] for n in self._feature_names if n not in sequences]
    return plain, [sequences[n] for n in self._feature_names if n in sequences]

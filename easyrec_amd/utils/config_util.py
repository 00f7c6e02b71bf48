"""Load EasyRec pipeline configs unchanged (the drop-in boundary).

Behaviour spec: reference easy_rec/python/utils/config_util.py:46-136 (`get_configs_from_pipeline_file`: `.config` =
protobuf text, `.json` = protobuf JSON; `shared_names` expansion; `name[a-b]` range expansion) and :583-590
(`get_compatible_feature_configs`: the deprecated repeated `feature_configs` wins over `feature_config.features`).
"""
import os
import re

from google.protobuf import json_format
from google.protobuf import text_format

from easyrec_amd.protos import pipeline_pb2

_RANGE_NAME = re.compile(r'([a-zA-Z_]+)\[([0-9]+)-([0-9]+)\]')
_PARSERS = {
    '.config': lambda text, msg: text_format.Merge(text, msg),
    '.json': lambda text, msg: json_format.Parse(text, msg),
}


def get_configs_from_pipeline_file(pipeline_config_path, auto_expand=True):
  """An EasyRecConfig from a `.config` (prototxt) or `.json` file; an EasyRecConfig passes through."""
  if isinstance(pipeline_config_path, pipeline_pb2.EasyRecConfig):
    return pipeline_config_path
  path = pipeline_config_path
  if not os.path.exists(path):
    raise AssertionError('pipeline_config_path [%s] not exists' % path)
  parse = _PARSERS.get(os.path.splitext(path)[1])
  if parse is None:
    raise AssertionError('invalid file format(%s), currently support formats: .config(prototxt) .json' % path)
  config = pipeline_pb2.EasyRecConfig()
  with open(path, 'r') as f:
    parse(f.read(), config)
  return auto_expand_share_feature_configs(config) if auto_expand else config


def parse_pipeline_text(config_str, auto_expand=True):
  config = pipeline_pb2.EasyRecConfig()
  text_format.Merge(config_str, config)
  return auto_expand_share_feature_configs(config) if auto_expand else config


def get_compatible_feature_configs(pipeline_config):
  """The feature list in use: `feature_configs` (deprecated) when non-empty, else `feature_config.features`."""
  old_style = pipeline_config.feature_configs
  return old_style if len(old_style) > 0 else pipeline_config.feature_config.features


def auto_expand_names(input_name):
  """`field[1-3]` -> [field1, field2, field3]; any other name -> [name]."""
  m = _RANGE_NAME.match(input_name)
  if m is None:
    return [input_name]
  stem, first, last = m.group(1), int(m.group(2)), int(m.group(3))
  return [stem + str(k) for k in range(first, last + 1)]


def auto_expand_share_feature_configs(pipeline_config):
  """A FeatureConfig with `shared_names` stands for one more FeatureConfig per shared name (same settings, that name
  as its only input); the clones are appended to the list in use and `shared_names` is cleared on the template."""
  features = get_compatible_feature_configs(pipeline_config)
  expand_ranges = pipeline_config.data_config.auto_expand_input_fields
  templates = [fc for fc in features if len(fc.shared_names) > 0]  # snapshot: clones carry no shared_names
  for template in templates:
    names = []
    for shared in template.shared_names:
      names.extend(auto_expand_names(shared) if expand_ranges else [shared])
    template.ClearField('shared_names')
    for name in names:
      clone = features.add()
      clone.CopyFrom(template)
      clone.ClearField('input_names')
      clone.input_names.append(name)
  return pipeline_config


def save_pipeline_config(pipeline_config, path):
  with open(path, 'w') as f:
    f.write(text_format.MessageToString(pipeline_config, as_utf8=True))

"""Load EasyRec pipeline configs unchanged (the drop-in boundary).

Mirrors reference easy_rec/python/utils/config_util.py:46-136 (get_configs_from_pipeline_file,
auto_expand_share_feature_configs, auto_expand_names) and :583-610 (get_compatible_feature_configs).
"""
import os
import re

from google.protobuf import json_format
from google.protobuf import text_format

from easyrec_amd.protos import feature_config_pb2
from easyrec_amd.protos import pipeline_pb2


def get_configs_from_pipeline_file(pipeline_config_path, auto_expand=True):
  """Read a `.config` (prototxt) or `.json` EasyRecConfig (reference config_util.py:46-78)."""
  if isinstance(pipeline_config_path, pipeline_pb2.EasyRecConfig):
    return pipeline_config_path
  assert os.path.exists(pipeline_config_path), \
      'pipeline_config_path [%s] not exists' % pipeline_config_path
  pipeline_config = pipeline_pb2.EasyRecConfig()
  with open(pipeline_config_path, 'r') as f:
    config_str = f.read()
  if pipeline_config_path.endswith('.config'):
    text_format.Merge(config_str, pipeline_config)
  elif pipeline_config_path.endswith('.json'):
    json_format.Parse(config_str, pipeline_config)
  else:
    assert False, 'invalid file format(%s), currently support formats: .config(prototxt) .json' % \
        pipeline_config_path
  if auto_expand:
    return auto_expand_share_feature_configs(pipeline_config)
  return pipeline_config


def parse_pipeline_text(config_str, auto_expand=True):
  pipeline_config = pipeline_pb2.EasyRecConfig()
  text_format.Merge(config_str, pipeline_config)
  return auto_expand_share_feature_configs(pipeline_config) if auto_expand else pipeline_config


def get_compatible_feature_configs(pipeline_config):
  """`feature_configs` (deprecated, repeated) or `feature_config.features` (config_util.py:583-590)."""
  if pipeline_config.feature_configs:
    return pipeline_config.feature_configs
  return pipeline_config.feature_config.features


def auto_expand_names(input_name):
  """field[1-3] -> field1, field2, field3 (reference config_util.py:114-133)."""
  m = re.match(r'([a-zA-Z_]+)\[([0-9]+)-([0-9]+)\]', input_name)
  if m:
    prefix, sid, eid = m.group(1), int(m.group(2)), int(m.group(3)) + 1
    return ['%s%d' % (prefix, t) for t in range(sid, eid)]
  return [input_name]


def auto_expand_share_feature_configs(pipeline_config):
  """Expand `shared_names` into one FeatureConfig per name (reference config_util.py:81-111)."""
  feature_configs = get_compatible_feature_configs(pipeline_config)
  for share_config in list(feature_configs):
    if len(share_config.shared_names) == 0:
      continue
    input_names = []
    for input_name in share_config.shared_names:
      if pipeline_config.data_config.auto_expand_input_fields:
        input_names.extend(auto_expand_names(input_name))
      else:
        input_names.append(input_name)
    del share_config.shared_names[:]
    fea_config = feature_config_pb2.FeatureConfig()
    fea_config.CopyFrom(share_config)
    del fea_config.input_names[:]
    for tmp_name in input_names:
      tmp_config = feature_config_pb2.FeatureConfig()
      tmp_config.CopyFrom(fea_config)
      tmp_config.input_names.append(tmp_name)
      if pipeline_config.feature_configs:
        pipeline_config.feature_configs.append(tmp_config)
      else:
        pipeline_config.feature_config.features.append(tmp_config)
  return pipeline_config


def save_pipeline_config(pipeline_config, path):
  with open(path, 'w') as f:
    f.write(text_format.MessageToString(pipeline_config, as_utf8=True))

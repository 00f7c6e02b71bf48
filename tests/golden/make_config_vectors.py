#!/usr/bin/env python
"""What the REFERENCE'S OWN utils/config_util.py:get_configs_from_pipeline_file (auto-expansion of `shared_names` and
`F[1-13]` ranges included, :46-136) makes of every config the reference ships - run in the build container where
/root/reference exists.  The function is executed unmodified (tf.gfile -> open, the protos -> this package's message
classes, which ARE the reference's schema); the fixture (tests/golden/config_vectors.json) holds, per config, the
SHA-256 of the deterministic serialisation of the loaded message, plus the full text of three small ones.
tests/test_config_pins.py loads the same files with easyrec_amd/utils/config_util.py where the reference tree is
present (the 224 files are not copied into this repository) and always checks the three embedded ones.

usage: python tests/golden/make_config_vectors.py [/root/reference]
"""
import glob
import hashlib
import importlib.util
import json
import os
import sys
import types

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
  from google.protobuf import text_format

  from easyrec_amd import protos
  tf = types.ModuleType('tensorflow')
  tf.__version__ = '1.15.0'
  tf.gfile = types.SimpleNamespace(GFile=open, Exists=os.path.exists)
  tf.compat = types.SimpleNamespace(v1=tf)
  sys.modules['tensorflow'] = tf
  for name in ('tensorflow.python', 'tensorflow.python.lib', 'tensorflow.python.lib.io', 'tensorflow.python.lib.io.file_io', 'easy_rec',
               'easy_rec.python', 'easy_rec.python.protos', 'easy_rec.python.utils', 'easy_rec.python.utils.pai_util',
               'easy_rec.python.utils.hive_utils'):
    sys.modules[name] = types.ModuleType(name)
  sys.modules['tensorflow.python.lib.io'].file_io = sys.modules['tensorflow.python.lib.io.file_io']
  sys.modules['easy_rec.python.protos'].pipeline_pb2 = protos.pipeline_pb2
  sys.modules['easy_rec.python.protos.pipeline_pb2'] = protos.pipeline_pb2
  sys.modules['easy_rec.python.protos.feature_config_pb2'] = protos.feature_config_pb2
  sys.modules['easy_rec.python.utils'].pai_util = sys.modules['easy_rec.python.utils.pai_util']
  sys.modules['easy_rec.python.utils.hive_utils'].HiveUtils = object
  spec = importlib.util.spec_from_file_location('ref_config_util', os.path.join(REF, 'easy_rec/python/utils/config_util.py'))
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)
  files = sorted(glob.glob(REF + '/samples/model_config/*.config') + glob.glob(REF + '/examples/configs/*.config'))
  digests, embedded = {}, {}
  for f in files:
    cfg = ref.get_configs_from_pipeline_file(f)
    rel = os.path.relpath(f, REF)
    digests[rel] = hashlib.sha256(cfg.SerializePartialToString(deterministic=True)).hexdigest()  # (required fields may be unset)
  # three small inputs whose expansion is visible, embedded with the reference's result as text
  small = {
      'ranges': "data_config { input_fields { input_name: 'label' input_type: INT32 } auto_expand_input_fields: true "
                "input_fields { input_name: 'f[1-3]' input_type: DOUBLE } input_fields { input_name: 'c[1-2]' input_type: STRING } "
                "label_fields: 'label' } feature_config { features { input_names: 'f[1-3]' feature_type: RawFeature } "
                "features { input_names: 'c[1-2]' feature_type: IdFeature hash_bucket_size: 10 embedding_dim: 4 } }",
      'shared_names': "feature_config { features { input_names: 'a' shared_names: 'b' shared_names: 'g[1-2]' feature_type: IdFeature "
                      "hash_bucket_size: 10 embedding_dim: 4 } features { input_names: 'p' feature_type: RawFeature } }",
      'feature_configs_list': "feature_configs { input_names: 'x' shared_names: 'y' feature_type: IdFeature num_buckets: 5 embedding_dim: 2 }",
  }
  for tag, text in small.items():
    path = os.path.join('/tmp', 'cfgpin_%s.config' % tag)
    with open(path, 'w') as fh:
      fh.write(text)
    embedded[tag] = {'input': text, 'loaded': text_format.MessageToString(ref.get_configs_from_pipeline_file(path))}
  out = os.path.join(HERE, 'config_vectors.json')
  with open(out, 'w') as fh:
    json.dump({'generator': 'tests/golden/make_config_vectors.py', 'digests': digests, 'embedded': embedded}, fh, indent=1,
              sort_keys=True)
  print('wrote %s: %d configs, %d embedded' % (out, len(digests), len(embedded)))


if __name__ == '__main__':
  main()

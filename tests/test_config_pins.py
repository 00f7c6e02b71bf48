"""easyrec_amd/utils/config_util.py:get_configs_from_pipeline_file against the REFERENCE'S OWN (utils/config_util.py:46-136,
executed by tests/golden/make_config_vectors.py where /root/reference exists): the loaded message - with `shared_names`
and `name[1-13]` ranges expanded - digest by digest for every config the reference ships (where its tree is present),
and in full text for three embedded inputs (always)."""
import hashlib
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'
with open(os.path.join(HERE, 'golden', 'config_vectors.json')) as f:
  FIX = json.load(f)


@pytest.mark.parametrize('tag', sorted(FIX['embedded']))
def test_embedded_configs_expand_as_the_reference_expands_them(tag, tmp_path):
  from google.protobuf import text_format

  from easyrec_amd.utils import config_util
  path = tmp_path / ('%s.config' % tag)
  path.write_text(FIX['embedded'][tag]['input'])
  got = config_util.get_configs_from_pipeline_file(str(path))
  assert text_format.MessageToString(got) == FIX['embedded'][tag]['loaded']


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='reference tree not present')
def test_every_shipped_config_loads_to_the_same_message():
  from easyrec_amd.utils import config_util
  assert len(FIX['digests']) >= 220
  for rel, want in FIX['digests'].items():
    cfg = config_util.get_configs_from_pipeline_file(os.path.join(REFERENCE, rel))
    assert hashlib.sha256(cfg.SerializePartialToString(deterministic=True)).hexdigest() == want, rel

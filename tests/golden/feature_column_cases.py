"""Feature configs of the FeatureColumnParser pin: every fixture under configs/ (its feature_config + the wide / deep
dictionary its feature groups imply) and a zoo covering the parser's branches.  Read by
make_feature_column_vectors.py (reference side) - the fixture then carries the texts, so the test needs only the JSON."""
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ZOO = """
features { input_names: 'uid' feature_type: IdFeature hash_bucket_size: 1000 embedding_dim: 8 }
features { input_names: 'city' feature_type: IdFeature num_buckets: 50 embedding_dim: 4 combiner: 'sum' }
features { input_names: 'os' feature_type: IdFeature vocab_list: ['ios', 'android', 'web'] embedding_dim: 4 }
features { feature_name: 'gender_named' input_names: 'gender' feature_type: IdFeature num_buckets: 3 embedding_dim: 2 }
features { input_names: 'tags' feature_type: TagFeature hash_bucket_size: 500 embedding_dim: 8 separator: '|' }
features { input_names: 'tags_kv' feature_type: TagFeature hash_bucket_size: 500 embedding_dim: 8 kv_separator: ':' combiner: 'sum' }
features { input_names: 'tags_w' input_names: 'tags_w_weight' feature_type: TagFeature num_buckets: 100 embedding_dim: 4 }
features { input_names: 'price' feature_type: RawFeature }
features { input_names: 'ctr_7d' feature_type: RawFeature embedding_dim: 4 }
features { input_names: 'emb_in' feature_type: RawFeature raw_input_dim: 3 embedding_dim: 6 }
features { input_names: 'age' feature_type: RawFeature boundaries: [30, 18, 45, 60] embedding_dim: 4 }
features { input_names: 'score' feature_type: RawFeature num_buckets: 10 min_val: 0 max_val: 5 embedding_dim: 4 }
features { input_names: 'dense3' feature_type: RawFeature raw_input_dim: 3 }
features { feature_name: 'uid_x_city' input_names: 'uid' input_names: 'city' feature_type: ComboFeature hash_bucket_size: 2000 embedding_dim: 8 }
features { feature_name: 'os_city' input_names: 'os' input_names: 'city' feature_type: ComboFeature hash_bucket_size: 300
           embedding_dim: 4 combo_join_sep: '_' }
features { feature_name: 'kv_lookup' input_names: 'kv_map' input_names: 'city' feature_type: LookupFeature hash_bucket_size: 400
           embedding_dim: 4 }
features { input_names: 'click_seq' feature_type: SequenceFeature sub_feature_type: IdFeature hash_bucket_size: 1000
           embedding_dim: 8 separator: '|' }
features { input_names: 'cate_seq' feature_type: SequenceFeature sub_feature_type: IdFeature num_buckets: 30 embedding_dim: 4
           sequence_combiner { attention {} } max_seq_len: 20 }
features { input_names: 'price_seq' feature_type: SequenceFeature sub_feature_type: RawFeature boundaries: [1, 5, 10]
           embedding_dim: 4 }
features { input_names: 'dur_seq' feature_type: SequenceFeature sub_feature_type: RawFeature embedding_dim: 4 }
features { input_names: 'raw_seq' feature_type: SequenceFeature sub_feature_type: RawFeature sequence_length: 12 }
features { input_names: 'item_a' feature_type: IdFeature hash_bucket_size: 777 embedding_dim: 8 embedding_name: 'item' }
features { input_names: 'item_b' feature_type: IdFeature hash_bucket_size: 777 embedding_dim: 8 embedding_name: 'item' }
features { input_names: 'item_hist' feature_type: SequenceFeature sub_feature_type: IdFeature hash_bucket_size: 777
           embedding_dim: 8 embedding_name: 'item' max_seq_len: 15 }
features { input_names: 'lonely' feature_type: IdFeature hash_bucket_size: 60 embedding_dim: 4 embedding_name: 'only_me' }
features { input_names: 'initialised' feature_type: IdFeature hash_bucket_size: 60 embedding_dim: 4 max_partitions: 4
           initializer { truncated_normal_initializer { stddev: 0.02 } } }
features { input_names: 'expr' feature_type: ExprFeature expression: 'price > 3' }
features { input_names: 'not_in_any_group' feature_type: IdFeature hash_bucket_size: 10 embedding_dim: 2 }
"""

ZOO_GROUPS = {'uid': 'WIDE_AND_DEEP', 'city': 'DEEP', 'os': 'WIDE', 'gender_named': 'DEEP', 'tags': 'WIDE_AND_DEEP', 'tags_kv': 'DEEP',
              'tags_w': 'DEEP', 'price': 'WIDE_AND_DEEP', 'ctr_7d': 'DEEP', 'emb_in': 'DEEP', 'age': 'WIDE_AND_DEEP', 'score': 'DEEP',
              'dense3': 'DEEP', 'uid_x_city': 'WIDE_AND_DEEP', 'os_city': 'DEEP', 'kv_lookup': 'DEEP', 'click_seq': 'DEEP',
              'cate_seq': 'DEEP', 'price_seq': 'DEEP', 'dur_seq': 'DEEP', 'raw_seq': 'DEEP', 'item_a': 'WIDE_AND_DEEP',
              'item_b': 'WIDE_AND_DEEP', 'item_hist': 'DEEP', 'lonely': 'DEEP', 'initialised': 'WIDE_AND_DEEP', 'expr': 'WIDE_AND_DEEP'}

EV_FEATURES = """
features { input_names: 'uid' feature_type: IdFeature hash_bucket_size: 1000 embedding_dim: 8 }
features { input_names: 'city' feature_type: IdFeature num_buckets: 50 embedding_dim: 4 ev_params { filter_freq: 2 } }
features { input_names: 'tags' feature_type: TagFeature hash_bucket_size: 500 embedding_dim: 8 }
features { input_names: 'click_seq' feature_type: SequenceFeature sub_feature_type: IdFeature hash_bucket_size: 1000 embedding_dim: 8 }
features { input_names: 'price' feature_type: RawFeature embedding_dim: 4 }
"""


def _fixture_cases():
  from google.protobuf import text_format

  from easyrec_amd.utils import config_util
  for path in sorted(glob.glob(os.path.join(ROOT, 'configs', '*.config'))):
    name = os.path.basename(path)
    if not name.endswith('_small.config'):
      continue  # (the full-size twins differ only in table sizes)
    cfg = config_util.get_configs_from_pipeline_file(path)
    wd = {}
    for group in cfg.model_config.feature_groups:
      kind = {0: 'DEEP', 1: 'WIDE', 2: 'WIDE_AND_DEEP'}[int(group.wide_deep)]
      for f in group.feature_names:
        wd[f] = kind if wd.get(f, kind) == kind else 'WIDE_AND_DEEP'
    for att in cfg.model_config.seq_att_groups:
      for m in att.seq_att_map:
        for f in list(m.key) + list(m.hist_seq):
          wd.setdefault(f, 'DEEP')
    wide_dim = 1 if any(v != 'DEEP' for v in wd.values()) else -1
    text = text_format.MessageToString(cfg.feature_config) if len(cfg.feature_config.features) else \
        ''.join('features { %s }\n' % text_format.MessageToString(f, as_one_line=True) for f in cfg.feature_configs)
    groups = [text_format.MessageToString(g, as_one_line=True) for g in cfg.model_config.feature_groups]
    yield name[:-len('.config')], text, wd, wide_dim, '', groups


def cases():
  zoo_groups = ["group_name: 'deep' wide_deep: DEEP " + ' '.join("feature_names: '%s'" % k for k in ZOO_GROUPS if ZOO_GROUPS[k] != 'WIDE'),
                "group_name: 'wide' wide_deep: WIDE " + ' '.join("feature_names: '%s'" % k for k in ZOO_GROUPS
                                                                 if ZOO_GROUPS[k] != 'DEEP' and 'seq' not in k),
                "group_name: 'ranged' wide_deep: DEEP feature_names: 'tags' feature_names: 'item_[a-b]' feature_names: 'uid'"]
  # (`item_[a-b]` does not match the digits-only range syntax: kept as a literal name and not selected below)
  yield 'zoo', ZOO, ZOO_GROUPS, 1, '', zoo_groups[:2]
  yield 'zoo_wide4', ZOO, ZOO_GROUPS, 4, '', zoo_groups[:2]
  yield 'feature_ev_params', EV_FEATURES, {k: 'DEEP' for k in ('uid', 'city', 'tags', 'click_seq', 'price')}, -1, '', []
  yield 'global_ev_params', EV_FEATURES, {k: 'DEEP' for k in ('uid', 'city', 'tags', 'click_seq', 'price')}, -1, 'filter_freq: 3', []
  for c in _fixture_cases():
    yield c

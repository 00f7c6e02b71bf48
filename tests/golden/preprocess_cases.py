"""Raw columns, their schema and the feature configs of the preprocessing pin (make_preprocess_vectors.py runs the
reference's Input._parse_* on them; tests/test_preprocess_pins.py runs easyrec_amd/input/input.py)."""

DATA_CONFIG = """
  input_fields { input_name: 'label' input_type: INT32 }
  input_fields { input_name: 'uid' input_type: STRING }
  input_fields { input_name: 'item' input_type: INT64 }
  input_fields { input_name: 'city' input_type: STRING }
  input_fields { input_name: 'level' input_type: INT32 }
  input_fields { input_name: 'price' input_type: DOUBLE }
  input_fields { input_name: 'ctr' input_type: STRING }
  input_fields { input_name: 'vec' input_type: STRING }
  input_fields { input_name: 'age' input_type: FLOAT }
  input_fields { input_name: 'tags' input_type: STRING }
  input_fields { input_name: 'tags_kv' input_type: STRING }
  input_fields { input_name: 'tag_ids' input_type: STRING }
  input_fields { input_name: 'tag_w' input_type: STRING }
  input_fields { input_name: 'clicks' input_type: STRING }
  input_fields { input_name: 'cates' input_type: STRING }
  input_fields { input_name: 'prices' input_type: STRING }
  input_fields { input_name: 'kv_map' input_type: STRING }
  label_fields: 'label'
  batch_size: 6
"""

FEATURES = """
  features { input_names: 'uid' feature_type: IdFeature hash_bucket_size: 1000 embedding_dim: 4 }
  features { input_names: 'item' feature_type: IdFeature hash_bucket_size: 500 embedding_dim: 4 }
  features { input_names: 'city' feature_type: IdFeature num_buckets: 20 embedding_dim: 4 }
  features { input_names: 'level' feature_type: IdFeature num_buckets: 8 embedding_dim: 4 }
  features { input_names: 'price' feature_type: RawFeature min_val: 1.0 max_val: 101.0 }
  features { input_names: 'ctr' feature_type: RawFeature }
  features { input_names: 'vec' feature_type: RawFeature raw_input_dim: 3 separator: ',' }
  features { input_names: 'age' feature_type: RawFeature boundaries: [18, 30, 45] embedding_dim: 4 }
  features { input_names: 'tags' feature_type: TagFeature hash_bucket_size: 100 embedding_dim: 4 separator: '|;' }
  features { input_names: 'tags_kv' feature_type: TagFeature hash_bucket_size: 100 embedding_dim: 4 separator: '|' kv_separator: ':' }
  features { input_names: 'tag_ids' input_names: 'tag_w' feature_type: TagFeature num_buckets: 50 embedding_dim: 4 separator: ',' }
  features { input_names: 'clicks' feature_type: SequenceFeature sub_feature_type: IdFeature hash_bucket_size: 200 embedding_dim: 4
             separator: '|' }
  features { input_names: 'cates' feature_type: SequenceFeature sub_feature_type: IdFeature num_buckets: 30 embedding_dim: 4
             separator: ';' }
  features { input_names: 'prices' feature_type: SequenceFeature sub_feature_type: RawFeature boundaries: [1, 5, 10] embedding_dim: 4
             separator: '|' }
  features { feature_name: 'city_value' input_names: 'city' input_names: 'kv_map' feature_type: LookupFeature hash_bucket_size: 300
             embedding_dim: 4 separator: '|' kv_separator: ':' lookup_max_sel_elem_num: 4 }
  features { feature_name: 'uid_x_level' input_names: 'uid' input_names: 'level' feature_type: ComboFeature hash_bucket_size: 400
             embedding_dim: 4 }
  features { feature_name: 'city_level' input_names: 'city' input_names: 'level' feature_type: ComboFeature hash_bucket_size: 400
             embedding_dim: 4 combo_join_sep: '_' }
"""

COLUMNS = {
    'label': [1, 0, 0, 1, 0, 1],
    'uid': ['u1', 'user two', '', 'u4', 'u1', '用户'],
    'item': [7, 0, 123456789012, 42, 7, 5],
    'city': ['3', '19', '0', '7', '3', '11'],
    'level': [0, 7, 3, 3, 1, 2],
    'price': [1.0, 101.0, 51.0, 26.5, 11.0, 76.0],
    'ctr': ['0.5', '1e-3', '0', '3', '-2.25', '7.125'],
    'vec': ['1,2,3', '0.5,0.25,0.125', '1,2', '-1,-2,-3', '4', '9,8,7'],
    'age': [17.0, 18.0, 29.5, 30.0, 44.0, 60.0],
    'tags': ['a|b;c', '', 'x', '||y||', 'a;a', 'p|q|r|s'],
    'tags_kv': ['a:0.5|b:2', 'c:1', 'd:0.25|e:0|f:3.5', 'g:1', 'h:2|i:4', 'j:1'],
    'tag_ids': ['1,2,3', '49', '0,0', '5', '7,8', '10,11,12,13'],
    'tag_w': ['0.1,0.2,0.3', '1', '2,3', '0.5', '1.5,2.5', '1,1,1,1'],
    'clicks': ['c1|c2|c3', 'c9', 'c1||c2', '', 'c5|c6', 'c7|c8|c9|c1'],
    'cates': ['1;2;3', '29', '0;0', '5', '7;8', '10;11;12;13'],
    'prices': ['0.5|3|12', '5', '10|1', '7.5', '0|100', '4.9|5.1'],
    'kv_map': ['3:a|4:b|3:c', '19:x', '1:no', '7:p|7:q|7:r|8:s', '', '11:z|12:y'],
}

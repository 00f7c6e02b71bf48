#!/usr/bin/env python
"""Write the in-repo pipeline configs used by bench.py / smoke() / the -m gpu tests.

The GPU box has no /root/reference, so the workload configs must live in this repo.  They are built
programmatically as `EasyRecConfig` messages and serialised with protobuf's text_format (a test
checks that `configs/deepfm_criteo.config` parses to the same message as the reference's
examples/configs/deepfm_on_criteo.config when the reference tree is present).

  python tools/make_configs.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from google.protobuf import text_format  # noqa: E402

from easyrec_amd.protos import pipeline_pb2  # noqa: E402
from easyrec_amd.protos.dataset_pb2 import DatasetConfig  # noqa: E402
from easyrec_amd.protos.feature_config_pb2 import FeatureConfig, WideOrDeep  # noqa: E402

# Criteo numeric features: [min_val, max_val] used for min/max normalisation
CRITEO_NUM_RANGE = [(0.0, 5775.0), (-3.0, 257675.0), (0.0, 65535.0), (0.0, 969.0), (0.0, 23159456.0),
                    (0.0, 431037.0), (0.0, 56311.0), (0.0, 6047.0), (0.0, 29019.0), (0.0, 46.0), (0.0, 231.0),
                    (0.0, 4008.0), (0.0, 7393.0)]


def criteo_base(optimizer='adam_optimizer', hash_bucket_size=1000000, embedding_dim=16, batch_size=4096):
  cfg = pipeline_pb2.EasyRecConfig()
  cfg.train_input_path = 'examples/data/criteo/criteo_train_data'
  cfg.eval_input_path = 'examples/data/criteo/criteo_test_data'
  cfg.model_dir = 'examples/ckpt/deepfm_criteo_ckpt'
  tc = cfg.train_config
  tc.log_step_count_steps = 500
  oc = tc.optimizer_config.add()
  opt = getattr(oc, optimizer)
  lr = opt.learning_rate.exponential_decay_learning_rate
  lr.initial_learning_rate = 0.001
  lr.decay_steps = 1000
  lr.decay_factor = 0.5
  lr.min_learning_rate = 0.00001
  oc.use_moving_average = False
  tc.save_checkpoints_steps = 1000
  tc.sync_replicas = True
  tc.num_steps = 20000
  cfg.eval_config.metrics_set.add().auc.SetInParent()
  dc = cfg.data_config
  dc.separator = '\t'
  f = dc.input_fields.add()
  f.input_name, f.input_type, f.default_val = 'label', DatasetConfig.FLOAT, '0'
  for i in range(1, 14):
    f = dc.input_fields.add()
    f.input_name, f.input_type, f.default_val = 'F%d' % i, DatasetConfig.FLOAT, '0'
  for i in range(1, 27):
    f = dc.input_fields.add()
    f.input_name, f.input_type, f.default_val = 'C%d' % i, DatasetConfig.STRING, ''
  dc.label_fields.append('label')
  dc.batch_size = batch_size
  dc.num_epochs = 1
  dc.prefetch_size = 32
  dc.input_type = DatasetConfig.CSVInput
  for i, (lo, hi) in enumerate(CRITEO_NUM_RANGE):
    fc = cfg.feature_config.features.add()
    fc.input_names.append('F%d' % (i + 1))
    fc.embedding_dim = embedding_dim
    fc.feature_type = FeatureConfig.RawFeature
    fc.min_val, fc.max_val = lo, hi
  for i in range(1, 27):
    fc = cfg.feature_config.features.add()
    fc.input_names.append('C%d' % i)
    fc.hash_bucket_size = hash_bucket_size
    fc.feature_type = FeatureConfig.IdFeature
    fc.embedding_dim = embedding_dim
  return cfg


def _all_names():
  return ['F%d' % i for i in range(1, 14)] + ['C%d' % i for i in range(1, 27)]


def deepfm_criteo(**kw):
  cfg = criteo_base(**kw)
  mc = cfg.model_config
  mc.model_class = 'DeepFM'
  for gname, wd in (('deep', WideOrDeep.DEEP), ('wide', WideOrDeep.WIDE)):
    g = mc.feature_groups.add()
    g.group_name = gname
    g.feature_names.extend(_all_names())
    g.wide_deep = wd
  mc.deepfm.dnn.hidden_units.extend([256, 128, 64])
  mc.deepfm.final_dnn.hidden_units.extend([256, 128, 64])
  mc.deepfm.wide_regularization = 1e-4
  mc.deepfm.dense_regularization = 1e-5
  mc.embedding_regularization = 1e-5
  return cfg


def dcn_criteo(**kw):
  """DCN-v1 (model_class DCN) on the same 39 features, one group `all` (BASELINE config 3 shape)."""
  cfg = criteo_base(**kw)
  cfg.model_dir = 'examples/ckpt/dcn_criteo_ckpt'
  mc = cfg.model_config
  mc.model_class = 'DCN'
  g = mc.feature_groups.add()
  g.group_name = 'all'
  g.feature_names.extend(_all_names())
  g.wide_deep = WideOrDeep.DEEP
  mc.dcn.deep_tower.input = 'all'
  mc.dcn.deep_tower.dnn.hidden_units.extend([256, 128, 64])
  mc.dcn.cross_tower.input = 'all'
  mc.dcn.cross_tower.cross_num = 3
  mc.dcn.final_dnn.hidden_units.extend([64, 32, 16])
  mc.dcn.l2_regularization = 1e-6
  mc.embedding_regularization = 1e-5
  return cfg


def dcn_v2_criteo(cross_steps=3, projection_dim=0, **kw):
  """DCN-v2 through the backbone API (model_class RankModel): deep MLP || `Cross` x cross_steps (recurrent,
  x0 fixed), concat, top_mlp - the block structure of samples/model_config/dcn_backbone_on_taobao.config on
  the 39 Criteo features (BASELINE config 3)."""
  cfg = criteo_base(**kw)
  cfg.model_dir = 'examples/ckpt/dcn_v2_criteo_ckpt'
  mc = cfg.model_config
  mc.model_class = 'RankModel'
  g = mc.feature_groups.add()
  g.group_name = 'all'
  g.feature_names.extend(_all_names())
  g.wide_deep = WideOrDeep.DEEP
  bb = mc.backbone
  b = bb.blocks.add()
  b.name = 'deep'
  b.inputs.add().feature_group_name = 'all'
  b.keras_layer.class_name = 'MLP'
  b.keras_layer.mlp.hidden_units.extend([256, 128, 64])
  b = bb.blocks.add()
  b.name = 'cross'
  i = b.inputs.add()
  i.feature_group_name = 'all'
  i.input_fn = 'lambda x: [x, x]'
  b.recurrent.num_steps = cross_steps
  b.recurrent.fixed_input_index = 0
  b.recurrent.keras_layer.class_name = 'Cross'
  if projection_dim:
    b.recurrent.keras_layer.st_params['projection_dim'] = projection_dim
  bb.concat_blocks.extend(['deep', 'cross'])
  bb.top_mlp.hidden_units.extend([64, 32, 16])
  mc.model_params.l2_regularization = 1e-6
  mc.embedding_regularization = 1e-5
  return cfg


# ---------------------------------------------------------------------------------------------
# Taobao-shaped configs (the feature set of the reference's samples/model_config/*_on_taobao.config)
# ---------------------------------------------------------------------------------------------
TAOBAO_ID_FEATURES = [  # (name, hash_bucket_size)
    ('pid', 10), ('adgroup_id', 100000), ('cate_id', 10000), ('campaign_id', 100000), ('customer', 100000),
    ('brand', 100000), ('user_id', 100000), ('cms_segid', 100), ('cms_group_id', 100), ('final_gender_code', 10),
    ('age_level', 10), ('pvalue_level', 10), ('shopping_level', 10), ('occupation', 10),
    ('new_user_class_level', 10)]
TAOBAO_USER = ['user_id', 'cms_segid', 'cms_group_id', 'age_level', 'pvalue_level', 'shopping_level', 'occupation',
               'new_user_class_level']
TAOBAO_ITEM = ['adgroup_id', 'cate_id', 'campaign_id', 'customer', 'brand', 'price', 'pid']


def taobao_base(list_type, batch_size=4096, scale=1.0, embedding_dim=16, seq_len=50, item_rows=None):
  """list_type: 'tag' (TagFeature multi-hot lists, MMoE/DCN samples) or 'seq' (SequenceFeature, DIN sample)."""
  cfg = pipeline_pb2.EasyRecConfig()
  cfg.train_input_path = 'data/test/tb_data/taobao_train_data'
  cfg.eval_input_path = 'data/test/tb_data/taobao_test_data'
  tc = cfg.train_config
  tc.log_step_count_steps = 100
  oc = tc.optimizer_config.add()
  lr = oc.adam_optimizer.learning_rate.exponential_decay_learning_rate
  lr.initial_learning_rate, lr.decay_steps, lr.decay_factor, lr.min_learning_rate = 0.001, 1000, 0.5, 0.00001
  oc.use_moving_average = False
  tc.save_checkpoints_steps = 100
  tc.sync_replicas = True
  tc.num_steps = 2500
  cfg.eval_config.metrics_set.add().auc.SetInParent()
  dc = cfg.data_config
  for name in ('clk', 'buy'):
    f = dc.input_fields.add()
    f.input_name, f.input_type = name, DatasetConfig.INT32
  for name, _ in TAOBAO_ID_FEATURES:
    f = dc.input_fields.add()
    f.input_name, f.input_type = name, DatasetConfig.STRING
  for name in ('tag_category_list', 'tag_brand_list'):
    f = dc.input_fields.add()
    f.input_name, f.input_type = name, DatasetConfig.STRING
  f = dc.input_fields.add()
  f.input_name, f.input_type = 'price', DatasetConfig.INT32
  dc.batch_size = batch_size
  dc.num_epochs = 10000
  dc.prefetch_size = 32
  dc.input_type = DatasetConfig.CSVInput
  for name, buckets in TAOBAO_ID_FEATURES:
    fc = cfg.feature_config.features.add()
    fc.input_names.append(name)
    fc.feature_type = FeatureConfig.IdFeature
    fc.embedding_dim = embedding_dim
    b = max(int(buckets * scale), 4) if buckets > 100 else buckets
    if item_rows and name == 'adgroup_id':
      b = item_rows
    fc.hash_bucket_size = b
  for name in ('tag_category_list', 'tag_brand_list'):
    fc = cfg.feature_config.features.add()
    fc.input_names.append(name)
    fc.embedding_dim = embedding_dim
    fc.hash_bucket_size = max(int(100000 * scale), 4)
    fc.separator = '|'
    if list_type == 'tag':
      fc.feature_type = FeatureConfig.TagFeature
    else:
      fc.feature_type = FeatureConfig.SequenceFeature
      fc.max_seq_len = seq_len
  fc = cfg.feature_config.features.add()
  fc.input_names.append('price')
  fc.feature_type = FeatureConfig.IdFeature
  fc.embedding_dim = embedding_dim
  fc.num_buckets = 50
  return cfg


def din_taobao(**kw):
  """MultiTowerDIN, the shape of samples/model_config/din_on_taobao.config (BASELINE config 4)."""
  cfg = taobao_base('seq', **kw)
  cfg.model_dir = 'experiments/din_taobao_ckpt'
  cfg.data_config.label_fields.append('clk')
  mc = cfg.model_config
  mc.model_class = 'MultiTowerDIN'
  for gname, names in (('user', TAOBAO_USER), ('item', TAOBAO_ITEM)):
    g = mc.feature_groups.add()
    g.group_name = gname
    g.feature_names.extend(names)
    g.wide_deep = WideOrDeep.DEEP
  sg = mc.seq_att_groups.add()
  sg.group_name = 'din'
  for key, hist in (('brand', 'tag_brand_list'), ('cate_id', 'tag_category_list')):
    m = sg.seq_att_map.add()
    m.key.append(key)
    m.hist_seq.append(hist)
  mt = mc.multi_tower
  for gname in ('user', 'item'):
    t = mt.towers.add()
    t.input = gname
    t.dnn.hidden_units.extend([256, 128, 96, 64])
  dt = mt.din_towers.add()
  dt.input = 'din'
  dt.dnn.hidden_units.extend([128, 64, 32, 1])
  mt.final_dnn.hidden_units.extend([128, 96, 64, 32, 16])
  mt.l2_regularization = 5e-7
  mc.embedding_regularization = 5e-5
  return cfg


def mmoe_taobao(n_tasks=2, **kw):
  """MMoE, the shape of samples/model_config/mmoe_on_taobao.config (BASELINE config 5 uses 4 tasks)."""
  cfg = taobao_base('tag', **kw)
  cfg.model_dir = 'experiments/mmoe_taobao_ckpt'
  cfg.data_config.label_fields.extend(['clk', 'buy'])
  mc = cfg.model_config
  mc.model_class = 'MMoE'
  g = mc.feature_groups.add()
  g.group_name = 'all'
  g.feature_names.extend(TAOBAO_USER + ['adgroup_id', 'cate_id', 'campaign_id', 'customer', 'brand', 'price', 'pid',
                                        'tag_category_list', 'tag_brand_list'])
  g.wide_deep = WideOrDeep.DEEP
  mm = mc.mmoe
  mm.expert_dnn.hidden_units.extend([256, 192, 128, 64])
  mm.num_expert = 4
  towers = [('ctr', 'clk'), ('cvr', 'buy'), ('ctr2', 'clk'), ('cvr2', 'buy')][:n_tasks]
  for tname, label in towers:
    t = mm.task_towers.add()
    t.tower_name, t.label_name = tname, label
    t.dnn.hidden_units.extend([256, 192, 128, 64])
    t.num_class = 1
    t.weight = 1.0
    t.metrics_set.add().auc.SetInParent()
  mm.l2_regularization = 1e-6
  mc.embedding_regularization = 5e-5
  return cfg


def _model_text(cfg, text):
  cfg.ClearField('model_config')
  text_format.Merge(text, cfg.model_config)
  return cfg


def din_backbone_taobao(**kw):
  """The shape of samples/model_config/din_backbone_on_taobao.config: RankModel over a backbone whose `seq_input`
  block hands (history [B, L, E], lengths, target features) to the keras `DIN` layer; MLP tower beside it."""
  cfg = taobao_base('seq', **kw)
  cfg.model_dir = 'experiments/din_backbone_taobao_ckpt'
  cfg.data_config.label_fields.append('clk')
  normal = TAOBAO_USER + ['adgroup_id', 'cate_id', 'campaign_id', 'customer', 'brand', 'price', 'pid']
  return _model_text(cfg, '''
    model_name: 'DIN'  model_class: 'RankModel'
    feature_groups { group_name: 'normal' %s wide_deep: DEEP }
    feature_groups { group_name: 'sequence' feature_names: 'cate_id' feature_names: 'brand'
                     feature_names: 'tag_category_list' feature_names: 'tag_brand_list' wide_deep: DEEP }
    backbone {
      blocks { name: 'deep' inputs { feature_group_name: 'normal' }
               keras_layer { class_name: 'MLP' mlp { hidden_units: [256, 128, 64] } } }
      blocks { name: 'seq_input' inputs { feature_group_name: 'sequence' }
               input_layer { output_seq_and_normal_feature: true } }
      blocks { name: 'DIN' inputs { block_name: 'seq_input' }
               keras_layer { class_name: 'DIN'
                             din { attention_dnn { hidden_units: 32 hidden_units: 1 activation: "dice" }
                                   need_target_feature: true } } }
      top_mlp { hidden_units: [256, 128, 64] }
    }
    model_params { l2_regularization: 0 }
    embedding_regularization: 0
  ''' % ' '.join("feature_names: '%s'" % n for n in normal))


def din_sequence_features_taobao(**kw):
  """MultiTower whose `item` group carries `sequence_features` (target attention inside the input layer,
  layers/input_layer.py:96-111): keys `brand` / `cate_id` are the group's own outputs (allow_key_search false)."""
  cfg = taobao_base('seq', **kw)
  cfg.model_dir = 'experiments/din_sequence_features_taobao_ckpt'
  cfg.data_config.label_fields.append('clk')
  return _model_text(cfg, '''
    model_class: 'MultiTower'
    feature_groups { group_name: 'user' %s wide_deep: DEEP }
    feature_groups { group_name: 'item' %s wide_deep: DEEP
      sequence_features { group_name: 'seq_fea' allow_key_search: false need_key_feature: true
        seq_att_map { key: 'brand' hist_seq: 'tag_brand_list' }
        seq_att_map { key: 'cate_id' hist_seq: 'tag_category_list' }
        seq_dnn { hidden_units: [32, 16, 1] } } }
    multi_tower {
      towers { input: 'user' dnn { hidden_units: [64, 32] } }
      towers { input: 'item' dnn { hidden_units: [64, 32] } }
      final_dnn { hidden_units: [32, 16] }
      l2_regularization: 1e-6
    }
    embedding_regularization: 5e-5
  ''' % (' '.join("feature_names: '%s'" % n for n in TAOBAO_USER), ' '.join("feature_names: '%s'" % n for n in TAOBAO_ITEM)))


def dbmtl_numeric_sequences_taobao(transform_dnn=False, **kw):
  """The shape of the reference's samples/model_config/dbmtl_on_multi_numeric_boundary_allow_key_transform(_dnn).config:
  DBMTL whose `all` group carries `sequence_features` over two sequences of NUMBERS bucketized by `boundaries`
  (SequenceFeature, sub_feature_type RawFeature), attended by one key narrower than the two histories together
  (allow_key_transform: zero-padded, or with transform_dnn a dense layer on key and history)."""
  cfg = taobao_base('seq', **kw)
  cfg.model_dir = 'experiments/dbmtl_numeric_sequences_taobao_ckpt'
  cfg.data_config.label_fields.extend(['clk', 'buy'])
  for fc in cfg.feature_config.features:
    if fc.feature_type == FeatureConfig.SequenceFeature:
      fc.sub_feature_type = FeatureConfig.RawFeature
      fc.ClearField('hash_bucket_size')
      fc.boundaries.extend([15.0, 20.0, 21.0, 23.0, 30.0, 32.0, 40.0, 47.0, 66.0, 70.0])
  names = [n for n, _ in TAOBAO_ID_FEATURES] + ['price']
  return _model_text(cfg, '''
    model_class: 'DBMTL'
    feature_groups { group_name: 'all' %s wide_deep: DEEP
      sequence_features { group_name: 'seq_fea' allow_key_transform: true %s
        seq_att_map { key: 'brand' hist_seq: 'tag_brand_list' hist_seq: 'tag_category_list' } } }
    dbmtl {
      bottom_dnn { hidden_units: [64, 32] }
      task_towers { tower_name: 'ctr' label_name: 'clk' loss_type: CLASSIFICATION metrics_set { auc {} }
                    dnn { hidden_units: [32, 16] } relation_dnn { hidden_units: [8] } weight: 1.0 }
      task_towers { tower_name: 'cvr' label_name: 'buy' loss_type: CLASSIFICATION metrics_set { auc {} }
                    dnn { hidden_units: [32, 16] } relation_tower_names: 'ctr' relation_dnn { hidden_units: [8] } weight: 1.0 }
      l2_regularization: 1e-6
    }
    embedding_regularization: 5e-6
  ''' % (' '.join("feature_names: '%s'" % n for n in names), 'transform_dnn: true' if transform_dnn else ''))


def xdeepfm_backbone_taobao(hidden=(64, 64, 64), mlp=(128, 64), final=(32, 1), **kw):
  """The shape of samples/model_config/xdeepfm_on_taobao_backbone.config: RankModel over a backbone with a wide block
  (feature list of width-1 embeddings, summed by `tf.add_n`), a CIN block over the stacked field embeddings
  (`input_slice: '[1]'` + `extra_input_fn: tf.stack(x, axis=1)`), an MLP over the concatenated embeddings, and a final
  MLP over [wide, cin, dnn]."""
  cfg = taobao_base('tag', **kw)
  cfg.model_dir = 'experiments/xdeepfm_taobao_ckpt'
  cfg.data_config.label_fields.append('clk')
  feats = TAOBAO_USER + ['tag_category_list', 'tag_brand_list'] + TAOBAO_ITEM
  names = ' '.join("feature_names: '%s'" % n for n in feats)
  return _model_text(cfg, '''
    model_name: 'xDeepFM'  model_class: 'RankModel'
    feature_groups { group_name: 'features' %s wide_deep: DEEP }
    feature_groups { group_name: 'wide' %s wide_deep: WIDE }
    backbone {
      blocks { name: 'wide' inputs { feature_group_name: 'wide' }
               input_layer { only_output_feature_list: true wide_output_dim: 1 } }
      blocks { name: 'features' inputs { feature_group_name: 'features' }
               input_layer { output_2d_tensor_and_feature_list: true } }
      blocks { name: 'cin' inputs { block_name: 'features' input_slice: '[1]' }
               extra_input_fn: 'lambda x: tf.stack(x, axis=1)'
               keras_layer { class_name: 'CIN' cin { hidden_feature_sizes: %s } } }
      blocks { name: 'dnn' inputs { block_name: 'features' input_slice: '[0]' }
               keras_layer { class_name: 'MLP' mlp { hidden_units: %s } } }
      blocks { name: 'final_logit'
               inputs { block_name: 'wide' input_fn: 'lambda x: tf.add_n(x)' }
               inputs { block_name: 'cin' }
               inputs { block_name: 'dnn' }
               keras_layer { class_name: 'MLP'
                             mlp { hidden_units: %s use_final_bn: false final_activation: 'linear' } } }
      concat_blocks: 'final_logit'
    }
    model_params { l2_regularization: 1e-4 }
    embedding_regularization: 1e-4
  ''' % (names, names, list(hidden), list(mlp), list(final)))


def deepfm_backbone_criteo(**kw):
  """The shape of examples/configs/deepfm_backbone_on_criteo.config: the wide group's width comes from the backbone's
  `input_layer { wide_output_dim }`, wide logit through a config lambda, keras FM + MLP, top_mlp."""
  cfg = deepfm_criteo(**kw)
  names = ' '.join("feature_names: '%s'" % n for n in _all_names())
  return _model_text(cfg, '''
    model_name: 'DeepFM'  model_class: 'RankModel'
    feature_groups { group_name: 'deep_features' %s wide_deep: DEEP }
    feature_groups { group_name: 'wide_features' %s wide_deep: WIDE }
    backbone {
      blocks { name: 'wide_features' inputs { feature_group_name: 'wide_features' } input_layer { wide_output_dim: 1 } }
      blocks { name: 'wide_logit' inputs { block_name: 'wide_features' }
               lambda { expression: 'lambda x: tf.reduce_sum(x, axis=1, keepdims=True)' } }
      blocks { name: 'deep_features' inputs { feature_group_name: 'deep_features' }
               input_layer { output_2d_tensor_and_feature_list: true } }
      blocks { name: 'fm' inputs { block_name: 'deep_features' input_slice: '[1]' }
               keras_layer { class_name: 'FM' st_params { fields { key: 'use_variant' value { bool_value: true } } } } }
      blocks { name: 'deep' inputs { block_name: 'deep_features' input_slice: '[0]' }
               keras_layer { class_name: 'MLP' mlp { hidden_units: [256, 128, 64] } } }
      concat_blocks: ['wide_logit', 'fm', 'deep']
      top_mlp { hidden_units: [256, 128, 64] }
    }
    model_params { l2_regularization: 1e-5 }
    embedding_regularization: 1e-5
  ''' % (names, names))


def dlrm_backbone_criteo(bottom=(64, 32, 16), top=(256, 128, 64), **kw):
  """The shape of examples/configs/dlrm_backbone_on_criteo.config: bottom MLP over the dense group, the keras
  DotInteraction over [bottom output] + the sparse group's feature list (inputs merged into one list), concatenated with
  the sparse embeddings, top_mlp."""
  cfg = deepfm_criteo(**kw)
  names = _all_names()
  dense = ' '.join("feature_names: '%s'" % n for n in names[:13])
  sparse = ' '.join("feature_names: '%s'" % n for n in names[13:])
  return _model_text(cfg, '''
    model_name: 'DLRM'  model_class: 'RankModel'
    feature_groups { group_name: 'dense' %s wide_deep: DEEP }
    feature_groups { group_name: 'sparse' %s wide_deep: DEEP }
    backbone {
      blocks { name: 'bottom_mlp' inputs { feature_group_name: 'dense' }
               keras_layer { class_name: 'MLP' mlp { hidden_units: %s } } }
      blocks { name: 'sparse' inputs { feature_group_name: 'sparse' }
               input_layer { output_2d_tensor_and_feature_list: true } }
      blocks { name: 'dot'
               inputs { block_name: 'bottom_mlp' input_fn: 'lambda x: [x]' }
               inputs { block_name: 'sparse' input_fn: 'lambda x: x[1]' }
               keras_layer { class_name: 'DotInteraction' } }
      blocks { name: 'sparse_2d' inputs { block_name: 'sparse' input_fn: 'lambda x: x[0]' } }
      concat_blocks: ['sparse_2d', 'dot']
      top_mlp { hidden_units: %s }
    }
    model_params { l2_regularization: 1e-5 }
    embedding_regularization: 1e-5
  ''' % (dense, sparse, list(bottom), list(top)))


def wide_and_deep_backbone_criteo(hidden=(256, 256, 256, 1), **kw):
  """The shape of examples/configs/wide_and_deep_backbone_on_movielens.config: the wide feature list summed by
  `tf.add_n`, a deep MLP down to one logit (no final BatchNorm, linear), the two merged into a list and summed by the
  standard keras `Add` layer."""
  cfg = deepfm_criteo(**kw)
  names = ' '.join("feature_names: '%s'" % n for n in _all_names())
  return _model_text(cfg, '''
    model_name: 'WideAndDeep'  model_class: 'RankModel'
    feature_groups { group_name: 'wide' %s wide_deep: WIDE }
    feature_groups { group_name: 'deep' %s wide_deep: DEEP }
    backbone {
      blocks { name: 'wide' inputs { feature_group_name: 'wide' }
               input_layer { wide_output_dim: 1 only_output_feature_list: true } }
      blocks { name: 'deep_logit' inputs { feature_group_name: 'deep' }
               keras_layer { class_name: 'MLP' mlp { hidden_units: %s use_final_bn: false final_activation: 'linear' } } }
      blocks { name: 'final_logit'
               inputs { block_name: 'wide' input_fn: 'lambda x: tf.add_n(x)' }
               inputs { block_name: 'deep_logit' }
               merge_inputs_into_list: true
               keras_layer { class_name: 'Add' } }
      concat_blocks: 'final_logit'
    }
    model_params { l2_regularization: 1e-4 }
    embedding_regularization: 1e-4
  ''' % (names, names, list(hidden)))


def write(cfg, name):
  out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs', name)
  with open(out, 'w') as f:
    f.write('# generated by tools/make_configs.py - do not edit\n')
    f.write(text_format.MessageToString(cfg, as_utf8=True))
  print('wrote', out)


def shared_embedding_variant(src_name, dst_name, embedding_name='embedding'):
  """The reference's own embedding-parallel Criteo config keeps ONE table for all categorical features
  (samples/model_config/dlrm_on_criteo_parquet_ep_v2.config: `embedding_name: "embedding"` on all 26): the same edit on
  one of the small fixtures."""
  from easyrec_amd.protos import pipeline_pb2
  from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  for fc in cfg.feature_config.features:
    if fc.feature_type == FeatureConfig.IdFeature and fc.hash_bucket_size > 0:
      fc.embedding_name = embedding_name
  write(cfg, dst_name)


def combo_feature_variant(src_name, dst_name):
  """+ one ComboFeature (crossed_column of two categorical inputs, reference feature_column.py:434-445) in every
  feature group that holds both inputs."""
  from easyrec_amd.protos import pipeline_pb2
  from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  fc = cfg.feature_config.features.add()
  fc.input_names.extend(['C1', 'C2'])
  fc.feature_name = 'C1_C2_cross'
  fc.feature_type = FeatureConfig.ComboFeature
  fc.hash_bucket_size = 1000
  fc.embedding_dim = 16
  for g in cfg.model_config.feature_groups:
    names = list(g.feature_names)
    if 'C1' in names or any(n.startswith('C[') for n in names):
      g.feature_names.append('C1_C2_cross')
  write(cfg, dst_name)


def lookup_feature_variant(src_name, dst_name):
  """+ one LookupFeature (reference input/input.py:941-1000, feature_column.py:457-476): key = C3, map = a new
  string field 'KV' of 'key:value' pairs joined by '|'."""
  from easyrec_amd.protos import pipeline_pb2
  from easyrec_amd.protos.dataset_pb2 import DatasetConfig
  from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  fld = cfg.data_config.input_fields.add()
  fld.input_name = 'KV'
  fld.input_type = DatasetConfig.STRING
  fc = cfg.feature_config.features.add()
  fc.input_names.extend(['C3', 'KV'])
  fc.feature_name = 'C3_lookup'
  fc.feature_type = FeatureConfig.LookupFeature
  fc.hash_bucket_size = 500
  fc.embedding_dim = 16
  fc.separator = '|'
  fc.kv_separator = ':'
  fc.lookup_max_sel_elem_num = 4
  fc.combiner = 'mean'
  for g in cfg.model_config.feature_groups:
    if 'C3' in list(g.feature_names):
      g.feature_names.append('C3_lookup')
  write(cfg, dst_name)


def simple_multi_task_variant(src_name, dst_name):
  """The MMoE fixture with `simple_multi_task { task_towers ... }` instead of the expert / gate layer
  (reference model/simple_multi_task.py)."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  mm = cfg.model_config.mmoe
  towers = [t for t in mm.task_towers]
  l2 = mm.l2_regularization
  cfg.model_config.ClearField('mmoe')
  cfg.model_config.model_class = 'SimpleMultiTask'
  smt = cfg.model_config.simple_multi_task
  smt.l2_regularization = l2
  for t in towers:
    smt.task_towers.add().CopyFrom(t)
  write(cfg, dst_name)


def mmoe_backbone_variant(src_name, dst_name, senet=True, bayes=False):
  """The MMoE fixture in the shape of the reference's samples/model_config/mmoe_backbone_on_taobao.config:
  `model_class: "MultiTaskModel"` over a backbone of the feature list -> SENet -> keras MMoE (3 expert MLPs), the task
  towers in model_params.  bayes: cvr's relation network also reads ctr's relation features."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  mc = cfg.model_config
  towers = [t for t in mc.mmoe.task_towers]
  l2 = mc.mmoe.l2_regularization
  mc.ClearField('mmoe')
  mc.model_class = 'MultiTaskModel'
  mc.model_name = 'MMoE'
  text_format.Merge("""
    blocks { name: 'all' inputs { feature_group_name: 'all' } input_layer { only_output_feature_list: true } }
    %s
    blocks { name: 'mmoe' inputs { block_name: '%s' }
             keras_layer { class_name: 'MMoE' mmoe { num_task: %d num_expert: 3 expert_mlp { hidden_units: [64, 32] } } } }
  """ % ("blocks { name: 'senet' inputs { block_name: 'all' } keras_layer { class_name: 'SENet' senet { reduction_ratio: 4 } } }"
         if senet else "blocks { name: 'flat' inputs { block_name: 'all' input_fn: 'lambda x: tf.concat(x, axis=-1)' } }",
         'senet' if senet else 'flat', len(towers)), mc.backbone)
  mc.model_params.l2_regularization = l2
  for i, t in enumerate(towers):
    bt = mc.model_params.task_towers.add()
    bt.tower_name, bt.label_name, bt.num_class, bt.weight = t.tower_name, t.label_name, t.num_class, t.weight
    for m in t.metrics_set:
      bt.metrics_set.add().CopyFrom(m)
    bt.dnn.CopyFrom(t.dnn)
    if bayes:
      bt.relation_dnn.hidden_units.extend([16])
      if i > 0:
        bt.relation_tower_names.append(towers[0].tower_name)
  write(cfg, dst_name)


def losses_variant(src_name, dst_name):
  """A single-task fixture with the `losses` list of the reference's samples/model_config/multi_tower_on_taobao.config:
  F1_REWEIGHTED_LOSS (f1_beta_square 2.25) + PAIR_WISE_LOSS, each with weight 1."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  text_format.Merge("""
    losses { loss_type: F1_REWEIGHTED_LOSS weight: 1.0 f1_reweighted_loss { f1_beta_square: 2.25 } }
    losses { loss_type: PAIR_WISE_LOSS weight: 1.0 }
  """, cfg.model_config)
  write(cfg, dst_name)


def tower_losses_variant(src_name, dst_name):
  """The MMoE fixture with a `losses` list on its first tower (the shape of the reference's
  samples/model_config/mmoe_on_taobao_with_multi_loss.config; the second weight differs from the first on purpose:
  the reference multiplies every loss of the list by the FIRST weight, model/multi_task_model.py:263-269)."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  text_format.Merge("""
    losses { loss_type: CLASSIFICATION weight: 0.8 }
    losses { loss_type: PAIR_WISE_LOSS weight: 0.25 loss_name: 'rank' pairwise_loss { margin: 0.1 temperature: 2.0 } }
  """, cfg.model_config.mmoe.task_towers[0])
  write(cfg, dst_name)


def adagrad_embedding_variant(src_name, dst_name):
  """Two optimizers, as the reference's samples/model_config/deepfm_combo_on_avazu_embed_adagrad.config: Adagrad for
  the embedding tables (initial accumulator 0.2), the fixture's Adam for the dense variables."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  dense = pipeline_pb2.EasyRecConfig().train_config.optimizer_config.add()
  dense.CopyFrom(cfg.train_config.optimizer_config[0])
  emb = cfg.train_config.optimizer_config[0]
  emb.Clear()
  text_format.Merge("adagrad_optimizer { learning_rate { constant_learning_rate { learning_rate: 0.05 } } "
                    "initial_accumulator_value: 0.2 }", emb)
  cfg.train_config.optimizer_config.add().CopyFrom(dense)
  write(cfg, dst_name)


def ple_variant(src_name, dst_name):
  """The MMoE fixture as PLE (reference model/ple.py): two extraction networks, 2 experts per task + 2 shared."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  mm = cfg.model_config.mmoe
  towers = [t for t in mm.task_towers]
  l2 = mm.l2_regularization
  cfg.model_config.ClearField('mmoe')
  cfg.model_config.model_class = 'PLE'
  ple = cfg.model_config.ple
  ple.l2_regularization = l2
  for i, units in enumerate(([64, 32], [32, 16])):
    net = ple.extraction_networks.add()
    net.network_name = 'network_%d' % i
    net.expert_num_per_task = 2
    net.share_num = 2
    net.task_expert_net.hidden_units.extend(units)
    net.share_expert_net.hidden_units.extend(units)
  for t in towers:
    ple.task_towers.add().CopyFrom(t)
  write(cfg, dst_name)


def dbmtl_variant(src_name, dst_name, experts=0):
  """The MMoE fixture as DBMTL (reference model/dbmtl.py): bottom_dnn, per-task towers, cvr's relation network also
  reads ctr's relation features.  experts > 0: + the MMoE block between the bottom and the towers (the shape of the
  reference's samples/model_config/dbmtl_mmoe_on_taobao.config)."""
  from easyrec_amd.protos import pipeline_pb2
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs')
  cfg = pipeline_pb2.EasyRecConfig()
  with open(os.path.join(here, src_name)) as f:
    text_format.Merge(f.read(), cfg)
  mm = cfg.model_config.mmoe
  towers = [t for t in mm.task_towers]
  l2 = mm.l2_regularization
  cfg.model_config.ClearField('mmoe')
  cfg.model_config.model_class = 'DBMTL'
  db = cfg.model_config.dbmtl
  db.l2_regularization = l2
  db.bottom_dnn.hidden_units.extend([128, 64])
  if experts > 0:
    db.expert_dnn.hidden_units.extend([64, 32])
    db.num_expert = experts
  for i, t in enumerate(towers):
    bt = db.task_towers.add()
    bt.tower_name, bt.label_name, bt.num_class, bt.weight = t.tower_name, t.label_name, t.num_class, t.weight
    for m in t.metrics_set:
      bt.metrics_set.add().CopyFrom(m)
    bt.dnn.hidden_units.extend([64, 32])
    bt.relation_dnn.hidden_units.extend([32])
    if i > 0:
      bt.relation_tower_names.append(towers[0].tower_name)
  write(cfg, dst_name)


if __name__ == '__main__':
  write(deepfm_criteo(), 'deepfm_criteo.config')
  write(deepfm_criteo(optimizer='lazy_adam_optimizer'), 'deepfm_criteo_lazy_adam.config')
  write(deepfm_criteo(hash_bucket_size=1000, batch_size=256), 'deepfm_criteo_small.config')
  write(dcn_criteo(), 'dcn_criteo.config')
  write(dcn_criteo(hash_bucket_size=1000, batch_size=256), 'dcn_criteo_small.config')
  write(dcn_v2_criteo(), 'dcn_v2_criteo.config')
  write(dcn_v2_criteo(hash_bucket_size=1000, batch_size=256), 'dcn_v2_criteo_small.config')
  write(dcn_v2_criteo(hash_bucket_size=1000, batch_size=256, projection_dim=32, cross_steps=2),
        'dcn_v2_lowrank_criteo_small.config')
  write(din_taobao(), 'din_taobao.config')
  write(din_taobao(item_rows=10000000), 'din_taobao_10m.config')
  write(din_taobao(batch_size=128, scale=0.01, seq_len=12), 'din_taobao_small.config')
  write(mmoe_taobao(), 'mmoe_taobao.config')
  write(mmoe_taobao(n_tasks=4, embedding_dim=64, batch_size=8192), 'mmoe_taobao_4task_d64.config')
  # BASELINE config 5 at full size (200 M embedding rows of 64 floats: 51 GB + Adam slots, row-sharded over 8 GPUs) and
  # the share one GPU owns of it (25 M rows) for single-GPU runs
  write(mmoe_taobao(n_tasks=4, embedding_dim=64, batch_size=8192, item_rows=200000000), 'mmoe_taobao_4task_d64_200m.config')
  write(mmoe_taobao(n_tasks=4, embedding_dim=64, batch_size=8192, item_rows=25000000), 'mmoe_taobao_4task_d64_25m.config')
  write(mmoe_taobao(batch_size=128, scale=0.01), 'mmoe_taobao_small.config')
  write(din_backbone_taobao(), 'din_backbone_taobao.config')
  write(din_backbone_taobao(batch_size=128, scale=0.01, seq_len=12), 'din_backbone_taobao_small.config')
  write(din_sequence_features_taobao(batch_size=128, scale=0.01, seq_len=12), 'din_sequence_features_taobao_small.config')
  write(deepfm_backbone_criteo(hash_bucket_size=1000, batch_size=256), 'deepfm_backbone_criteo_small.config')
  write(xdeepfm_backbone_taobao(), 'xdeepfm_taobao.config')
  # hash-table (ev_params) embeddings: C1..C4 keyed by the full 63-bit hash, rows created on first sight
  cfg = deepfm_criteo(hash_bucket_size=1000, batch_size=256)
  for fc in cfg.feature_config.features:
    if fc.input_names[0] in ('C1', 'C2', 'C3', 'C4'):
      fc.ev_params.max_capacity = 4096
  write(cfg, 'deepfm_kv_criteo_small.config')
  cfg = mmoe_taobao(batch_size=128, scale=0.01)
  for fc in cfg.feature_config.features:
    if fc.input_names[0] in ('adgroup_id', 'user_id', 'tag_category_list', 'tag_brand_list'):
      fc.ev_params.max_capacity = 2048
  write(cfg, 'mmoe_kv_taobao_small.config')
  write(dlrm_backbone_criteo(bottom=(32, 16), top=(64, 32), hash_bucket_size=1000, batch_size=256), 'dlrm_backbone_criteo_small.config')
  write(wide_and_deep_backbone_criteo(hidden=(64, 32, 1), hash_bucket_size=1000, batch_size=256),
        'wide_and_deep_backbone_criteo_small.config')
  write(xdeepfm_backbone_taobao(hidden=(8, 6), mlp=(32, 16), final=(16, 1), batch_size=128, scale=0.01, embedding_dim=8),
        'xdeepfm_taobao_small.config')
  shared_embedding_variant('dlrm_criteo_small.config', 'dlrm_shared_criteo_small.config')
  shared_embedding_variant('deepfm_criteo_small.config', 'deepfm_shared_criteo_small.config')
  combo_feature_variant('deepfm_criteo_small.config', 'deepfm_combo_criteo_small.config')
  lookup_feature_variant('deepfm_criteo_small.config', 'deepfm_lookup_criteo_small.config')
  simple_multi_task_variant('mmoe_taobao_small.config', 'simple_multi_task_taobao_small.config')
  ple_variant('mmoe_taobao_small.config', 'ple_taobao_small.config')
  dbmtl_variant('mmoe_taobao_small.config', 'dbmtl_taobao_small.config')
  dbmtl_variant('mmoe_taobao_small.config', 'dbmtl_mmoe_taobao_small.config', experts=3)
  mmoe_backbone_variant('mmoe_taobao_small.config', 'mmoe_backbone_taobao_small.config')
  losses_variant('multi_tower_criteo_small.config', 'multi_tower_f1_pairwise_criteo_small.config')
  adagrad_embedding_variant('deepfm_criteo_small.config', 'deepfm_adagrad_criteo_small.config')
  tower_losses_variant('mmoe_taobao_small.config', 'mmoe_tower_losses_taobao_small.config')
  write(dbmtl_numeric_sequences_taobao(batch_size=128, scale=0.01, seq_len=12), 'dbmtl_numeric_sequences_taobao_small.config')
  write(dbmtl_numeric_sequences_taobao(transform_dnn=True, batch_size=128, scale=0.01, seq_len=12),
        'dbmtl_numeric_sequences_dnn_taobao_small.config')
  mmoe_backbone_variant('mmoe_taobao_small.config', 'mmoe_backbone_bayes_taobao_small.config', senet=False, bayes=True)

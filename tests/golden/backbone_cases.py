"""Backbone configs of the backbone pins: read by make_reference_layer_vectors.py (which runs the REFERENCE's
layers/backbone.py Backbone / Package - with its own EnhancedInputLayer, DAG, Parameter and keras layers - on them) and by
tests/test_reference_layers.py (which runs THE PRODUCT's layers/backbone.py on them).  tag -> (backbone text, groups).

Groups as in model_assembly_cases.py: ('cat', [widths]) a feature group's per-feature widths; ('seq', E, L, [target
widths]) a group with one sequence feature [B, L, E] (+ lengths) and target features, for input layers with
output_seq_and_normal_feature.  The shapes follow the fixtures of tools/make_configs.py / the reference's
examples/configs/*_backbone_*.config and samples/model_config/*backbone*.config.
"""
from collections import OrderedDict

B = 9

CASES = OrderedDict()

# DCN-v2 (BASELINE config 3): MLP beside a recurrent Cross with a fixed first input, concatenated, top_mlp
CASES['bb_dcn_v2'] = ("""
  blocks { name: 'deep' inputs { feature_group_name: 'all' } keras_layer { class_name: 'MLP' mlp { hidden_units: [8, 4] } } }
  blocks { name: 'cross' inputs { feature_group_name: 'all' input_fn: 'lambda x: [x, x]' }
           recurrent { num_steps: 3 fixed_input_index: 0 keras_layer { class_name: 'Cross' } } }
  concat_blocks: ['deep', 'cross']
  top_mlp { hidden_units: [6, 3] }
""", OrderedDict(all=('cat', [3, 2, 2])))

CASES['bb_dcn_v2_lowrank'] = ("""
  blocks { name: 'cross' inputs { feature_group_name: 'all' input_fn: 'lambda x: [x, x]' }
           recurrent { num_steps: 2 fixed_input_index: 0
                       keras_layer { class_name: 'Cross' st_params { fields { key: 'projection_dim' value { number_value: 3 } }
                                                                      fields { key: 'diag_scale' value { number_value: 0.5 } } } } } }
  blocks { name: 'deep' inputs { block_name: 'cross' } keras_layer { class_name: 'MLP' mlp { hidden_units: [5] } } }
""", OrderedDict(all=('cat', [3, 4])))

CASES['bb_dlrm'] = ("""
  blocks { name: 'bottom_mlp' inputs { feature_group_name: 'dense' } keras_layer { class_name: 'MLP' mlp { hidden_units: [6, 4] } } }
  blocks { name: 'sparse' inputs { feature_group_name: 'sparse' } input_layer { output_2d_tensor_and_feature_list: true } }
  blocks { name: 'dot'
           inputs { block_name: 'bottom_mlp' input_fn: 'lambda x: [x]' }
           inputs { block_name: 'sparse' input_fn: 'lambda x: x[1]' }
           keras_layer { class_name: 'DotInteraction' } }
  blocks { name: 'sparse_2d' inputs { block_name: 'sparse' input_fn: 'lambda x: x[0]' } }
  concat_blocks: ['sparse_2d', 'dot']
  top_mlp { hidden_units: [7, 3] }
""", OrderedDict(dense=('cat', [1, 1, 1, 1, 1]), sparse=('cat', [4, 4, 4])))

CASES['bb_wide_and_deep'] = ("""
  blocks { name: 'wide' inputs { feature_group_name: 'wide' } input_layer { wide_output_dim: 1 only_output_feature_list: true } }
  blocks { name: 'deep_logit' inputs { feature_group_name: 'deep' }
           keras_layer { class_name: 'MLP' mlp { hidden_units: [6, 4, 1] use_final_bn: false final_activation: 'linear' } } }
  blocks { name: 'final_logit'
           inputs { block_name: 'wide' input_fn: 'lambda x: tf.add_n(x)' }
           inputs { block_name: 'deep_logit' }
           merge_inputs_into_list: true
           keras_layer { class_name: 'Add' } }
  concat_blocks: 'final_logit'
""", OrderedDict(wide=('cat', [1, 1, 1, 1]), deep=('cat', [3, 4, 2])))

CASES['bb_deepfm'] = ("""
  blocks { name: 'wide_features' inputs { feature_group_name: 'wide_features' } input_layer { wide_output_dim: 1 } }
  blocks { name: 'wide_logit' inputs { block_name: 'wide_features' }
           lambda { expression: 'lambda x: tf.reduce_sum(x, axis=1, keepdims=True)' } }
  blocks { name: 'deep_features' inputs { feature_group_name: 'deep_features' }
           input_layer { output_2d_tensor_and_feature_list: true } }
  blocks { name: 'fm' inputs { block_name: 'deep_features' input_slice: '[1]' }
           keras_layer { class_name: 'FM' st_params { fields { key: 'use_variant' value { bool_value: true } } } } }
  blocks { name: 'deep' inputs { block_name: 'deep_features' input_slice: '[0]' }
           keras_layer { class_name: 'MLP' mlp { hidden_units: [8, 4] } } }
  concat_blocks: ['wide_logit', 'fm', 'deep']
  top_mlp { hidden_units: [6, 3] }
""", OrderedDict(wide_features=('cat', [1, 1, 1]), deep_features=('cat', [4, 4, 4])))

CASES['bb_xdeepfm'] = ("""
  blocks { name: 'wide' inputs { feature_group_name: 'wide' } input_layer { only_output_feature_list: true wide_output_dim: 1 } }
  blocks { name: 'features' inputs { feature_group_name: 'features' } input_layer { output_2d_tensor_and_feature_list: true } }
  blocks { name: 'cin' inputs { block_name: 'features' input_slice: '[1]' }
           extra_input_fn: 'lambda x: tf.stack(x, axis=1)'
           keras_layer { class_name: 'CIN' cin { hidden_feature_sizes: [5, 3] } } }
  blocks { name: 'dnn' inputs { block_name: 'features' input_slice: '[0]' }
           keras_layer { class_name: 'MLP' mlp { hidden_units: [8, 4] } } }
  blocks { name: 'final_logit'
           inputs { block_name: 'wide' input_fn: 'lambda x: tf.add_n(x)' }
           inputs { block_name: 'cin' }
           inputs { block_name: 'dnn' }
           keras_layer { class_name: 'MLP' mlp { hidden_units: [6, 1] use_final_bn: false final_activation: 'linear' } } }
  concat_blocks: 'final_logit'
""", OrderedDict(wide=('cat', [1, 1, 1, 1]), features=('cat', [3, 3, 3, 3])))

CASES['bb_din'] = ("""
  blocks { name: 'deep' inputs { feature_group_name: 'normal' } keras_layer { class_name: 'MLP' mlp { hidden_units: [8, 4] } } }
  blocks { name: 'seq_input' inputs { feature_group_name: 'sequence' } input_layer { output_seq_and_normal_feature: true } }
  blocks { name: 'DIN' inputs { block_name: 'seq_input' }
           keras_layer { class_name: 'DIN' din { attention_dnn { hidden_units: [6, 1] activation: 'relu' }
                                                need_target_feature: true } } }
  top_mlp { hidden_units: [6, 3] }
""", OrderedDict(normal=('cat', [3, 2, 2]), sequence=('seq', 4, 5, [4])))

CASES['bb_mmoe'] = ("""
  blocks { name: 'all' inputs { feature_group_name: 'all' } input_layer { only_output_feature_list: true } }
  blocks { name: 'senet' inputs { block_name: 'all' } keras_layer { class_name: 'SENet' senet { reduction_ratio: 4 } } }
  blocks { name: 'mmoe' inputs { block_name: 'senet' }
           keras_layer { class_name: 'MMoE' mmoe { num_task: 2 num_expert: 3 expert_mlp { hidden_units: [6, 4] } } } }
""", OrderedDict(all=('cat', [4, 4, 2, 4])))

# the remaining block mechanics: an implicit input block (a feature group named as an input without an input_layer
# block), sequential `layers`, `repeat` with input_fn / output_concat_axis, ignore_input, input_concat of two blocks,
# no concat_blocks (the leaves, in config order)
CASES['bb_mechanics'] = ("""
  blocks { name: 'tower' inputs { feature_group_name: 'user' }
           layers { keras_layer { class_name: 'MLP' mlp { hidden_units: [6] } } }
           layers { lambda { expression: 'lambda x: x * 2.0' } }
           layers { keras_layer { class_name: 'MLP' mlp { hidden_units: [4] use_final_bn: false } } } }
  blocks { name: 'heads' inputs { feature_group_name: 'item' }
           repeat { num_repeat: 3 output_concat_axis: 1 input_fn: 'lambda x, i: x + float(i)'
                    keras_layer { class_name: 'MLP' mlp { hidden_units: [2] } } } }
  blocks { name: 'joined' inputs { block_name: 'tower' } inputs { block_name: 'heads' ignore_input: true }
           inputs { feature_group_name: 'item' input_fn: 'lambda x: x[:, :2]' }
           keras_layer { class_name: 'MLP' mlp { hidden_units: [5] } } }
  blocks { name: 'side' inputs { block_name: 'heads' input_slice: '[:, 1:4]' } }
""", OrderedDict(user=('cat', [3, 2]), item=('cat', [2, 3])))

# standard Keras layers named directly (tensorflow.keras.layers.Dense: constructed from the st_params as keyword
# arguments, backbone.py:381-397; the shape of examples/configs/mlp_on_movielens.config without its Dropout layers)
CASES['bb_standard_keras'] = ("""
  blocks { name: 'mlp' inputs { feature_group_name: 'features' }
           layers { keras_layer { class_name: 'Dense' st_params { fields { key: 'units' value { number_value: 6 } }
                                                                 fields { key: 'activation' value { string_value: 'relu' } } } } }
           layers { keras_layer { class_name: 'Dense' st_params { fields { key: 'units' value { number_value: 4 } }
                                                                 fields { key: 'activation' value { string_value: 'sigmoid' } }
                                                                 fields { key: 'use_bias' value { bool_value: false } } } } }
           layers { keras_layer { class_name: 'Dense' st_params { fields { key: 'units' value { number_value: 1 } } } } } }
  blocks { name: 'side' inputs { feature_group_name: 'features' input_fn: 'lambda x: x[:, :3]' }
           keras_layer { class_name: 'Dense' st_params { fields { key: 'units' value { number_value: 2 } }
                                                         fields { key: 'activation' value { string_value: 'tanh' } } } } }
  concat_blocks: ['mlp', 'side']
""", OrderedDict(features=('cat', [3, 2, 2])))


def backbone_config(text):
  from google.protobuf import text_format

  from easyrec_amd.protos import backbone_pb2
  cfg = backbone_pb2.BackboneTower()
  text_format.Merge(text, cfg)
  return cfg

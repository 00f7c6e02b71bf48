"""EasyRecModel base class: the model-level plugin surface.

Mirror of reference easy_rec/python/model/easy_rec_model.py:41-183: constructor contract
`(model_config, feature_configs, features, labels, is_training)`, `build_predict_graph()`,
`build_loss_graph()`, `build_metric_graph()`, `get_outputs()`, regulariser resolution
(`embedding_regularization`, `l2_regularization` incl. the deprecated `dense_regularization`,
:129-155) and `build_input_layer` (:157-168).

Execution model: `features` are persistent device buffers (input/features.py); the `build_*_graph`
methods EXECUTE the forward on whatever batch is currently loaded (eager HIP launches that a hipGraph
can capture), instead of building a TF graph once.  `backward()` seeds torch.autograd with the loss
gradients produced by the fused loss kernels.
"""
import logging
from abc import abstractmethod

import six
import torch

from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.layers import input_layer
from easyrec_amd.utils import constant
from easyrec_amd.utils.load_class import get_register_class_meta

_EASY_REC_MODEL_CLASS_MAP = {}
_meta_type = get_register_class_meta(_EASY_REC_MODEL_CLASS_MAP, have_abstract_class=True)


class EasyRecModel(six.with_metaclass(_meta_type, object)):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    # the whole model config stays reachable; `_model_config` is narrowed to the class's own member by _take_config
    self._base_model_config = self._model_config = model_config
    self._feature_configs, self._feature_dict, self._labels = feature_configs, features, labels
    self._is_training, self._is_predicting = is_training, labels is None
    self._global_ev_params = model_config.ev_params if model_config.HasField('ev_params') else None
    # regularisers: a coefficient of 0 means "none" (easy_rec_model.py:66-77)
    self._emb_reg = self.embedding_regularization or None
    self._l2_reg = self.l2_regularization or None
    self._wide_output_dim = self._declared_wide_output_dim()
    self.build_input_layer(model_config, feature_configs)
    self._metric_dict = {}
    self.begin_step()
    weight = getattr(features, 'sample_weight', None)
    self._sample_weight = 1.0 if weight is None else weight
    self._backbone_net = self.build_backbone_network()

  def _declared_wide_output_dim(self):
    """Wide feature groups are embeddings of dimension wide_output_dim.  WideAndDeep / DeepFM / FM set it from their own
    config (in their build_input_layer); a backbone model takes it from the `input_layer { wide_output_dim }` of its
    blocks (easy_rec_model.py:79-84); otherwise -1: no wide group may be used."""
    if self.has_backbone:
      from easyrec_amd.layers.backbone import Backbone
      declared = Backbone.wide_embed_dim(self._base_model_config.backbone)
      if declared:
        logging.info('set `wide_output_dim` to %d' % declared)
        return declared
    return -1

  # -- what every model class of this package does around its own wiring
  def _take_config(self, member):
    """Narrow `_model_config` to the class's own member of the `model` oneof; a config for another class is refused."""
    chosen = self._model_config.WhichOneof('model')
    if chosen != member:
      raise AssertionError('invalid model config: %s' % chosen)
    self._model_config = getattr(self._model_config, member)
    return self._model_config

  def _group(self, name):
    """(concatenated output, per-feature outputs) of a feature group, through the input layer"""
    return self._input_layer(self._feature_dict, name)

  def _dnn(self, x, config, name, defer_last_apply=False):
    """a layers/dnn.py DNN with this model's kernel regulariser and training flag"""
    from easyrec_amd.layers import dnn
    return dnn.DNN(config, self._l2_reg, name, self._is_training)(x, defer_last_apply=defer_last_apply)

  def _emit(self, output):
    self._add_to_prediction_dict(output)
    return self._prediction_dict

  def build_backbone_network(self):
    """reference easy_rec_model.py:100-107"""
    if self.has_backbone:
      from easyrec_amd.layers.backbone import Backbone
      return Backbone(self._base_model_config.backbone, self._feature_dict, input_layer=self._input_layer,
                      l2_reg=self._l2_reg)
    return None

  @property
  def backbone(self):
    """Executes the backbone network on the batch currently loaded (reference easy_rec_model.py:113-127)."""
    if self._backbone_net is None:
      return None
    kwargs = {
        'loss_dict': self._loss_dict,
        'metric_dict': self._metric_dict,
        'prediction_dict': self._prediction_dict,
        'labels': self._labels,
        constant.SAMPLE_WEIGHT: self._sample_weight,
    }
    return self._backbone_net(self._is_training, **kwargs)

  @property
  def has_backbone(self):
    return self._base_model_config.HasField('backbone')

  @property
  def embedding_regularization(self):
    return self._base_model_config.embedding_regularization

  @property
  def feature_groups(self):
    return self._base_model_config.feature_groups

  @property
  def l2_regularization(self):
    """The kernel regulariser's coefficient of the model's own config: `l2_regularization`, or its deprecated alias
    `dense_regularization` when that one is set (easy_rec_model.py:143-155); 0 when the config has neither."""
    own = getattr(self._base_model_config, self._base_model_config.WhichOneof('model'))
    fields = own.DESCRIPTOR.fields_by_name
    if 'dense_regularization' in fields and own.HasField('dense_regularization'):
      logging.warning('dense_regularization is deprecated, please use l2_regularization')
      return own.dense_regularization
    return own.l2_regularization if 'l2_regularization' in fields else 0.0

  def build_input_layer(self, model_config, feature_configs):
    dropout = model_config.variational_dropout if model_config.HasField('variational_dropout') else None
    self._input_layer = input_layer.InputLayer(
        feature_configs, model_config.feature_groups, wide_output_dim=self._wide_output_dim,
        ev_params=self._global_ev_params, embedding_regularizer=self._emb_reg, kernel_regularizer=self._l2_reg,
        variational_dropout_config=dropout, is_training=self._is_training, is_predicting=self._is_predicting,
        engine=context.current().engine)

  def begin_step(self):
    """Per-step state, emptied by the estimator before build_predict_graph (and once by the constructor)."""
    self._prediction_dict, self._loss_dict, self._backward_seeds = {}, {}, []
    ctx = context._stack()[-1] if context._stack() else None
    if ctx is not None and hasattr(ctx, 'heads'):
      ctx.heads.clear()
      del ctx.tail_jobs[:]
      ctx.grad_slots.clear()
      if getattr(ctx, 'bf16_state', None) is not None:
        ctx.bf16_state.begin_step()

  @abstractmethod
  def build_predict_graph(self):
    pass

  @abstractmethod
  def build_loss_graph(self):
    pass

  def build_metric_graph(self, eval_config):
    return self._metric_dict

  @abstractmethod
  def get_outputs(self):
    pass

  def build_output_dict(self):
    """name -> tensor for every exported output (easy_rec_model.py:175-183)"""
    missing = [n for n in self.get_outputs() if n not in self._prediction_dict]
    if missing:
      raise KeyError('output node {} not in prediction_dict, can not be exported'.format(missing[0]))
    return {n: self._prediction_dict[n] for n in self.get_outputs()}

  def backward(self, flush=True):
    """Back-propagate the loss gradients recorded by build_loss_graph through the dense graph.  flush=False leaves
    the queued weight gradients to the caller (`kernels.hip().flush_wgrads()`), which may contract them on another
    stream while the embedding backward runs."""
    if not self._backward_seeds:
      return
    tensors = [t for t, _ in self._backward_seeds]
    grads = [g.reshape(t.shape) for t, g in self._backward_seeds]
    be = kernels.hip()
    be.defer_wgrads()  # the layers' weight gradients (x^T . dz, K = batch) are contracted in one grouped launch
    ok = False
    try:
      torch.autograd.backward(tensors, grads)
      ok = True
    finally:
      if flush or not ok:
        be.flush_wgrads()

  def get_grouped_vars(self, opt_num):
    assert opt_num == 2, 'could only support 2 optimizers, one for embedding, one for the other layers'
    return ['embedding', 'dense']

"""RankModel: logits -> predictions -> loss.

API and key names of reference easy_rec/python/model/rank_model.py:20-332: prediction keys `logits` / `probs` (binary
heads), `y` (regression heads), each with the tower suffix; loss keys `cross_entropy_loss` / `l2_loss` (+ suffix) or
the configured `loss_name`; `losses { loss_type, weight, loss_name }` lists with the Fixed weight strategy.

Here a head is described once by a `_Head` record (which prediction keys it writes, which of them the loss reads, the
default loss name) and the three methods that the reference spells out per loss type
(`_output_to_prediction_impl`, `_build_loss_impl`, `_get_outputs_impl`) are lookups in that table.  The loss itself is
one fused HIP launch that also yields d(loss)/d(logits) (builders/loss_builder.py); `build_loss_graph` records the
(prediction, gradient) pairs that seed the backward pass.
"""
import logging
from collections import namedtuple

import torch

from easyrec_amd import kernels
from easyrec_amd.builders import loss_builder
from easyrec_amd.layers.dnn import dense
from easyrec_amd.model.easy_rec_model import EasyRecModel
from easyrec_amd.protos.loss_pb2 import LossType

# outputs: exported prediction keys; loss_input: the key the loss reads; loss_key: default name in the loss dict
_Head = namedtuple('_Head', 'kind outputs loss_input loss_key')
_BINARY = _Head('binary', ('probs', 'logits'), 'logits', 'cross_entropy_loss')
_REGRESSION = _Head('regression', ('y',), 'y', 'l2_loss')
_SIGMOID_REGRESSION = _Head('sigmoid_regression', ('y',), 'y', 'l2_loss')

# (the other binary loss types name their loss after the type: rank_model.py:241-246)
_F1_REWEIGHTED = _Head('binary', ('probs', 'logits'), 'logits', 'f1_reweighted_loss')

_PAIR_WISE = _Head('binary', ('probs', 'logits'), 'logits', 'pair_wise_loss')

_HEADS = {
    LossType.PAIR_WISE_LOSS: _PAIR_WISE,
    LossType.CLASSIFICATION: _BINARY,
    LossType.BINARY_CROSS_ENTROPY_LOSS: _BINARY,
    LossType.F1_REWEIGHTED_LOSS: _F1_REWEIGHTED,
    LossType.L2_LOSS: _REGRESSION,
    LossType.SIGMOID_L2_LOSS: _SIGMOID_REGRESSION,
}
# loss types whose PREDICTIONS are those of a binary head but whose loss is not built on this path
_BINARY_PREDICTION_ONLY = (LossType.BINARY_FOCAL_LOSS,)


def _head_of(loss_type, for_loss=True):
  head = _HEADS.get(loss_type)
  if head is None and not for_loss and loss_type in _BINARY_PREDICTION_ONLY:
    head = _BINARY
  if head is None:
    raise ValueError('%s loss type: %s' % ('invalid' if for_loss else 'unsupported', LossType.Name(loss_type)))
  return head


class RankModel(EasyRecModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(RankModel, self).__init__(model_config, feature_configs, features, labels, is_training)
    cfg = self._model_config
    self._loss_type, self._num_class, self._losses = cfg.loss_type, cfg.num_class, cfg.losses
    self._outputs = []
    if self._labels is not None:
      # the configured label, else the first label field
      self._label_name = model_config.label_name if model_config.HasField('label_name') else next(iter(self._labels))

  # -- predictions
  def build_predict_graph(self):
    """Backbone models: backbone output (+ a `output` projection when its width is not num_class)."""
    if not self.has_backbone:
      raise NotImplementedError(
          'method `build_predict_graph` must be implemented when backbone network do not exits')
    assert self._model_config.WhichOneof('model') == 'model_params', '`model_params` must be configured'
    self._outputs.extend(self._model_config.model_params.outputs)
    logits = self.backbone
    if int(logits.shape[-1]) != self._num_class:
      logging.info('add head logits layer for rank model')
      logits = dense(logits, self._num_class, 'output', head=True)
    self._add_to_prediction_dict(logits)
    return self._prediction_dict

  def _output_to_prediction_impl(self, output, loss_type, num_class=1, suffix='', **kwargs):
    head = _head_of(loss_type, for_loss=False)
    column = output.squeeze(1)
    if head.kind == 'binary':
      assert num_class == 1, 'num_class > 1 (softmax heads) is outside the hot-path scope'
      # while training, the fused loss kernel produces the probabilities; otherwise they are computed here
      probs = None if self._is_training else torch.sigmoid(column.detach())
      return {'logits' + suffix: column, 'probs' + suffix: probs}
    # (a regression head reads its output right here or through a plain torch loss: a lazily computed logit head -
    # layers/dnn.py dense(head=True) - is computed now)
    pending = kernels.pending_head(column)
    if pending is not None:
      kernels.materialize_head(pending)
    if head is _SIGMOID_REGRESSION:
      column = torch.sigmoid(column)
    return {'y' + suffix: column}

  def _loss_types(self):
    """The loss types this model's single output feeds: the `losses` list, else the one `loss_type`."""
    return [entry.loss_type for entry in self._losses] or [self._loss_type]

  def _add_to_prediction_dict(self, output):
    for loss_type in self._loss_types():
      self._prediction_dict.update(self._output_to_prediction_impl(output, loss_type, num_class=self._num_class))

  # -- losses
  def _build_loss_impl(self, loss_type, label_name, loss_weight=1.0, num_class=1, suffix='', loss_name='',
                       loss_param=None, loss_scale=1.0):
    head = _head_of(loss_type)
    pred = self._prediction_dict[head.loss_input + suffix]
    value, dpred = loss_builder.build(loss_type, self._labels[label_name], pred, loss_weight, num_class,
                                      loss_scale=loss_scale, loss_param=loss_param, fuse_head=len(self._losses) <= 1)
    self._backward_seeds.append((pred, dpred))
    # rank_model.py:236-250: a given loss_name stands as it is for the cross-entropy / L2 losses and gets the suffix for
    # the other binary loss types
    if loss_name and head not in (_BINARY, _REGRESSION, _SIGMOID_REGRESSION):
      loss_name = loss_name + suffix
    return {loss_name or (head.loss_key + suffix): value}

  def _build_losses_impl(self, entries):
    """Several _build_loss_impl calls ([kwargs]) whose sigmoid cross-entropy heads share one launch
    (loss_builder.build_many); loss dict entries and backward seeds in the entries' order."""
    specs, metas = [], []
    for kw in entries:
      head = _head_of(kw['loss_type'])
      suffix = kw.get('suffix', '')
      pred = self._prediction_dict[head.loss_input + suffix]
      specs.append(dict(loss_type=kw['loss_type'], label=self._labels[kw['label_name']], pred=pred,
                        loss_weight=kw.get('loss_weight', 1.0), num_class=kw.get('num_class', 1),
                        loss_scale=kw.get('loss_scale', 1.0), loss_param=kw.get('loss_param')))
      metas.append((head, suffix, kw.get('loss_name', ''), pred))
    out = {}
    for (head, suffix, loss_name, pred), (value, dpred) in zip(metas, loss_builder.build_many(specs)):
      self._backward_seeds.append((pred, dpred))
      if loss_name and head not in (_BINARY, _REGRESSION, _SIGMOID_REGRESSION):
        loss_name = loss_name + suffix
      out[loss_name or (head.loss_key + suffix)] = value
    return out

  def build_loss_graph(self):
    common = dict(label_name=self._label_name, loss_weight=self._sample_weight, num_class=self._num_class)
    if len(self._losses) == 0:
      self._loss_dict.update(self._build_loss_impl(self._loss_type, **common))
      return self._loss_dict
    base = self._base_model_config
    assert base.loss_weight_strategy == base.Fixed, 'only the Fixed loss weight strategy is supported'
    for entry in self._losses:
      which = entry.WhichOneof('loss_param')
      self._loss_dict.update(self._build_loss_impl(entry.loss_type, loss_name=entry.loss_name, loss_scale=entry.weight,
                                                   loss_param=getattr(entry, which) if which else None, **common))
    return self._loss_dict

  # -- exported outputs
  def _get_outputs_impl(self, loss_type, num_class=1, suffix=''):
    return [key + suffix for key in _head_of(loss_type).outputs]

  def get_outputs(self):
    names = []
    for loss_type in self._loss_types():
      for name in self._get_outputs_impl(loss_type, self._num_class):
        if name not in names:
          names.append(name)
    return names

"""Loss construction (reference easy_rec/python/builders/loss_builder.py:28-53).

CLASSIFICATION with num_class == 1 -> tf.losses.sigmoid_cross_entropy(label, logits, weights)
(reduction SUM_BY_NONZERO_WEIGHTS): one HIP launch (`er_sigmoid_ce_fwd_bwd`) produces the loss, the
probabilities and d(loss)/d(logits); the backward of the model is seeded with that gradient.
"""
import torch

from easyrec_amd import kernels
from easyrec_amd.protos.loss_pb2 import LossType


def build(loss_type, label, pred, loss_weight=1.0, num_class=1, loss_scale=1.0, **kwargs):
  """Returns (loss [1] tensor, d loss / d pred).  `loss_weight`: scalar or per-example tensor."""
  if loss_type in (LossType.CLASSIFICATION, LossType.BINARY_CROSS_ENTROPY_LOSS):
    assert num_class == 1, 'multi-class softmax cross entropy is outside the hot-path scope'
    weights = loss_weight if torch.is_tensor(loss_weight) else None
    scale = loss_scale * (1.0 if torch.is_tensor(loss_weight) else float(loss_weight))
    labels = label if label.dtype == torch.float32 else label.to(torch.float32)
    loss, dlogits, _ = kernels.hip().sigmoid_ce(pred.detach().contiguous(), labels.contiguous(), weights,
                                                scale)
    return loss, dlogits
  if loss_type == LossType.L2_LOSS:
    # tf.losses.mean_squared_error: mean((label - pred)^2) (elementwise torch ops; not the hot path)
    labels = label.to(torch.float32)
    p = pred.detach()
    diff = p - labels
    w = loss_weight if torch.is_tensor(loss_weight) else torch.full_like(diff, float(loss_weight))
    nz = (w != 0).sum().clamp(min=1).to(torch.float32)
    loss = (w * diff * diff).sum().reshape(1) / nz * loss_scale
    return loss, 2.0 * w * diff / nz * loss_scale
  raise ValueError('unsupported loss type on the MI355X path: %s' % LossType.Name(loss_type))

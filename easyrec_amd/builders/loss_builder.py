"""Loss construction (reference easy_rec/python/builders/loss_builder.py:28-53).

CLASSIFICATION with num_class == 1 -> tf.losses.sigmoid_cross_entropy(label, logits, weights)
(reduction SUM_BY_NONZERO_WEIGHTS): one HIP launch (`er_sigmoid_ce_fwd_bwd`) produces the loss, the
probabilities and d(loss)/d(logits); the backward of the model is seeded with that gradient.
"""
import torch

from easyrec_amd import kernels
from easyrec_amd.protos.loss_pb2 import LossType


def f1_reweight_sigmoid_cross_entropy(labels, logits, beta_square, weights=None):
  """reference easy_rec/python/loss/f1_reweight_loss.py:10-39 ("Adaptive Scaling for Sparse Detection"): sigmoid cross
  entropy with weight 1 on the positives and tp / (beta^2 * #pos + #neg - tn + 1e-8) on the negatives, tp = sum of the
  batch's probabilities, tn = B - tp; reduction SUM_BY_NONZERO_WEIGHTS.  The negatives' weight is a function of the
  logits and the reference does not stop its gradient: plain autograd ops here (a loss variant, not the hot path), so
  that d(loss)/d(logits) carries that term as TensorFlow's does."""
  y = labels.to(torch.float32)
  probs = torch.sigmoid(logits)
  batch = float(y.shape[0])
  num_pos = y.sum()
  tp = probs.sum()
  neg_weight = tp / (beta_square * num_pos + (batch - num_pos) - (batch - tp) + 1e-8)
  w = torch.where(y == 1.0, torch.ones_like(y), neg_weight.expand_as(y))
  if weights is not None:
    w = w * weights
  pos = logits >= 0  # tf.nn.sigmoid_cross_entropy_with_logits' own form (exact derivative at a zero logit)
  ce = torch.where(pos, logits, torch.zeros_like(logits)) - logits * y + \
      torch.log1p(torch.exp(torch.where(pos, -logits, logits)))
  present = (w != 0).sum().clamp(min=1).to(torch.float32)
  return (w * ce).sum() / present


def pairwise_loss(labels, logits, margin=0.0, temperature=1.0, weights=None):
  """reference easy_rec/python/loss/pairwise_loss.py:15-70 (the deprecated `pairwise_loss`, without session ids): over
  the ordered pairs (i, j) with label_i > label_j, sigmoid cross entropy of z_i - z_j - margin against 1, i.e. the mean of
  softplus(-(z_i - z_j - margin)); per-example weights weigh the pairs of their FIRST member.  No pair: the mean of
  nothing - 0 with SUM_BY_NONZERO_WEIGHTS' safe division."""
  y = labels.to(torch.float32)
  z = logits / temperature if temperature != 1.0 else logits
  diff = z[:, None] - z[None, :] - margin
  mask = y[:, None] > y[None, :]
  x = diff[mask]
  per = torch.where(x >= 0, x, torch.zeros_like(x)) - x + torch.log1p(torch.exp(torch.where(x >= 0, -x, x)))
  if weights is None:
    return per.sum() / max(int(x.numel()), 1)
  w = weights[:, None].expand_as(diff)[mask]
  return (w * per).sum() / (w != 0).sum().clamp(min=1).to(torch.float32)


_SIGMOID_CE = (LossType.CLASSIFICATION, LossType.BINARY_CROSS_ENTROPY_LOSS)


def build_many(specs):
  """[dict(loss_type, label, pred, loss_weight, num_class, loss_scale, loss_param)] -> [(loss, d loss / d pred)], one per
  entry, in order.  The sigmoid cross-entropy heads among them (the towers of a multi-task model: the reference calls
  tf.losses once per tower, model/multi_task_model.py:228-275) share ONE launch (er_sigmoid_ce_multi: each head's own
  workgroup runs the single-head kernel's body - same values); everything else goes through build()."""
  out = [None] * len(specs)
  fused = [i for i, sp in enumerate(specs) if sp['loss_type'] in _SIGMOID_CE and sp.get('num_class', 1) == 1]
  be = kernels.hip()
  if len(fused) >= 2 and hasattr(be, 'sigmoid_ce_multi'):
    heads = []
    for i in fused:
      sp = specs[i]
      lw = sp.get('loss_weight', 1.0)
      label = sp['label']
      heads.append((sp['pred'].detach().contiguous(), (label if label.dtype == torch.float32 else label.to(torch.float32)).contiguous(),
                    lw if torch.is_tensor(lw) else None,
                    sp.get('loss_scale', 1.0) * (1.0 if torch.is_tensor(lw) else float(lw))))
    for i, res in zip(fused, be.sigmoid_ce_multi(heads)):
      out[i] = res
  for i, sp in enumerate(specs):
    if out[i] is None:
      out[i] = build(sp['loss_type'], sp['label'], sp['pred'], sp.get('loss_weight', 1.0), sp.get('num_class', 1),
                     loss_scale=sp.get('loss_scale', 1.0), loss_param=sp.get('loss_param'))
  return out


def _fused_head(head, label, scale):
  """The pending logit head of a rank model (layers/dnn.py dense(head=True)) under plain sigmoid cross entropy: ONE launch
  computes the projection, the loss, d loss / d logits and the projection's backward (kernels.HipBackend.head_sigmoid_ce);
  the loss value and the head's dW / db are finished by the loss tail's launch (EasyRecEstimator._loss_tail)."""
  from easyrec_amd.core import context
  be = kernels.hip()
  labels = label if label.dtype == torch.float32 else label.to(torch.float32)
  res = be.head_sigmoid_ce(head.x, head.w, head.b, labels.contiguous(), scale, src=head.src, logits=head.logits)
  B, K = head.x.shape
  head.state, head.dx, head.dz = 'fused', res['dx'], res['dlogits']
  if head.src is not None:  # the producing layer's BatchNorm backward finds its column sums ready (kernels.LinearBNActFn)
    head.src.partial, head.src.dx_ptr = res['bn_partials'].view(-1), res['dx'].data_ptr()
  ctx = context.current()
  wb = res['wb_partials']
  ctx.tail_jobs.append((wb, head.w_grad.view(-1), K))
  if head.b is not None:
    ctx.tail_jobs.append((wb[:, K:], head.b_grad.view(-1), 1))
  loss = torch.empty(1, dtype=torch.float32, device=head.x.device)
  loss._er_partials = (res['loss_partials'], float(scale), float(B))  # loss = scale * sum / B (the loss tail's launch)
  return loss, res['dlogits'], res['probs']


def build(loss_type, label, pred, loss_weight=1.0, num_class=1, loss_scale=1.0, loss_param=None, **kwargs):
  """Returns (loss [1] tensor, d loss / d pred).  `loss_weight`: scalar or per-example tensor."""
  head = kernels.pending_head(pred)
  if head is not None:
    if loss_type in _SIGMOID_CE and num_class == 1 and not torch.is_tensor(loss_weight) and head.fusable and \
        kwargs.get('fuse_head', True):
      loss, dlogits, _ = _fused_head(head, label, loss_scale * float(loss_weight))
      return loss, dlogits
    kernels.materialize_head(head)  # any other loss reads the logits
  if loss_type == LossType.PAIR_WISE_LOSS:
    assert num_class == 1, 'num_class must be 1 when loss type is PAIR_WISE_LOSS'
    assert loss_param is None or not loss_param.session_name, 'session ids in pairwise losses are outside the hot-path scope'
    margin = 0.0 if loss_param is None else float(loss_param.margin)
    temperature = 1.0 if loss_param is None else float(loss_param.temperature)
    z = pred.detach().clone().requires_grad_(True)
    with torch.enable_grad():
      # (a python float weight is not a "numeric tensor" for the reference: it multiplies the mean as a scalar)
      loss = pairwise_loss(label, z, margin, temperature, loss_weight if torch.is_tensor(loss_weight) else None)
      loss = loss * (loss_scale if torch.is_tensor(loss_weight) else loss_scale * float(loss_weight))
      if loss.requires_grad:
        dz, = torch.autograd.grad(loss, z)
      else:  # no (positive, negative) pair in the batch
        dz = torch.zeros_like(z)
    return loss.detach().reshape(1), dz
  if loss_type == LossType.F1_REWEIGHTED_LOSS:
    assert num_class == 1, 'num_class must be 1 when loss type is F1_REWEIGHTED_LOSS'
    beta_square = 1.0 if loss_param is None else float(loss_param.f1_beta_square)
    assert loss_param is None or not loss_param.label_smoothing, 'label smoothing is outside the hot-path scope'
    z = pred.detach().clone().requires_grad_(True)
    with torch.enable_grad():
      weights = loss_weight if torch.is_tensor(loss_weight) else torch.full_like(z, float(loss_weight))
      loss = f1_reweight_sigmoid_cross_entropy(label, z, beta_square, weights) * loss_scale
      dz, = torch.autograd.grad(loss, z)
    return loss.detach().reshape(1), dz
  if loss_type in (LossType.CLASSIFICATION, LossType.BINARY_CROSS_ENTROPY_LOSS):
    assert num_class == 1, 'multi-class softmax cross entropy is outside the hot-path scope'
    weights = loss_weight if torch.is_tensor(loss_weight) else None
    scale = loss_scale * (1.0 if torch.is_tensor(loss_weight) else float(loss_weight))
    labels = label if label.dtype == torch.float32 else label.to(torch.float32)
    loss, dlogits, _ = kernels.hip().sigmoid_ce(pred.detach().contiguous(), labels.contiguous(), weights,
                                                scale)
    return loss, dlogits
  if loss_type in (LossType.L2_LOSS, LossType.SIGMOID_L2_LOSS):
    # tf.losses.mean_squared_error: mean((label - pred)^2) (elementwise torch ops; not the hot path).  For
    # SIGMOID_L2_LOSS `pred` is already sigmoid(output) (rank_model.py:126-128) and autograd carries dz through it
    labels = label.to(torch.float32)
    p = pred.detach()
    diff = p - labels
    w = loss_weight if torch.is_tensor(loss_weight) else torch.full_like(diff, float(loss_weight))
    nz = (w != 0).sum().clamp(min=1).to(torch.float32)
    loss = (w * diff * diff).sum().reshape(1) / nz * loss_scale
    return loss, 2.0 * w * diff / nz * loss_scale
  raise ValueError('unsupported loss type on the MI355X path: %s' % LossType.Name(loss_type))

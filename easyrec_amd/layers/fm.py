"""FM second-order interaction (reference easy_rec/python/layers/fm.py:9-26).

`0.5 * ((sum_f e_f)^2 - sum_f e_f^2)` kept as [B, D].  The reference stacks the F field tensors
([B,F,D] copy) and runs 4 elementwise/reduction ops; here one HIP kernel streams the fields straight
out of the input-layer concat buffer (no stack), saving sum_f e_f for the backward.
"""
import torch

from easyrec_amd import kernels


class FM(object):

  def __init__(self, name='fm'):
    self._name = name

  @staticmethod
  def group_block(fm_fea):
    """(x, F, D, sink, col0) when the fields are one uniform column block of an embedding group output, else None."""
    blk = fm_fea.uniform_block() if hasattr(fm_fea, 'uniform_block') else None
    if blk is None:
      return None
    base, col0, F, D = blk
    x = base if (col0 == 0 and base.shape[1] == F * D) else base[:, col0:col0 + F * D]
    return x, F, D, kernels.grad_sink_of(base), col0

  def __call__(self, fm_fea):
    blk = fm_fea.uniform_block() if hasattr(fm_fea, 'uniform_block') else None
    sink, col0 = None, 0
    if blk is not None:
      base, col0, F, D = blk
      x = base if (col0 == 0 and base.shape[1] == F * D) else base[:, col0:col0 + F * D]
      sink = kernels.grad_sink_of(base)  # the block lives in an embedding group output: deposit its gradient there
    else:
      F, D = len(fm_fea), fm_fea[0].shape[1]
      if any(t.shape[1] != D for t in fm_fea):  # (the reference's tf.stack rejects it when the graph is built)
        raise ValueError('FM %s: every field must have the same embedding_dim, got %s' %
                         (self._name, [int(t.shape[1]) for t in fm_fea]))
      x = torch.cat(list(fm_fea), dim=1)
    return kernels.FMFn.apply(x, F, D, sink, col0)

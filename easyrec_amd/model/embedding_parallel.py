"""Multi-GPU training step: row-sharded embedding tables + data-parallel dense variables.

Mirror of the reference's `train_distribute: EmbeddingParallelStrategy` path
(protos/train.proto:26-27, docs/source/train.md:191-227): one process per GPU; each rank reads its
own B examples (weak scaling); embedding tables are sharded by `id % world`
(layers/sharded_embedding.py); dense variables are replicated, their gradients all-reduced and
AVERAGED (compat/optimizers.py:328-331); embedding gradients are divided by the world size
(:315-316); BatchNorm statistics stay per-rank (the reference does not sync them); dense variables
start identical on every rank (same seed; the reference broadcasts rank 0's, utils/hvd_utils.py:43-56).

Per step: 1 small all-gather (split sizes, the one host sync) + 3 RCCL all-to-alls per embedding-dim
group (keys, rows, row gradients) + 1 all-reduce of the flat dense-gradient buffer + 1 all-reduce of
the replicated small tables' gradients.  The whole-step hipGraph of the single-GPU path does not apply
(the all-to-all split sizes change every step), so launches are eager here.
"""
import torch

from easyrec_amd.core.comm import LocalComm, TorchDistComm
from easyrec_amd.layers.sharded_embedding import ShardedEmbeddingEngine
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator


class EmbeddingParallelEstimator(EasyRecEstimator):

  def __init__(self, pipeline_config, device='cuda', batch_size=None, seed=0, rank=0, world=1, comm=None,
               schema_kwargs=None, is_training=True, replicate_bytes=256 * 1024, recv_slack=2.0):
    if comm is None:
      comm = TorchDistComm() if world > 1 else LocalComm()
    assert comm.rank == rank and comm.world == world, 'comm (%d/%d) does not match rank/world (%d/%d)' % (
        comm.rank, comm.world, rank, world)
    self.comm = comm
    self.rank, self.world = rank, world
    self._engine_kwargs = dict(replicate_bytes=replicate_bytes, recv_slack=recv_slack)
    super(EmbeddingParallelEstimator, self).__init__(pipeline_config, device=device, batch_size=batch_size, seed=seed,
                                                     schema_kwargs=schema_kwargs, is_training=is_training,
                                                     overlap_sweep=False)
    # embedding gradients are divided by the world size (compat/optimizers.py:315-316)
    self.emb_grad_scale = self.emb_grad_scale / float(world)

  def _make_engine(self):
    return ShardedEmbeddingEngine(self.device, self.batch_size, self.comm, seed=self.seed, **self._engine_kwargs)

  def _dense_grad_scale(self):
    return 1.0 / float(self.world)  # hvd.allreduce(op=Average)

  def _sync_dense_grads(self):
    if self.world > 1:
      self.comm.all_reduce_sum(self.varstore.flat_grad)

  def capture(self, warmup=3):
    raise NotImplementedError('the embedding-parallel step is not graph-capturable: its all-to-all split sizes '
                              'are data dependent (one host sync per step)')

  def loss_values(self, average=False):
    vals = super(EmbeddingParallelEstimator, self).loss_values()
    if average and self.world > 1:
      keys = sorted(vals)
      t = torch.tensor([vals[k] for k in keys], dtype=torch.float64, device=self.device)
      self.comm.all_reduce_sum(t)
      vals = {k: float(v) / self.world for k, v in zip(keys, t.tolist())}
    return vals

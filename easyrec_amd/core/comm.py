"""Collectives of the embedding-parallel path: one process per GPU, RCCL over xGMI.

The reference issues these through Horovod (`hvd.alltoall(ids, splits)`, `hvd.alltoall(rows, ...)`,
`hvd.allreduce(grad, op=Average)`; compat/feature_column/feature_column.py:296-357,
compat/optimizers.py:285-345).  Here they are `torch.distributed` calls on device tensors: backend
"nccl" IS RCCL on ROCm; the CPU tests use "gloo".

  all_to_all_equal  fixed capacity per peer (padded buffers): numel / world elements to and from every peer, no
                    host-side sizes, nothing to synchronise on - the default exchange (layers/sharded_embedding.py):
                    per route one for [count, keys], one for rows, one for row gradients
  all_reduce_sum    dense gradients with the replicated small tables' gradients behind them (scaled by 1 / W in the
                    optimizer kernels)
  all_reduce_sum_async / wait
                    the same, issued on a SECOND communicator as soon as the dense backward has finished and joined right
                    before the optimizer kernels: it is in flight while the local gradient reduction runs and while the
                    gradient all-to-all of the first communicator is on the wire (the reference's Horovod issues one
                    all-reduce per dense gradient as backward produces it, compat/optimizers.py:315-331, overlapping
                    them with the rest of backward the same way)
  exchange_counts   compact exchange only (more than 16 ranks): all-gather of a small [G, W] int32 matrix of
                    per-owner unique-key counts -> host-side send / recv split lists (one host synchronisation a step)
  all_to_all        compact exchange only: variable-split exchange of keys (int32), rows and gradient rows
  all_gather_rows   checkpoint / test only: collect the shards of a table (all_gather_varlen: of a hash-table table)

Messages are small at B=4096 (<= 1 MB per peer), i.e. latency-bound on xGMI's point-to-point links: what counts is
how many collectives a step issues (4 for DeepFM), not their bytes.
"""
import torch


class LocalComm(object):
  """world == 1: every exchange is a local copy (the single-GPU degenerate of the same code path)."""

  rank = 0
  world = 1

  def exchange_counts(self, counts):
    c = counts.cpu().tolist()
    return c, c

  def all_to_all(self, send, send_splits, recv, recv_splits):
    n = int(send_splits[0])
    assert n == int(recv_splits[0])
    if n:
      recv[:n].copy_(send[:n])

  def all_to_all_equal(self, send, recv):
    recv.copy_(send)

  def all_reduce_sum(self, t):
    return t

  def all_reduce_sum_async(self, t):
    return None

  def wait(self, handle):
    pass

  def all_gather_rows(self, t):
    return [t]

  def all_gather_varlen(self, t):
    return [t]

  def barrier(self):
    pass


class TorchDistComm(object):
  """torch.distributed process group (nccl = RCCL on the MI355X node, gloo in the CPU tests)."""

  def __init__(self, group=None, overlap_group=None):
    """overlap_group: create the second communicator (None: only where the asynchronous all-reduce can be used - more than
    one rank, or EASYREC_AMD_EP_OVERLAP=1 - and not with EASYREC_AMD_EP_OVERLAP=0).  torch.distributed.new_group is
    collective over the DEFAULT process group: every rank of it - not only the members of `group` - must construct its
    TorchDistComm at the same point when the second communicator is created."""
    import os
    import torch.distributed as dist
    assert dist.is_initialized(), 'init_process_group first (bench.py / the launcher does it)'
    self.dist = dist
    self.group = group
    self.rank = dist.get_rank(group)
    self.world = dist.get_world_size(group)
    # a second communicator over the same ranks (collective: every rank constructs its comm at the same point): RCCL
    # serialises the collectives of ONE communicator on its stream, so the dense all-reduce that should overlap the
    # gradient all-to-all needs its own
    self.group2 = None
    if overlap_group is None:
      sw = os.environ.get('EASYREC_AMD_EP_OVERLAP', 'auto')
      overlap_group = sw == '1' or (sw != '0' and self.world > 1)
    if overlap_group:
      ranks = dist.get_process_group_ranks(group) if group is not None else None
      self.group2 = dist.new_group(ranks=ranks)

  def exchange_counts(self, counts):
    """counts: int32 [G, W] on device: counts[g, w] = unique keys of dim-group g this rank sends to w.
    Returns (send[g][w], recv[g][src]) as python ints."""
    G, W = counts.shape
    assert W == self.world
    gathered = torch.empty(self.world, G, W, dtype=counts.dtype, device=counts.device)
    self.dist.all_gather_into_tensor(gathered.view(-1), counts.contiguous().view(-1), group=self.group)
    allc = gathered.cpu()  # host sync: the split sizes of the following all-to-alls
    send = allc[self.rank].tolist()
    recv = allc[:, :, self.rank].t().contiguous().tolist()
    return send, recv

  def all_to_all(self, send, send_splits, recv, recv_splits):
    ns, nr = int(sum(send_splits)), int(sum(recv_splits))
    self.dist.all_to_all_single(recv[:nr], send[:ns], output_split_sizes=[int(x) for x in recv_splits],
                                input_split_sizes=[int(x) for x in send_splits], group=self.group)

  def all_to_all_equal(self, send, recv):
    """Equal splits (numel / world elements per peer): no host-side sizes, nothing to synchronise on."""
    self.dist.all_to_all_single(recv, send, group=self.group)

  def all_reduce_sum(self, t):
    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
    return t

  def all_reduce_sum_async(self, t):
    """-> handle for wait().  nccl (= RCCL): the collective runs on the second communicator's stream, ordered after the
    work already queued on the current stream; wait() makes the CURRENT STREAM wait for it (the host does not block).
    gloo (CPU tests): a background thread; wait() blocks the host."""
    return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group2 if self.group2 is not None else self.group,
                                async_op=True)

  def wait(self, handle):
    if handle is not None:
      handle.wait()

  def all_gather_rows(self, t):
    out = [torch.empty_like(t) for _ in range(self.world)]
    self.dist.all_gather(out, t.contiguous(), group=self.group)
    return out

  def all_gather_varlen(self, t):
    """checkpoint / test only: every rank's [n_r, ...] array (the n_r differ: hash-table shards)."""
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.empty_like(n) for _ in range(self.world)]
    self.dist.all_gather(sizes, n, group=self.group)
    sizes = [int(x.item()) for x in sizes]
    pad = torch.zeros((max(sizes + [1]),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(self.world)]
    self.dist.all_gather(out, pad, group=self.group)
    return [o[:k] for o, k in zip(out, sizes)]

  def barrier(self):
    self.dist.barrier(group=self.group)

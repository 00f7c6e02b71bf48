"""Test infrastructure: an in-process stand-in for the RCCL collectives so that W embedding-parallel
ranks can run as W threads on ONE GPU (gpurun exposes a single MI355X).  Same interface as
easyrec_amd.core.comm.TorchDistComm.  All ranks launch on the device's default stream, so stream
order + the thread barriers give the producer -> consumer ordering."""
import threading

import torch


class SimWorld(object):

  def __init__(self, world):
    self.world = world
    self.barrier = threading.Barrier(world)
    self.slots = [None] * world
    self.errors = []

  def comm(self, rank):
    return ThreadSimComm(self, rank)

  def run(self, fn):
    """fn(rank, comm) on every rank; returns the list of results; re-raises the first failure."""
    results = [None] * self.world

    def work(r):
      try:
        results[r] = fn(r, self.comm(r))
      except BaseException as e:  # noqa: BLE001
        self.errors.append(e)
        self.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
    for t in threads:
      t.start()
    for t in threads:
      t.join()
    if self.errors:
      raise self.errors[0]
    return results


class ThreadSimComm(object):

  def __init__(self, sim, rank):
    self.sim, self.rank, self.world = sim, rank, sim.world

  def _publish(self, obj):
    self.sim.slots[self.rank] = obj
    self.sim.barrier.wait()
    return list(self.sim.slots)

  def exchange_counts(self, counts):
    allc = self._publish(counts.cpu())
    send = allc[self.rank].tolist()
    recv = torch.stack([c[:, self.rank] for c in allc], dim=1).tolist()
    self.sim.barrier.wait()
    return send, recv

  def all_to_all(self, send, send_splits, recv, recv_splits):
    offs = [0]
    for n in send_splits:
      offs.append(offs[-1] + int(n))
    allp = self._publish((send, offs))
    pos = 0
    for src in range(self.world):
      buf, so = allp[src]
      n = so[self.rank + 1] - so[self.rank]
      assert n == int(recv_splits[src])
      if n:
        recv[pos:pos + n].copy_(buf[so[self.rank]:so[self.rank + 1]])
      pos += n
    self.sim.barrier.wait()

  def all_to_all_equal(self, send, recv):
    n = send.shape[0] // self.world
    allp = self._publish(send)
    for src in range(self.world):
      recv[src * n:(src + 1) * n].copy_(allp[src][self.rank * n:(self.rank + 1) * n])
    self.sim.barrier.wait()

  def all_reduce_sum(self, t):
    allt = self._publish(t)
    total = allt[0].clone()
    for o in allt[1:]:
      total += o
    self.sim.barrier.wait()
    t.copy_(total)
    self.sim.barrier.wait()
    return t

  def all_reduce_sum_async(self, t):
    self.all_reduce_sum(t)  # (threads on one device: nothing to overlap with)
    return None

  def wait(self, handle):
    pass

  def all_gather_rows(self, t):
    allt = self._publish(t.contiguous())
    out = [x.clone() for x in allt]
    self.sim.barrier.wait()
    return out

  def all_gather_varlen(self, t):
    return self.all_gather_rows(t)

  def barrier(self):
    self.sim.barrier.wait()

"""Gradient slots (kernels.grad_slot / slot_gate): the slot-aware consumers of one activation share a gradient buffer; an
ORDINARY consumer of the same activation (a torch op) must not break that, whatever the order the consumers were built in
(round-4 advisor finding: the buffer went stale when an ordinary gradient arrived between two slot consumers)."""
import itertools

import pytest
import torch

from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.core.variables import VarStore
from easyrec_amd.layers import dnn


def _build(order, x, ws, slots_on, monkeypatch):
  monkeypatch.setattr(kernels, '_GRAD_SLOTS', slots_on)
  vs = VarStore('cpu')
  ctx = context.ModelContext(vs, None, is_training=True)
  outs = []
  with context.use(ctx):
    for kind in order:
      if kind == 'tanh':
        outs.append(torch.tanh(x).sum())
      elif kind == 'mul':
        outs.append((x * x).sum() * 0.5)
      else:
        outs.append(dnn._linear(x, ws[kind], None).pow(2).sum())
    loss = sum(outs)
    (g,) = torch.autograd.grad(loss, x)
  return g


@pytest.mark.parametrize('order', list(itertools.permutations(['w1', 'tanh', 'w2'])) +
                         list(itertools.permutations(['w1', 'w2', 'w3', 'mul'])))
def test_mixed_consumers_in_every_order(ref_backend, monkeypatch, order):
  torch.manual_seed(3)
  base = torch.randn(8, 6)
  ws = {k: torch.randn(6, 5, requires_grad=True) for k in ('w1', 'w2', 'w3')}
  x1 = (base.clone().requires_grad_() * 1.0)  # a non-leaf activation, as inside a model
  x2 = (base.clone().requires_grad_() * 1.0)
  got = _build(order, x1, ws, True, monkeypatch)
  exp = _build(order, x2, ws, False, monkeypatch)
  assert torch.allclose(got, exp, rtol=1e-6, atol=1e-6), (order, (got - exp).abs().max())


def test_slot_consumers_alone_still_share_one_buffer(ref_backend, monkeypatch):
  """three dense layers on one activation: ONE gradient reaches the producer, no add of separate tensors"""
  monkeypatch.setattr(kernels, '_GRAD_SLOTS', True)
  torch.manual_seed(4)
  x = torch.randn(8, 6, requires_grad=True) * 1.0
  ws = [torch.randn(6, 5, requires_grad=True) for _ in range(3)]
  seen = []
  x.register_hook(lambda g: seen.append(g.clone()))
  ctx = context.ModelContext(VarStore('cpu'), None, is_training=True)
  with context.use(ctx):
    loss = sum(dnn._linear(x, w, None).pow(2).sum() for w in ws)
    loss.backward()
    assert len([k for k in ctx.grad_slots if k != 'gates']) == 1  # one slot for the three consumers
  exp = sum(2.0 * (x.detach() @ w.detach()) @ w.detach().t() for w in ws)
  assert len(seen) == 1 and torch.allclose(seen[0], exp, rtol=1e-5, atol=1e-5)


def test_two_view_objects_of_one_activation_share_one_gate(ref_backend, monkeypatch):
  """dense() on a 3-D input reshapes it on every call: two view OBJECTS with the same storage start, shape and strides.  Their
  slot consumers share ONE gradient buffer (grad_slot keys on storage start and shape), so they must share ONE gate too -
  with a gate per view object the first consumer hands the buffer to autograd through its own gate while the second is
  still adding into it (round-5 advisor finding).  Gradient: that of plain autograd."""
  torch.manual_seed(5)
  base = torch.randn(4, 3, 6)
  ws = {k: torch.randn(6, 5, requires_grad=True) for k in ('w1', 'w2')}

  def run(slots_on):
    monkeypatch.setattr(kernels, '_GRAD_SLOTS', slots_on)
    x3 = (base.clone().requires_grad_() * 1.0)
    vs = VarStore('cpu')
    ctx = context.ModelContext(vs, None, is_training=True)
    with context.use(ctx):
      a, b = x3.reshape(-1, 6), x3.reshape(-1, 6)  # two view objects
      if slots_on:
        ga, gb = kernels.slot_gate(a), kernels.slot_gate(b)
        assert ga is gb
      loss = dnn._linear(a, ws['w1'], None).pow(2).sum() + dnn._linear(b, ws['w2'], None).pow(2).sum()
      (g,) = torch.autograd.grad(loss, x3)
    return g

  assert torch.allclose(run(True), run(False), rtol=1e-6, atol=1e-6)

"""Deferred parameter gradients (cores/runtime.py) on CPU: the accumulate-inside-the-node path must step aside for
DistributedDataParallel (its reducer hooks live on the AccumulateGrad nodes, invisible from Python -- ADVICE round 2)
and for hooked parameters, and must survive a re-entrant backward (activation checkpointing)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from camliflow_amd.cores import runtime
from camliflow_amd.cores.blocks import _PointwiseConv


class _Net(nn.Module):
    """Two shared 1x1 convolutions applied three times, like the iteration-shared GRU weights."""

    def __init__(self, use_checkpoint=False):
        super().__init__()
        self.a = nn.Conv1d(6, 6, 1, bias=False)
        self.b = nn.Conv1d(6, 6, 1, bias=False)
        self.use_checkpoint = use_checkpoint

    def body(self, x):
        return torch.tanh(_PointwiseConv.apply(_PointwiseConv.apply(x, self.a.weight), self.b.weight))

    def forward(self, x):
        for _ in range(3):
            x = checkpoint(self.body, x, use_reentrant=True) if self.use_checkpoint else self.body(x)
        return x


def _grads(net, x, deferred):
    runtime.set_deferred_param_grads(deferred)
    try:
        net.zero_grad()
        net(x).square().sum().backward()
    finally:
        runtime.set_deferred_param_grads(False)
    return [p.grad.clone() for p in net.parameters()]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_deferred_equals_plain_and_leaves_nothing_behind():
    torch.manual_seed(0)
    net, x = _Net(), torch.randn(2, 6, 11, requires_grad=True)
    want = _grads(net, x, False)
    got = _grads(net, x, True)
    assert not runtime.PARAM_GRADS.tables
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6)


def test_deferred_survives_reentrant_backward():
    """use_reentrant=True checkpointing runs a nested backward per segment: each graph task keeps its own table."""
    torch.manual_seed(0)
    net, x = _Net(use_checkpoint=True), torch.randn(2, 6, 11, requires_grad=True)
    plain = _Net()
    plain.load_state_dict(net.state_dict())
    want = _grads(plain, x, False)
    got = _grads(net, x, True)
    assert not runtime.PARAM_GRADS.tables
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6)


def test_failed_backward_does_not_leak_into_the_next():
    torch.manual_seed(0)
    net, x = _Net(), torch.randn(2, 6, 11, requires_grad=True)
    want = _grads(net, x, False)
    runtime.set_deferred_param_grads(True)
    try:
        net.zero_grad()
        y = net(x)

        def boom(_g):
            raise RuntimeError('boom')
        x.register_hook(boom)
        try:
            y.square().sum().backward()
        except RuntimeError:
            pass
    finally:
        runtime.set_deferred_param_grads(False)
    x2 = x.detach().clone().requires_grad_(True)
    got = _grads(net, x2, True)
    assert len(runtime.PARAM_GRADS.tables) <= 1      # at most the failed task's leftovers, never merged into a live one
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6)
    runtime.PARAM_GRADS.tables.clear()


def test_hooked_parameter_is_not_deferred():
    torch.manual_seed(0)
    net, x = _Net(), torch.randn(2, 6, 11, requires_grad=True)
    seen = []
    net.a.weight.register_hook(lambda g: seen.append(g.clone()))
    _grads(net, x, True)
    assert len(seen) == 1        # the hook saw the (summed) gradient: the parameter went through autograd


def test_ddp_reducer_sees_every_gradient():
    """Under DistributedDataParallel the fused nodes hand their gradients to autograd (the reducer's hooks fire); with the
    round-2 guard the first step ended with unreduced gradients and the next forward raised."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net, x = _Net(), torch.randn(2, 6, 11)
        want = _grads(net, x, False)
        ddp = nn.parallel.DistributedDataParallel(net)
        runtime.set_deferred_param_grads(True)
        try:
            for _ in range(3):      # the second forward is where an unreduced first step used to fail
                ddp.zero_grad()
                ddp(x).square().sum().backward()
                assert not runtime.PARAM_GRADS.tables
                for p, w in zip(net.parameters(), want):
                    torch.testing.assert_close(p.grad, w, rtol=1e-5, atol=1e-6)
        finally:
            runtime.set_deferred_param_grads(False)
    finally:
        dist.destroy_process_group()


def test_accumulators_of_a_failed_backward_are_dropped_by_the_next_forward():
    """a backward that raises never runs its flush callback; its table (and the accumulators it holds) must not
    outlive the next forward, and the next step's gradients must be those of that step alone."""
    torch.manual_seed(0)
    net, x = _Net(), torch.randn(2, 6, 11, requires_grad=True)
    want = _grads(net, x, False)

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('simulated out-of-memory in a backward')

    runtime.set_deferred_param_grads(True)
    try:
        net.zero_grad()
        # the failing node sits UPSTREAM of the deferring ones: their backwards have run (table filled) when it raises
        loss = net(_Boom.apply(x)).square().sum()
        try:
            loss.backward()
        except RuntimeError:
            pass
        assert runtime.PARAM_GRADS.tables, 'the failed backward left its table behind (nothing flushed it)'
        net.zero_grad()
        net(x).square().sum().backward()        # next step: forward drops the stale table, backward flushes its own
        assert not runtime.PARAM_GRADS.tables
    finally:
        runtime.set_deferred_param_grads(False)
    for p, w in zip(net.parameters(), want):
        torch.testing.assert_close(p.grad, w, rtol=1e-5, atol=1e-6)

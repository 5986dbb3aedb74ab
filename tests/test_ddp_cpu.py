"""Batch-dimension data parallelism of the training step, world_size 2, gloo backend on CPU.

What bench.py does on RCCL is exercised here with the same functions (train_step,
allreduce_gradients, make_optimizer):
two ranks each take one sample; the averaged gradients / updated weights must equal a single
process stepping on the 2-sample batch.  BatchNorm is frozen (freeze_bn) because SyncBatchNorm has
no CPU implementation; the operators are the oracle-backed ones (no GPU in this container)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build(seed_inputs_rank=None):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import bench
    from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs
    from camliflow_amd.cores import CamLiRAFT
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2, freeze_bn=True))).train()
    full = synthetic_inputs(2, 128, 160, 4608, seed=3)
    return bench, model, full


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        bench, model, full = _build()
        from modelutils import oracle_boundary
        shard = {k: v[rank:rank + 1] for k, v in full.items()}
        opt = bench.make_optimizer(model)
        with oracle_boundary():
            model(shard)
            model.get_loss().backward()
            bench.allreduce_gradients(model, world)    # one flat-bucket all-reduce (mean)
            grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            model.zero_grad()
            loss = bench.train_step(model, opt, shard, world)   # and the full harness step, data parallel
        metrics = model.get_metrics()              # packed all-reduce across both ranks (empty after clear)
        assert isinstance(metrics, dict)
        assert all(torch.isfinite(p).all() for p in model.parameters())
        if rank == 0:
            torch.save({'grads': grads, 'loss': float(loss)}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_matches_single_process(tmp_path):
    out_path = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path)

    bench, model, full = _build()
    from modelutils import oracle_boundary
    torch.set_num_threads(4)
    with oracle_boundary():
        model(full)
        model.get_loss().backward()
    want = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert want.keys() == got['grads'].keys() and len(want) > 400
    num = sum(((got['grads'][n] - want[n]).double() ** 2).sum().item() for n in want) ** 0.5
    den = sum((want[n].double() ** 2).sum().item() for n in want) ** 0.5
    assert num / den < 1e-4, num / den      # mean of per-rank gradients == gradient of the 2-sample batch


def test_packed_metric_allreduce_single_process():
    """FlowModel.get_metrics(): one packed reduction; without a process group it is a plain mean."""
    from camliflow_amd.cores.objectives import FlowModel
    m = FlowModel()
    m.update_metrics('a', torch.tensor([1.0, 3.0]))
    m.update_metrics('a', torch.tensor([5.0]))
    m.update_metrics('b', torch.tensor([True, False, True, True]))
    out = m.get_metrics()
    assert abs(out['a'] - 3.0) < 1e-6 and abs(out['b'] - 0.75) < 1e-6

"""The multi-GPU code path of bench.py on ONE GPU: a 1-rank RCCL process group, SyncBatchNorm conversion, parameter
broadcast and the flat-bucket gradient all-reduce (bench.allreduce_gradients), against the plain single-process step
from the same weights.  With world size 1 every collective is an identity, so losses and gradients must agree to
float-atomic noise; what the test proves is that the forced-dist path (the one the driver's N > 1 runs take) executes
on RCCL with the two-lane execution and the deferred parameter gradients, and leaves the same gradients behind."""
import os
import socket

import pytest
import torch

from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_forced_dist_step_matches_plain_step():
    import torch.distributed as dist
    import bench
    from camliflow_amd.cores import CamLiRAFT, runtime
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.manual_seed(0)
    base = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=3)), scale=0.5)
    state = {k: v.clone() for k, v in base.state_dict().items()}
    inputs = {k: v.cuda() for k, v in synthetic_inputs(2, 128, 160, 4608).items()}

    def grads_of(model, world, force):
        model.zero_grad()
        model(inputs)
        loss = model.get_loss()
        loss.backward()
        bench.allreduce_gradients(model, world, force)
        torch.cuda.synchronize()
        return loss.item(), {n.replace('module.', ''): p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    runtime.set_overlap(True)
    runtime.set_deferred_param_grads(True)
    try:
        with runtime.use_backend('hip'):
            plain = base.cuda().train()
            loss0, g0 = grads_of(plain, 1, False)
            # as bench.py: no device_id (an eagerly bound communicator slows every later step of the process by 8 %)
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1)
            try:
                model = CamLiRAFT(camliraft_cfg(n_iters=3))
                model.load_state_dict(state)
                model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).cuda().train()
                assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0)
                loss1, g1 = grads_of(model, 1, True)
                # a measured RCCL round trip of the SyncBatchNorm-sized payload (DESIGN section 7 quotes it)
                small = torch.zeros(385, device='cuda')
                for _ in range(5):
                    dist.all_reduce(small)
                torch.cuda.synchronize()
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                for _ in range(50):
                    dist.all_reduce(small)
                end.record()
                torch.cuda.synchronize()
                print('RCCL all-reduce of 385 floats, world 1: %.1f us per call' % (start.elapsed_time(end) / 50 * 1e3))
            finally:
                dist.destroy_process_group()
    finally:
        runtime.set_overlap(False)
        runtime.set_deferred_param_grads(False)
    assert abs(loss0 - loss1) <= 1e-5 * max(1.0, abs(loss0))
    assert g0.keys() == g1.keys()
    num = sum(((g0[n] - g1[n]).double() ** 2).sum().item() for n in g0) ** 0.5
    den = sum((g0[n].double() ** 2).sum().item() for n in g0) ** 0.5
    assert num / den < 1e-3, num / den


def _rccl_worker(rank, world, port, out_path):
    """One rank of a real multi-GPU run: its own device, RCCL over xGMI, SyncBatchNorm, parameter broadcast, the flat
    gradient all-reduce of bench.py -- each rank steps on its own sample."""
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    import bench
    from camliflow_amd.cores import CamLiRAFT, runtime
    dist.init_process_group('nccl', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2, freeze_bn=True)), scale=0.5)
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).cuda().train()
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
        full = synthetic_inputs(world, 128, 160, 4608, seed=3)
        shard = {k: v[rank:rank + 1].cuda() for k, v in full.items()}
        runtime.set_deferred_param_grads(True)
        with runtime.use_backend('hip'):
            model(shard)
            model.get_loss().backward()
            bench.allreduce_gradients(model, world)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_rccl_step_matches_two_sample_batch(tmp_path):
    """world = min(2, visible GPUs): on a multi-GPU box the data-parallel path runs on REAL RCCL between two devices (one process
    per GPU, as bench.py launches them) and the averaged gradients must equal one process stepping on the 2-sample batch --
    the GPU twin of tests/test_ddp_cpu.py (gloo).  BatchNorm statistics are frozen (freeze_bn) so that the two formulations
    are the same function.  Skipped where only one GPU is visible (the 1-rank test above covers the code path there)."""
    if torch.cuda.device_count() < 2:
        pytest.skip('one GPU visible: the two-rank RCCL step needs two')
    import torch.multiprocessing as mp
    from camliflow_amd.cores import CamLiRAFT, runtime
    out_path = str(tmp_path / 'rank0.pt')
    mp.spawn(_rccl_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2, freeze_bn=True)), scale=0.5).cuda().train()
    full = {k: v.cuda() for k, v in synthetic_inputs(2, 128, 160, 4608, seed=3).items()}
    with runtime.use_backend('hip'):
        model(full)
        model.get_loss().backward()
    want = {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}
    assert want.keys() == got.keys() and len(want) > 400
    num = sum(((got[n] - want[n]).double() ** 2).sum().item() for n in want) ** 0.5
    den = sum((want[n].double() ** 2).sum().item() for n in want) ** 0.5
    assert num / den < 1e-3, num / den

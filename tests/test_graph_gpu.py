"""HIP-graph replay of the whole training step (bench.py --graph): forward, losses, backward through every custom
adjoint on both lanes, clip and the optimizer captured once and replayed.  On the batch-1 configurations the step is
bound by host enqueue time and the replay is 1.6x (CamLiPWC) to 2.9x (KITTI shape, 32 iterations) faster; here the
replayed steps must leave the same parameters as the same number of eager steps."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _run(graphed, n_steps, warmup):
    import bench
    from camliflow_amd.cores import CamLiRAFT, runtime
    from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2)), scale=0.5).cuda().train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)
    batch = {k: v.cuda() for k, v in synthetic_inputs(1, 128, 160, 4608).items()}

    def step():
        model(batch)
        loss = model.get_loss()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        model.clear_metrics()
        return loss

    with runtime.use_backend('hip'):
        runtime.set_deferred_param_grads(True)
        runtime.set_overlap(True)
        try:
            # both runs start with ONE one-lane priming step: runtime.Lanes runs the first pass of a process on one stream
            # (two-lane start-up stall); GraphedStep takes it on the side stream its capture will use
            runtime.reset_lane_priming()
            if graphed:
                g = bench.GraphedStep(step, warmup=warmup)      # priming + `warmup` eager steps, then the capture
                for _ in range(n_steps - warmup):
                    loss = g()
            else:
                for _ in range(n_steps + 1):
                    loss = step()
            torch.cuda.synchronize()
        finally:
            runtime.set_overlap(False)
            runtime.set_deferred_param_grads(False)
    return float(loss.detach()), {n: p.detach().clone() for n, p in model.named_parameters()}


def test_graph_replay_matches_eager_steps():
    loss_e, params_e = _run(False, n_steps=5, warmup=0)
    loss_g, params_g = _run(True, n_steps=5, warmup=3)
    assert abs(loss_g - loss_e) <= 1e-3 * abs(loss_e) + 1e-5, (loss_g, loss_e)
    num = sum(((params_g[n].double() - params_e[n].double()) ** 2).sum() for n in params_e).sqrt().item()
    # the update itself is tiny (lr 1e-4, clipped): compare against the distance the eager run moved
    torch.manual_seed(0)
    from camliflow_amd.cores import CamLiRAFT
    from modelutils import camliraft_cfg, hashed_fill_
    start = {n: p.detach().cuda() for n, p in hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2)), scale=0.5).named_parameters()}
    moved = sum(((params_e[n].double() - start[n].double()) ** 2).sum() for n in params_e).sqrt().item()
    assert moved > 0
    assert num <= 2e-2 * moved, (num, moved)

"""No autograd graph outlives its training step.

The reference keeps the pass's cost-volume pyramids as module state until the next build overwrites them
(models/raft_core.py:41-68, camliraft_l_core.py:41-60).  On the HIP path those pyramids carry tokens of custom autograd nodes,
and through them the whole graph of the step back to the encoders' parameters: left in the modules they kept the PREVIOUS
step's graph alive while the next forward pass ran, torch then reused that step's AccumulateGrad nodes with the streams they
were created on ("The AccumulateGrad node's stream does not match ...", every bench run of rounds 3-5) -- and 250 MB of
point volumes per step stayed allocated a forward pass longer than needed.  Correlation2D / Correlation3D.release() now drops
them after the last lookup (cores/raft2d.py, raft3d.py).  tools/graph_leak_probe.py is the interactive form of this test."""
import gc
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _live_graph_tensors():
    gc.collect()
    found = []
    for obj in gc.get_objects():
        try:
            if isinstance(obj, torch.Tensor) and obj.grad_fn is not None:
                found.append((tuple(obj.shape), type(obj.grad_fn).__name__))
        except Exception:       # noqa: BLE001 -- objects that raise on attribute access are not tensors of ours
            continue
    return found


@pytest.mark.parametrize('two_lanes', [False, True], ids=['one_lane', 'two_lanes'])
def test_no_graph_survives_a_training_step(two_lanes):
    import bench
    from camliflow_amd.cores import CamLiRAFT, runtime
    from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2)), scale=0.5).cuda().train()
    optimizer = torch.optim.SGD(model.parameters(), lr=1e-4)
    batch = {k: v.cuda() for k, v in synthetic_inputs(1, 128, 160, 4608).items()}
    with runtime.use_backend('hip'):
        runtime.set_deferred_param_grads(True)
        runtime.set_overlap(two_lanes)
        try:
            runtime.reset_lane_priming()
            allocated = []
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter('always')
                for _ in range(4):
                    loss = bench.train_step(model, optimizer, batch)
                    torch.cuda.synchronize()
                    assert loss.grad_fn is None
                    live = _live_graph_tensors()
                    assert not live, 'tensors with a grad_fn alive after the step: %s' % live[:6]
                    allocated.append(torch.cuda.memory_allocated())
            stale = [str(w.message)[:80] for w in caught if 'AccumulateGrad' in str(w.message)]
            assert not stale, stale
            # steady state from the second step on (the first two-lane step allocates the side lane's buffers)
            assert max(allocated[2:]) - min(allocated[2:]) <= 8 << 20, allocated
        finally:
            runtime.set_overlap(False)
            runtime.set_deferred_param_grads(False)

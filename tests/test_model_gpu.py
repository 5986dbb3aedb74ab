"""CamLiRAFT end to end on the GPU: product path ('hip' backend, HIP kernels) against (a) the
torch-composed formulation on the same device and (b) the CPU run with oracle operators.
Tolerance: EPE2D / EPE3D difference <= 1e-4 (north star), fp32."""
import pytest
import torch

from modelutils import camliraft_cfg, hashed_fill_, oracle_boundary, synthetic_inputs

pytestmark = pytest.mark.gpu


def _to(inputs, device):
    return {k: v.to(device) for k, v in inputs.items()}


def _epe(a, b):
    return torch.linalg.norm(a - b, dim=1).mean().item()


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_camliraft_hip_vs_composed_vs_cpu_oracle(mode):
    from camliflow_amd.cores import CamLiRAFT, runtime
    torch.manual_seed(0)
    cfg = camliraft_cfg(n_iters=3)
    cpu_model = hashed_fill_(CamLiRAFT(cfg))
    gpu_model = CamLiRAFT(cfg)
    gpu_model.load_state_dict(cpu_model.state_dict())
    gpu_model.cuda()
    getattr(cpu_model, mode)()
    getattr(gpu_model, mode)()
    inputs = synthetic_inputs(1, 128, 160, 4608)

    with oracle_boundary():
        out_cpu = cpu_model(inputs)
        loss_cpu = cpu_model.get_loss().item()
    with runtime.use_backend('hip'):
        out_hip = gpu_model(_to(inputs, 'cuda'))
        loss_hip = gpu_model.get_loss()
        if mode == 'train':
            gpu_model.zero_grad()
            loss_hip.backward()
            grads_hip = {n: p.grad.clone() for n, p in gpu_model.named_parameters() if p.grad is not None}
    with runtime.use_backend('composed'):
        out_cmp = gpu_model(_to(inputs, 'cuda'))
        loss_cmp = gpu_model.get_loss()
        if mode == 'train':
            gpu_model.zero_grad()
            loss_cmp.backward()
            grads_cmp = {n: p.grad.clone() for n, p in gpu_model.named_parameters() if p.grad is not None}

    for key in ('flow_2d', 'flow_3d'):
        assert _epe(out_hip[key], out_cmp[key]) <= 1e-4, key
        assert _epe(out_hip[key].cpu(), out_cpu[key]) <= 1e-4, key
    assert abs(loss_hip.item() - loss_cmp.item()) <= 1e-4 * max(1.0, abs(loss_cmp.item()))
    assert abs(loss_hip.item() - loss_cpu) <= 1e-4 * max(1.0, abs(loss_cpu))
    if mode == 'train':
        assert grads_hip.keys() == grads_cmp.keys()
        worst = max(((grads_hip[n] - grads_cmp[n]).abs().max() / (grads_cmp[n].abs().max() + 1e-6)).item()
                    for n in grads_hip)
        assert worst < 2e-3, worst

"""Fused convex up-sampling (camli_convex_upsample_fwd/bwd): forward vs the C oracle and the torch
composition of the reference (utils.py:191-204), gradients vs autograd of that composition; fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(2, 68, 120, 8, 0.25), (1, 17, 30, 8, 0.25), (2, 36, 60, 4, 1.0), (1, 5, 70, 8, 1.0)], ids=str)
def test_convex_upsample(case, oracle_lib):
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.geometry import convex_upsample
    b, h, w, s, ms = case
    g = torch.Generator(device='cpu').manual_seed(h * w)
    flow = torch.randn(b, 2, h, w, generator=g).cuda().requires_grad_(True)
    mask = (torch.randn(b, 9 * s * s, h, w, generator=g) * 3).cuda().requires_grad_(True)
    gout = torch.randn(b, 2, h * s, w * s, generator=g).cuda()
    res = {}
    for backend in ('hip', 'composed'):
        flow.grad = mask.grad = None
        with runtime.use_backend(backend):
            out = convex_upsample(flow, mask, scale_factor=s, mask_scale=ms)
        out.backward(gout)
        res[backend] = (out.detach(), flow.grad.clone(), mask.grad.clone())
    want = oracle_lib.convex_upsample_fwd(flow.detach().cpu().numpy(), (mask.detach() * ms).cpu().numpy(), s)
    assert np.allclose(res['hip'][0].cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    assert torch.allclose(res['hip'][0], res['composed'][0], rtol=1e-5, atol=1e-5)
    assert torch.allclose(res['hip'][1], res['composed'][1], rtol=1e-4, atol=1e-4)
    assert torch.allclose(res['hip'][2], res['composed'][2], rtol=1e-4, atol=1e-5)

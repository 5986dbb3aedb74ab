"""camli_sk_* (selective-kernel fusion, full-size part) against oracle/glue.py (numpy in float64, forward and hand-written
adjoints, pinned on the reference's own SKFusion with autograd: tests/test_glue_oracle.py; models/clfm.py:170-213).  Round 4
compared these kernels with torch on the same GPU."""
import numpy as np
import pytest
import torch


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(8, 128, 68, 120), (2, 64, 2048), (3, 5, 7, 9), (1, 16, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('w_grad', [True, False], ids=['gate_differentiable', 'gate_constant'])
def test_pool_and_mix_vs_oracle(shape, w_grad):
    from camliflow_amd.csrc import fused
    from oracle import glue
    torch.manual_seed(sum(shape))
    a0 = torch.randn(*shape, device='cuda')
    b0 = torch.randn(*shape, device='cuda')
    lin = torch.nn.Linear(shape[1], 2 * shape[1], bias=False).cuda().requires_grad_(w_grad)
    gout = torch.randn(*shape, device='cuda')
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    state = fused.SkState()
    s = fused.sk_pool(a, b, state)
    w = torch.softmax(lin(s).reshape(shape[0], -1, 2), dim=-1)
    w.retain_grad()
    s.retain_grad()
    out = fused.sk_mix(a, b, w, state)
    out.backward(gout)
    an, bn, gn = a0.cpu().numpy(), b0.cpu().numpy(), gout.cpu().numpy()
    want_s = glue.sk_pool_fwd(an, bn)
    want_out = glue.sk_mix_fwd(an, bn, w.detach().cpu().numpy())
    # the gradient reaching s comes through the (torch) gate between the two kernels: taken from autograd
    gs = s.grad.cpu().numpy() if (w_grad or s.grad is not None) and s.grad is not None else np.zeros_like(want_s)
    want_ga, want_gb, want_gw = glue.sk_fuse_bwd(gn, an, bn, w.detach().cpu().numpy(), gs)
    assert torch.allclose(s.detach(), _t(want_s), rtol=1e-5, atol=1e-6)
    assert torch.allclose(out.detach(), _t(want_out), rtol=1e-5, atol=1e-6)
    assert torch.allclose(a.grad, _t(want_ga), rtol=1e-4, atol=1e-5)
    assert torch.allclose(b.grad, _t(want_gb), rtol=1e-4, atol=1e-5)
    assert (w.grad - _t(want_gw)).norm() <= 1e-4 * _t(want_gw).norm() + 1e-6


@pytest.mark.parametrize('fmt', ['nchw', 'ncm'])
def test_skfusion_module_hip_vs_composed(fmt):
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.fusion import SKFusion
    from modelutils import hashed_fill_
    torch.manual_seed(1)
    mod = hashed_fill_(SKFusion(128, 128, 128, fmt, None, reduction=2)).cuda()
    shape = (4, 128, 34, 60) if fmt == 'nchw' else (4, 128, 2048)
    x0, y0 = torch.randn(*shape, device='cuda'), torch.randn(*shape, device='cuda')
    gout = torch.randn(*shape, device='cuda')
    res = {}
    for backend in ('hip', 'composed'):
        x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
        mod.zero_grad()
        with runtime.use_backend(backend):
            out = mod(x, y)
        out.backward(gout)
        res[backend] = (out.detach(), x.grad, y.grad, [p.grad.clone() for p in mod.parameters()])
    a, b = res['hip'], res['composed']
    assert torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5)
    for u, v in zip([a[1], a[2]] + a[3], [b[1], b[2]] + b[3]):
        assert (u - v).norm() <= 2e-4 * v.norm() + 1e-6, ((u - v).norm() / v.norm()).item()


@pytest.mark.parametrize('dims', [(8, 128, 64), (3, 20, 7), (1, 256, 128), (8, 324, 162), (2, 512, 256), (1, 627, 313), (2, 1024, 512)],
                         ids=lambda d: 'B%d_C%d_R%d' % d)
def test_gate_vs_oracle(dims):
    from camliflow_amd.csrc import fused
    from oracle import glue
    b, c, r = dims
    torch.manual_seed(sum(dims))
    s = torch.randn(b, c, device='cuda').requires_grad_(True)
    wmid = (torch.randn(r, c, device='cuda') * c ** -0.5).requires_grad_(True)
    wout = (torch.randn(2 * c, r, device='cuda') * r ** -0.5).requires_grad_(True)
    gw = torch.randn(b, c, 2, device='cuda')
    w = fused.sk_gate(s, wmid, wout)
    w.backward(gw)
    sn, wm, wo = s.detach().cpu().numpy(), wmid.detach().cpu().numpy(), wout.detach().cpu().numpy()
    want = [glue.sk_gate_fwd(sn, wm, wo)] + list(glue.sk_gate_bwd(gw.cpu().numpy(), sn, wm, wo))
    for x, y in zip((w.detach(), s.grad, wmid.grad, wout.grad), want):
        assert torch.allclose(x, _t(y), rtol=1e-4, atol=1e-6), (x - _t(y)).abs().max()

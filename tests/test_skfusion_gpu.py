"""camli_sk_* (selective-kernel fusion, full-size part) against the torch formulation of
models/clfm.py:170-213: fp32, tolerances stated."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(8, 128, 68, 120), (2, 64, 2048), (3, 5, 7, 9), (1, 16, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('w_grad', [True, False], ids=['gate_differentiable', 'gate_constant'])
def test_pool_and_mix_vs_torch(shape, w_grad):
    from camliflow_amd.csrc import fused
    torch.manual_seed(sum(shape))
    a0 = torch.randn(*shape, device='cuda')
    b0 = torch.randn(*shape, device='cuda')
    lin = torch.nn.Linear(shape[1], 2 * shape[1], bias=False).cuda().requires_grad_(w_grad)
    gout = torch.randn(*shape, device='cuda')
    res = []
    for impl in ('hip', 'torch'):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        lin.zero_grad()
        if impl == 'hip':
            state = fused.SkState()
            s = fused.sk_pool(a, b, state)
            w = torch.softmax(lin(s).reshape(shape[0], -1, 2), dim=-1)
            out = fused.sk_mix(a, b, w, state)
        else:
            s = (a + b).flatten(2).mean(-1)
            w = torch.softmax(lin(s).reshape(shape[0], -1, 2), dim=-1)
            bshape = [shape[0], -1] + [1] * (len(shape) - 2)
            out = a * w[..., 0].reshape(bshape) + b * w[..., 1].reshape(bshape)
        out.backward(gout)
        res.append((s.detach(), out.detach(), a.grad, b.grad, lin.weight.grad.clone() if w_grad else None))
    (s1, o1, ga1, gb1, gl1), (s2, o2, ga2, gb2, gl2) = res
    assert torch.allclose(s1, s2, rtol=1e-5, atol=1e-6)
    assert torch.allclose(o1, o2, rtol=1e-5, atol=1e-6)
    assert torch.allclose(ga1, ga2, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gb1, gb2, rtol=1e-4, atol=1e-5)
    if w_grad:
        assert (gl1 - gl2).norm() <= 1e-4 * gl2.norm() + 1e-6


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
def test_gate_vs_torch(dims):
    from camliflow_amd.csrc import fused
    b, c, r = dims
    torch.manual_seed(sum(dims))
    s0 = torch.randn(b, c, device='cuda')
    wmid = (torch.randn(r, c, device='cuda') * c ** -0.5).requires_grad_(True)
    wout = (torch.randn(2 * c, r, device='cuda') * r ** -0.5).requires_grad_(True)
    gw = torch.randn(b, c, 2, device='cuda')
    res = []
    for impl in ('hip', 'torch'):
        s = s0.clone().requires_grad_(True)
        wmid.grad = wout.grad = None
        if impl == 'hip':
            w = fused.sk_gate(s, wmid, wout)
        else:
            w = torch.softmax(torch.sigmoid(torch.relu(s @ wmid.t()) @ wout.t()).reshape(b, c, 2), dim=-1)
        w.backward(gw)
        res.append((w.detach(), s.grad, wmid.grad.clone(), wout.grad.clone()))
    for x, y in zip(*res):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-6), (x - y).abs().max()

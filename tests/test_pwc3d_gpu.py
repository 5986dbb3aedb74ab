"""PointPWC learnable cost volume (PWC-style Correlation3D, models/camlipwc_l_core.py:39-106) on the HIP path:
the three kernels against the oracle and torch autograd, the module against its torch-composed twin (values,
input and parameter gradients incl. the gradient that reaches the warped cloud xyz2), and against the golden
recorded from the reference's own module (tests/golden/make_module_golden.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('case', [(2, 32, 300, 200, 16), (1, 192, 256, 256, 16), (2, 7, 50, 33, 4)], ids=str)
def test_pair_ksum_gather_wsum_vs_oracle_and_autograd(case, oracle_lib):
    from camliflow_amd.csrc import fused
    b, c, m, n, k = case
    rng = np.random.default_rng(n)
    a = rng.standard_normal((b, c, n)).astype(np.float32)
    bm = rng.standard_normal((b, c, m)).astype(np.float32)
    e = rng.standard_normal((b, c, n, k)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, k)).astype(np.int64)
    ta, tb, te = (dev(x).requires_grad_(True) for x in (a, bm, e))
    h1 = fused.pwc3d_pair(ta, tb, te, dev(idx), 0.1)
    assert np.array_equal(h1.detach().cpu().numpy(), oracle_lib.pwc3d_pair_fwd(a, bm, e, idx, 0.1))
    # composed twin in torch for the adjoints
    ra, rb, re = (dev(x).requires_grad_(True) for x in (a, bm, e))
    gathered = torch.gather(rb, 2, dev(idx).reshape(b, 1, n * k).expand(b, c, n * k)).view(b, c, n, k)
    ref = torch.nn.functional.leaky_relu(ra[..., None] + gathered + re, 0.1)
    g = dev(rng.standard_normal(h1.shape).astype(np.float32))
    h1.backward(g)
    ref.backward(g)
    for got, want in ((ta.grad, ra.grad), (tb.grad, rb.grad), (te.grad, re.grad)):
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)

    w = rng.standard_normal((b, c, n, k)).astype(np.float32)
    tw, th = dev(w).requires_grad_(True), h1.detach().clone().requires_grad_(True)
    out = fused.ksum(tw, th)
    assert np.allclose(out.detach().cpu().numpy(), oracle_lib.ksum_fwd(w, th.detach().cpu().numpy()), rtol=1e-5, atol=1e-5)
    g2 = dev(rng.standard_normal(out.shape).astype(np.float32))
    out.backward(g2)
    assert torch.allclose(tw.grad, g2[..., None] * th.detach(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(th.grad, g2[..., None] * tw.detach(), rtol=1e-6, atol=1e-6)

    feat = rng.standard_normal((b, c, m)).astype(np.float32)
    tw2, tf = dev(w).requires_grad_(True), dev(feat).requires_grad_(True)
    out = fused.gather_wsum(tw2, tf, dev(idx))
    assert np.allclose(out.detach().cpu().numpy(), oracle_lib.gather_wsum_fwd(w, feat, idx), rtol=1e-5, atol=1e-5)
    rw, rf = dev(w).requires_grad_(True), dev(feat).requires_grad_(True)
    ref = (rw * torch.gather(rf, 2, dev(idx).reshape(b, 1, n * k).expand(b, c, n * k)).view(b, c, n, k)).sum(-1)
    out.backward(g2)
    ref.backward(g2)
    assert torch.allclose(tw2.grad, rw.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(tf.grad, rf.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('channels', [32, 192])
def test_module_hip_vs_composed_with_live_coordinates(channels):
    """Correlation3D (PWC) under 'hip' (strict: nothing may fall back) vs the torch-composed formulation: output,
    feature gradients, the gradient reaching xyz2 (CamLiPWC warps it with a live flow) and all parameter gradients."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.pwc3d import Correlation3D
    from modelutils import hashed_fill_
    torch.manual_seed(0)
    mod = hashed_fill_(Correlation3D(channels, channels, 64)).cuda()
    b, n = 2, 300
    xyz1 = torch.rand(b, 3, n, device='cuda') * 4
    xyz2 = (xyz1 + torch.randn(b, 3, n, device='cuda') * 0.2).requires_grad_(True)
    f1 = torch.randn(b, channels, n, device='cuda', requires_grad=True)
    f2 = torch.randn(b, channels, n, device='cuda', requires_grad=True)
    g = torch.randn(b, 64, n, device='cuda')
    res = {}
    for backend in ('hip', 'composed'):
        mod.zero_grad()
        with runtime.use_backend(backend):
            runtime.set_strict(backend == 'hip')
            try:
                out = mod(xyz1, f1, xyz2, f2)
                grads = torch.autograd.grad(out, [f1, f2, xyz2] + list(mod.parameters()), g)
            finally:
                runtime.set_strict(False)
        res[backend] = (out.detach(), grads)
    (oh, gh), (oc, gc) = res['hip'], res['composed']
    assert (oh - oc).abs().max() <= 1e-4 * oc.abs().max() + 1e-5
    names = ['f1', 'f2', 'xyz2'] + [n_ for n_, _ in mod.named_parameters()]
    for name, x, y in zip(names, gh, gc):
        assert (x - y).norm() <= 2e-3 * y.norm() + 1e-5, (name, (x - y).norm().item(), y.norm().item())

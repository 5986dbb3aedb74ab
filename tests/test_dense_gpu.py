"""The round-3 dense kernels against oracle/dense.py -- the numpy restatement pinned on the reference's own modules
(tests/test_dense_oracle.py) -- instead of against torch on the same GPU (a kernel-level failure must localise):

  camli_corr3d_mlp_fwd/bwd            cost MLP + neighbour sum          models/camliraft_l_core.py:68-101
  camli_conv3x3_co2_fwd/bwd_*         two-channel 3x3 heads             models/raft_core.py:169-182
  camli_allpairs_build_fwd/bwd        all-pairs volume pyramid          models/raft_core.py:52-68
  camli_maxpool3x3s2_fwd/bwd          ResNet stem pooling               mmdet ResNet (call site raft_core.py:10-38)
  camli_bias_act_res_fwd, camli_bias_act_nhwc_fwd/bwd                   bottleneck epilogue, NCHW and channels-last

and against the committed golden tensors of the reference modules themselves (tests/golden/dense_*.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, want, rtol=1e-4, atol=1e-5, what=''):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.allclose(got, want, rtol=rtol, atol=atol * scale), (what, float(np.abs(got - want).max()), scale)


def _cost_mlp_modules(w1, b1, w2, b2):
    c1, c2 = torch.nn.Conv2d(4, w1.shape[0], 1).cuda(), torch.nn.Conv2d(w2.shape[1], w2.shape[0], 1).cuda()
    with torch.no_grad():
        c1.weight.copy_(dev(w1).view_as(c1.weight)), c1.bias.copy_(dev(b1))
        c2.weight.copy_(dev(w2).view_as(c2.weight)), c2.bias.copy_(dev(b2))
    return c1, c2


def _run_cost_mlp(lookup, w1, b1, w2, b2, levels, gout):
    from camliflow_amd.csrc import fused
    c1, c2 = _cost_mlp_modules(w1, b1, w2, b2)
    x = dev(lookup).requires_grad_(True)
    assert fused.corr3d_cost_mlp_supported(x, [c1, c2], levels)
    out = fused.corr3d_cost_mlp(x, c1, c2, levels)
    out.backward(dev(gout))
    return out, x.grad, c1.weight.grad.view(w1.shape), c1.bias.grad, c2.weight.grad.view(w2.shape), c2.bias.grad


def test_cost_mlp_kernel_vs_reference_golden(golden):
    g = golden('dense_cost_mlp')
    out, gx, gw1, gb1, gw2, gb2 = _run_cost_mlp(g['lookup'], g['w1'], g['b1'], g['w2'], g['b2'], int(g['levels']), g['gout'])
    _close(out, g['out'], what='out')
    _close(gx[:, 3], g['glookup'][:, 3], what='glookup[:,3]')          # the one channel this path differentiates
    assert float(gx[:, :3].abs().max()) == 0.0                          # zeros, not uninitialised memory
    for got, key in ((gw1, 'gw1'), (gb1, 'gb1'), (gw2, 'gw2'), (gb2, 'gb2')):
        _close(got, g[key], rtol=2e-4, atol=2e-5, what=key)


@pytest.mark.parametrize('shape', [(1, 8), (2, 64), (3, 200), (2, 1024)], ids=str)
def test_cost_mlp_kernel_vs_oracle(shape, oracle_dense):
    b, n = shape
    rng = np.random.default_rng(7 * b + n)
    lookup = rng.standard_normal((b, 4, n, 64), dtype=np.float32)
    w1, b1 = (rng.standard_normal((32, 4)) * 0.4).astype(np.float32), (rng.standard_normal(32) * 0.4).astype(np.float32)
    w2, b2 = (rng.standard_normal((32, 32)) * 0.4).astype(np.float32), (rng.standard_normal(32) * 0.4).astype(np.float32)
    gout = rng.standard_normal((b, 128, n), dtype=np.float32)
    out, gx, gw1, gb1, gw2, gb2 = _run_cost_mlp(lookup, w1, b1, w2, b2, 4, gout)
    _close(out, oracle_dense.cost_mlp_fwd(lookup, w1, b1, w2, b2, 4), what='out')
    ox, ow1, ob1, ow2, ob2 = oracle_dense.cost_mlp_bwd(gout, lookup, w1, b1, w2, b2, 4)
    # ReLU's derivative is discontinuous: among b*n*64*64 (column, unit) pairs a few pre-activations sit within one fp32
    # rounding of zero and take the other side -- isolated entries of the per-column gradient, at most 1 in 100,000
    bad = np.abs(gx[:, 3].cpu().numpy() - ox[:, 3]) > 1e-4 * max(1.0, np.abs(ox[:, 3]).max())
    assert bad.sum() <= ox[:, 3].size // 100000, bad.sum()
    for got, want, what in ((gw1, ow1, 'gw1'), (gb1, ob1, 'gb1'), (gw2, ow2, 'gw2'), (gb2, ob2, 'gb2')):
        _close(got, want, rtol=5e-4, atol=5e-5, what=what)


def _run_conv(x, w, b, gy):
    from camliflow_amd.csrc import fused
    xt, wt = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
    bt = dev(b).requires_grad_(True) if b is not None else None
    y = fused.conv3x3_co2(xt, wt, bt)
    grads = torch.autograd.grad(y, [xt, wt] + ([bt] if bt is not None else []), dev(gy))
    return (y,) + tuple(grads)


def test_conv3x3_co2_kernel_vs_reference_flow_head_golden(golden):
    g = golden('dense_flow_head')
    y, gx, gw, gb = _run_conv(g['x'], g['w'], g['b'], g['gy'])
    for got, key in ((y, 'y'), (gx, 'gx'), (gw, 'gw'), (gb, 'gb')):
        _close(got, g[key], what=key)


@pytest.mark.parametrize('case', [(2, 256, 17, 30), (1, 7, 1, 1), (3, 5, 2, 65), (2, 33, 17, 129), (1, 529, 9, 15)],
                         ids=lambda c: 'B%d_C%d_%dx%d' % c)
@pytest.mark.parametrize('with_bias', [True, False])
def test_conv3x3_co2_kernel_vs_oracle(case, with_bias, oracle_dense):
    b, c, h, w = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((b, c, h, w), dtype=np.float32)
    wt = (rng.standard_normal((2, c, 3, 3)) * (9 * c) ** -0.5).astype(np.float32)
    bias = rng.standard_normal(2).astype(np.float32) if with_bias else None
    gy = rng.standard_normal((b, 2, h, w), dtype=np.float32)
    res = _run_conv(x, wt, bias, gy)
    _close(res[0], oracle_dense.conv3x3_fwd(x, wt, bias), what='y')
    ogx, ogw, ogb = oracle_dense.conv3x3_bwd(gy, x, wt)
    _close(res[1], ogx, what='gx')
    _close(res[2], ogw, what='gw')
    if with_bias:
        _close(res[3], ogb, what='gb')


def _run_build(f1, f2, gpyr):
    from camliflow_amd.csrc import fused
    a, b = dev(f1).requires_grad_(True), dev(f2).requires_grad_(True)
    pyr = fused.allpairs_pyramid(a, b, 4)
    levels = [lvl.clone() for lvl in pyr.levels]
    pyr.grads = [dev(q) for q in gpyr]
    g1, g2 = torch.autograd.grad(pyr.token, [a, b], torch.zeros(1, device='cuda'))
    return levels, g1, g2


@pytest.mark.parametrize('tag', ['even', 'odd'])
def test_allpairs_build_kernel_vs_reference_golden(tag, golden):
    g = golden('dense_allpairs_' + tag)
    levels, g1, g2 = _run_build(g['f1'], g['f2'], [g['gpyr%d' % l] for l in range(4)])
    for l, lvl in enumerate(levels):
        _close(lvl, g['pyr%d' % l], rtol=1e-4, atol=1e-5, what='level %d' % l)
    _close(g1, g['gf1'], rtol=1e-4, atol=1e-4, what='gf1')
    _close(g2, g['gf2'], rtol=1e-4, atol=1e-4, what='gf2')


@pytest.mark.parametrize('shape', [(2, 256, 20, 24), (1, 64, 17, 30), (1, 256, 9, 13)], ids=str)
def test_allpairs_build_kernel_vs_oracle(shape, oracle_dense):
    b, c, h, w = shape
    rng = np.random.default_rng(h * w)
    f1, f2 = rng.standard_normal(shape, dtype=np.float32), rng.standard_normal(shape, dtype=np.float32)
    want = oracle_dense.allpairs_pyramid_fwd(f1, f2, 4)
    gpyr = [rng.standard_normal(p.shape, dtype=np.float32) for p in want]
    levels, g1, g2 = _run_build(f1, f2, gpyr)
    for l, lvl in enumerate(levels):
        _close(lvl, want[l], rtol=1e-4, atol=1e-5, what='level %d' % l)
    og1, og2 = oracle_dense.allpairs_pyramid_bwd(gpyr, f1, f2)
    _close(g1, og1, rtol=1e-4, atol=1e-4, what='gf1')
    _close(g2, og2, rtol=1e-4, atol=1e-4, what='gf2')


@pytest.mark.parametrize('shape', [(2, 8, 64, 96), (1, 3, 17, 33), (1, 2, 1, 1), (2, 4, 2, 300)], ids=str)
def test_maxpool_kernel_vs_oracle(shape, oracle_dense):
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(sum(shape))
    x = np.maximum(rng.standard_normal(shape, dtype=np.float32), 0.0)           # post-ReLU: many exact ties at zero
    want, arg = oracle_dense.maxpool3x3s2_fwd(x)
    gy = rng.standard_normal(want.shape, dtype=np.float32)
    xt = dev(x).requires_grad_(True)
    y = fused.maxpool3x3s2(xt)
    assert np.array_equal(y.detach().cpu().numpy(), want)
    gx = torch.autograd.grad(y, xt, dev(gy))[0].cpu().numpy()
    assert np.allclose(gx, oracle_dense.maxpool3x3s2_bwd(gy, arg, shape[-2:]), rtol=1e-6, atol=1e-6)   # first maximum in row-major order


def test_maxpool_and_epilogue_kernels_vs_golden(golden):
    from camliflow_amd.csrc import fused
    g = golden('dense_resnet_glue')
    xt = dev(g['pool_x']).requires_grad_(True)
    y = fused.maxpool3x3s2(xt)
    assert np.array_equal(y.detach().cpu().numpy(), g['pool_y'])
    _close(torch.autograd.grad(y, xt, dev(g['pool_gy']))[0], g['pool_gx'], rtol=1e-6, atol=1e-6, what='pool gx')
    for channels_last in (False, True):
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        c = dev(g['conv_out']).contiguous(memory_format=fmt).requires_grad_(True)
        r = dev(g['identity']).contiguous(memory_format=fmt).requires_grad_(True)
        bias = dev(g['bias']).requires_grad_(True)
        out = fused.bias_act_res(c * 1.0, bias, r, 'relu')
        gc, gr, gb = torch.autograd.grad(out, [c, r, bias], dev(g['gout']).contiguous(memory_format=fmt))
        _close(out, g['out'], rtol=1e-6, atol=1e-6, what='epilogue out')
        _close(gc, g['gconv'], rtol=1e-6, atol=1e-6, what='gconv')
        _close(gr, g['gconv'], rtol=1e-6, atol=1e-6, what='gidentity')
        _close(gb, g['gbias'], what='gbias')


@pytest.mark.parametrize('shape', [(2, 16, 12, 20), (1, 5, 3, 7), (3, 256, 17, 30), (1, 1024, 2, 3)], ids=str)
@pytest.mark.parametrize('channels_last', [False, True])
def test_epilogue_kernels_vs_oracle(shape, channels_last, oracle_dense):
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(sum(shape))
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    x, res = rng.standard_normal(shape, dtype=np.float32), rng.standard_normal(shape, dtype=np.float32)
    bias, gout = rng.standard_normal(shape[1]).astype(np.float32), rng.standard_normal(shape, dtype=np.float32)
    for relu in (True, False):
        for with_res in (True, False):
            xt = dev(x).contiguous(memory_format=fmt).requires_grad_(True)
            rt = dev(res).contiguous(memory_format=fmt).requires_grad_(True)
            bt = dev(bias).requires_grad_(True)
            act = 'relu' if relu else None
            out = fused.bias_act_res(xt * 1.0, bt, rt, act) if with_res else fused.bias_act(xt * 1.0, bt, act)
            grads = torch.autograd.grad(out, [xt, bt] + ([rt] if with_res else []), dev(gout).contiguous(memory_format=fmt))
            want = oracle_dense.bias_act_res_fwd(x, bias, res if with_res else None, relu)
            wgx, wgb = oracle_dense.bias_act_res_bwd(gout, want, relu)
            _close(out, want, rtol=1e-6, atol=1e-6, what='out')
            _close(grads[0], wgx, rtol=1e-6, atol=1e-6, what='gx')
            _close(grads[1], wgb, what='gbias')
            if with_res:
                _close(grads[2], wgx, rtol=1e-6, atol=1e-6, what='gres')

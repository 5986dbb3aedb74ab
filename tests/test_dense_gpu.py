"""The round-3 dense kernels against oracle/dense.py -- the numpy restatement pinned on the reference's own modules
(tests/test_dense_oracle.py) -- instead of against torch on the same GPU (a kernel-level failure must localise):

  camli_corr3d_mlp_fwd/bwd            cost MLP + neighbour sum          models/camliraft_l_core.py:68-101
  camli_conv3x3_co2_fwd/bwd_*         two-channel 3x3 heads             models/raft_core.py:169-182
  camli_allpairs_build_fwd/bwd        all-pairs volume pyramid          models/raft_core.py:52-68
  camli_maxpool3x3s2_fwd/bwd          ResNet stem pooling               mmdet ResNet (call site raft_core.py:10-38)
  camli_bias_act_res_fwd, camli_bias_act_nhwc_fwd/bwd                   bottleneck epilogue, NCHW and channels-last
  camli_allpairs_build_fwd/bwd (scale 1/C, KNN-averaged target features) point cost-volume pyramid  models/camliraft_l_core.py:51-60
  camli_gather_cl_fwd/bwd_sorted      channel-last batch_indexing       models/utils.py:85-104

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


# ---- point cost-volume pyramid build (camliraft_l_core.py:51-60) and channel-last batch_indexing (utils.py:85-104) ----
def _run_point_volumes(f1, f2, parents, gpyr):
    from camliflow_amd.csrc import fused
    a, b = dev(f1).requires_grad_(True), dev(f2).requires_grad_(True)
    pyr = fused.point_volume_pyramid(a, b, [dev(p) for p in parents])
    sum((p * dev(q)).sum() for p, q in zip(pyr, gpyr)).backward()
    return pyr, a.grad, b.grad


def test_point_volume_build_vs_reference_golden(golden):
    g = golden('dense_point_volume')
    parents = [g['parents%d' % lvl] for lvl in range(3)]
    pyr, gf1, gf2 = _run_point_volumes(g['f1'], g['f2'], parents, [g['gpyr%d' % lvl] for lvl in range(4)])
    for lvl, p in enumerate(pyr):
        _close(p, g['pyr%d' % lvl], what='level %d' % lvl)
    _close(gf1, g['gf1'], what='gf1')
    _close(gf2, g['gf2'], what='gf2')


def test_point_volume_module_build_matches_reference_golden(golden):
    """the module path: Correlation3D.build_cost_volume_pyramid computes the neighbour tables itself (camli_knn)."""
    from camliflow_amd.cores.raft3d import Correlation3D
    g = golden('dense_point_volume')
    corr = Correlation3D(out_channels=128, k=16).cuda()
    xyzs2 = [dev(g['xyz%d' % lvl]) for lvl in range(4)]
    f1, f2 = dev(g['f1']).requires_grad_(True), dev(g['f2']).requires_grad_(True)
    corr.build_cost_volume_pyramid(f1, f2, xyzs2, k=3)
    for lvl, p in enumerate(corr.cost_volume_pyramid):
        _close(p, g['pyr%d' % lvl], what='level %d' % lvl)
    sum((p * dev(g['gpyr%d' % lvl])).sum() for lvl, p in enumerate(corr.cost_volume_pyramid)).backward()
    _close(f1.grad, g['gf1'], what='gf1')
    _close(f2.grad, g['gf2'], what='gf2')


@pytest.mark.parametrize('shape', [(2, 128, 2048, (2048, 1024, 512, 256)), (1, 64, 300, (257, 130, 33)), (3, 20, 129, (129,))],
                         ids=lambda s: 'b%dc%dn%d' % s[:3])
def test_point_volume_build_vs_oracle(shape, oracle_dense):
    b, c, n, sizes = shape
    rng = np.random.default_rng(n)
    f1 = rng.standard_normal((b, c, n)).astype(np.float32)
    f2 = rng.standard_normal((b, c, sizes[0])).astype(np.float32)
    parents = [rng.integers(0, fine, size=(b, coarse, 3)) for fine, coarse in zip(sizes[:-1], sizes[1:])]
    gpyr = [rng.standard_normal((b, n, m)).astype(np.float32) for m in sizes]
    pyr, gf1, gf2 = _run_point_volumes(f1, f2, parents, gpyr)
    want = oracle_dense.point_volume_pyramid_fwd(f1, f2, parents)
    for lvl, (p, w) in enumerate(zip(pyr, want)):
        _close(p, w, what='level %d' % lvl)
    if n <= 300:        # the numpy adjoint scatters column by column: small cases only
        wf1, wf2 = oracle_dense.point_volume_pyramid_bwd(gpyr, f1, f2, parents)
        _close(gf1, wf1, what='gf1')
        _close(gf2, wf2, what='gf2')


def test_point_volume_build_full_size_adjoint_property():
    """<V(f1, f2), G> is bilinear in (f1, f2): <G, V> = <f1, dV/df1^T G> = <f2, dV/df2^T G> at the configs[2] size."""
    b, c, n, sizes = 2, 128, 2048, (2048, 1024, 512, 256)
    gen = torch.Generator(device='cuda').manual_seed(5)
    f1 = torch.randn(b, c, n, device='cuda', generator=gen)
    f2 = torch.randn(b, c, sizes[0], device='cuda', generator=gen)
    parents = [torch.randint(0, fine, (b, coarse, 3), device='cuda', generator=gen) for fine, coarse in zip(sizes[:-1], sizes[1:])]
    gpyr = [torch.randn(b, n, m, device='cuda', generator=gen).numpy(force=True) for m in sizes]
    pyr, gf1, gf2 = _run_point_volumes(f1.numpy(force=True), f2.numpy(force=True), [p.numpy(force=True) for p in parents], gpyr)
    total = sum((p.double() * dev(q).double()).sum() for p, q in zip(pyr, gpyr)).item()
    assert abs((f1.double() * gf1.double()).sum().item() - total) <= 1e-5 * abs(total) + 1e-3
    assert abs((f2.double() * gf2.double()).sum().item() - total) <= 1e-5 * abs(total) + 1e-3


def test_channel_last_gather_vs_reference_golden(golden):
    from camliflow_amd.cores.geometry import batch_indexing
    g = golden('dense_point_volume')
    for tag in ('cl', 'cl2'):
        data = dev(g[tag + '_data']).requires_grad_(True)
        out = batch_indexing(data, dev(g[tag + '_idx']), layout='channel_last')
        assert np.array_equal(out.detach().cpu().numpy(), g[tag + '_out'])
        out.backward(dev(g[tag + '_gout']))
        _close(data.grad, g[tag + '_gdata'], what=tag)


@pytest.mark.parametrize('case', [(2, 100, 1, (37,)), (3, 513, 7, (40, 3)), (2, 2048, 64, (1024, 16)), (1, 5, 4, (1,)),
                                  (2, 4096, 128, (33,))], ids=lambda c: 'b%dm%dc%d' % c[:3])
@pytest.mark.parametrize('rank2', [False, True], ids=['rows', 'flat'])
def test_channel_last_gather_vs_oracle(case, rank2, oracle_lib):
    """forward bit-exact; the adjoint sums a row's contributions in ascending position like the oracle: bit-exact too."""
    from camliflow_amd.csrc import fused
    b, m, c, ishape = case
    if rank2 and c != 1:
        pytest.skip('rank-2 data is the C = 1 form')
    rng = np.random.default_rng(m + c)
    data = rng.standard_normal((b, m) if rank2 else (b, m, c)).astype(np.float32)
    idx = rng.integers(0, m, size=(b,) + ishape)
    if m > 8:
        idx[0].flat[:5] = 3                      # repeated targets: the adjoint has to add
    t = dev(data).requires_grad_(True)
    out = fused.gather_rows(t, dev(idx))
    flat = idx.reshape(b, -1)
    want = oracle_lib.gather_cl(data, flat)
    assert out.shape == (b,) + ishape + (() if rank2 else (c,))
    assert np.array_equal(out.detach().cpu().numpy().reshape(want.shape), want)
    gout = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(dev(gout).view_as(out))
    assert np.array_equal(t.grad.cpu().numpy(), oracle_lib.scatter_add_cl(gout, flat, m))



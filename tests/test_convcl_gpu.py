"""Channels-last tap convolution on the fp32 matrix cores (csrc/hip/convcl.hip: camli_convcl_fwd / camli_convcl_wrw, round 5):
forward, data gradient (the same kernel on the negated taps) and weight gradient against oracle/dense.conv_taps_fwd / _bwd
(numpy, pinned on the reference's GRU2D convolutions and on torch's conv2d: tests/test_dense_oracle.py), the training path of
GRU2D's half-step convolutions (cores/blocks._CatConvCL) against the library path it replaces, and this repo's GRU2D against
the reference module's recorded output (tests/golden/dense_gru2d.npz, models/raft_core.py:110-140)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return dev(np.transpose(a, (0, 2, 3, 1)))


def _close(got, want, tol=2e-5, what=''):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape and np.abs(got - want).max() <= tol * scale, (what, float(np.abs(got - want).max()), scale)


# (B, C0, C1, Cout, H, W, kh, kw): partial pixel tiles, every pixel near a border, two inputs, 128 / 256 / 384 output channels,
# GRU2D's own shapes at a small image, a 3 x 3 and a 5 x 5 kernel through the same code (at most 32 taps)
FWD_CASES = [
    (1, 32, 0, 256, 7, 9, 1, 5), (2, 64, 0, 128, 20, 30, 5, 1), (3, 16, 32, 256, 17, 33, 5, 1), (2, 16, 16, 128, 16, 40, 1, 5),
    (1, 128, 128, 256, 17, 30, 1, 5), (1, 128, 128, 128, 17, 30, 5, 1), (2, 48, 0, 384, 5, 140, 3, 3), (1, 16, 0, 128, 9, 9, 5, 5),
    (1, 16, 0, 128, 1, 1, 1, 5),
    # the product's maps at the per-rank batch of configs[3] (4), KITTI's (1, 47 x 156) and batch 2: half-empty chips -> narrow tiles
    (4, 16, 16, 256, 68, 120, 1, 5), (1, 32, 0, 256, 47, 156, 5, 1), (2, 16, 0, 128, 68, 120, 1, 5),
]


@pytest.mark.parametrize('tiles', ['auto', 'wide', 'narrow'])
@pytest.mark.parametrize('case', FWD_CASES, ids=str)
def test_convcl_forward_vs_oracle(case, tiles, oracle_dense, monkeypatch):
    """``tiles``: the channel tile is chosen by problem size (r6: the narrow one where the pixel tiles alone leave CUs idle --
    every small case here); CAMLI_CONVCL_TILES forces the wide / narrow kernels onto the same shapes."""
    from camliflow_amd.csrc import fused
    if tiles != 'auto':
        monkeypatch.setenv('CAMLI_CONVCL_TILES', tiles)
    b, c0, c1, cout, h, w, kh, kw = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((b, c0 + c1, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, c0 + c1, kh, kw)) * (kh * kw * (c0 + c1)) ** -0.5).astype(np.float32)
    pad = (kh // 2, kw // 2)
    wp, _ = fused.convcl_pack(dev(wt))
    xs = [nhwc(x[:, :c0])] + ([nhwc(x[:, c0:])] if c1 else [])
    got = fused.convcl(xs, wp, fused.convcl_taps(kh, kw, *pad))
    _close(got.permute(0, 3, 1, 2), oracle_dense.conv_taps_fwd(x, wt, pad), what='forward')


def test_convcl_reads_channel_slices_and_splits_its_output(oracle_dense):
    """Inputs that are channel slices of wider NHWC tensors (pixel stride > channels) and an output split on a channel boundary:
    how the data gradient hands the two parts of cat([h, motion]) back."""
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(11)
    b, h, w = 2, 11, 23
    wide = rng.standard_normal((b, 96, h, w), dtype=np.float32)
    wt = (rng.standard_normal((256, 48, 1, 5)) * 0.1).astype(np.float32)
    wide_d = nhwc(wide)
    xs = [wide_d[..., 16:48], wide_d[..., 80:96]]
    x = np.concatenate([wide[:, 16:48], wide[:, 80:96]], axis=1)
    y0, y1 = fused.convcl(xs, fused.convcl_pack(dev(wt))[0], fused.convcl_taps(1, 5, 0, 2), split=128)
    want = oracle_dense.conv_taps_fwd(x, wt, (0, 2))
    _close(y0.permute(0, 3, 1, 2), want[:, :128], what='first output')
    _close(y1.permute(0, 3, 1, 2), want[:, 128:], what='second output')


BWD_CASES = [  # (B, C0, C1, Cout, H, W, kh, kw): Cin a multiple of 256, or ONE 128-multiple input with 256-multiple outputs
    (1, 256, 0, 256, 7, 9, 1, 5), (2, 128, 128, 128, 20, 30, 5, 1), (3, 256, 0, 256, 17, 33, 5, 1), (2, 384, 128, 128, 6, 40, 1, 5),
    (1, 128, 128, 256, 13, 21, 3, 3), (2, 128, 0, 512, 11, 19, 3, 3), (1, 128, 0, 256, 9, 40, 1, 5),
]


@pytest.mark.parametrize('case', BWD_CASES, ids=str)
def test_convcl_adjoints_vs_oracle(case, oracle_dense):
    from camliflow_amd.csrc import fused
    b, c0, c1, cout, h, w, kh, kw = case
    rng = np.random.default_rng(sum(case) + 1)
    x = rng.standard_normal((b, c0 + c1, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, c0 + c1, kh, kw)) * (kh * kw * (c0 + c1)) ** -0.5).astype(np.float32)
    gy = rng.standard_normal((b, cout, h, w), dtype=np.float32)
    pad = (kh // 2, kw // 2)
    want_gx, want_gw = oracle_dense.conv_taps_bwd(gy, x, wt, pad)
    _, wpt = fused.convcl_pack(dev(wt))
    gx = fused.convcl([nhwc(gy)], wpt, fused.convcl_taps(kh, kw, *pad, negate=True))
    _close(gx.permute(0, 3, 1, 2), want_gx, what='data gradient')
    xs = [nhwc(x[:, :c0])] + ([nhwc(x[:, c0:])] if c1 else [])
    gw = fused.convcl_wrw(xs, nhwc(gy), fused.convcl_taps(kh, kw, *pad), (kh, kw))
    _close(gw, want_gw, tol=5e-5, what='weight gradient')
    # accumulate into an existing gradient; run-to-run bit-identical (split K with an ordered reduction, no atomics)
    again = fused.convcl_wrw(xs, nhwc(gy), fused.convcl_taps(kh, kw, *pad), (kh, kw))
    assert torch.equal(gw, again)
    acc = gw.clone()
    fused.convcl_wrw(xs, nhwc(gy), fused.convcl_taps(kh, kw, *pad), (kh, kw), out=acc)
    _close(acc, 2 * want_gw, tol=5e-5, what='accumulated weight gradient')


@pytest.mark.parametrize('vertical', [False, True])
def test_gru_half_step_convolution_own_kernels_vs_library(vertical, monkeypatch):
    """cores/blocks._CatConvCL (cat([h, motion]) -> 1x5 / 5x1 convolution, GRU2D's shapes): output, both input gradients and the
    weight gradient on this repo's kernels against the library path (CAMLI_CONVCL=0) they replace."""
    from camliflow_amd.cores import blocks, runtime
    runtime.set_backend('hip')
    torch.manual_seed(3)
    b, hd, h, w = 2, 128, 21, 37
    shape = (5, 1) if vertical else (1, 5)
    pad = (2, 0) if vertical else (0, 2)
    res = {}
    for own in (True, False):
        monkeypatch.setattr(blocks, '_CONVCL', own)
        hh = torch.randn(b, hd, h, w, device='cuda', requires_grad=True)
        x = torch.randn(b, hd, h, w, device='cuda', requires_grad=True)
        wt = (torch.randn((2 * hd, 2 * hd) + shape, device='cuda') * 0.03).requires_grad_()
        torch.manual_seed(4)
        y = blocks.cat_conv_cl([hh, x], wt, pad)
        g = torch.randn_like(y)
        y.backward(g)
        res[own] = (y.detach(), hh.grad, x.grad, wt.grad)
        torch.manual_seed(3)
    for a, bb, name in zip(res[True], res[False], ('output', 'gradient of h', 'gradient of motion', 'weight gradient')):
        scale = max(1.0, float(bb.abs().max()))
        assert float((a - bb).abs().max()) <= 3e-5 * scale, name


WINO1D_CASES = [  # (B, C0, C1, Cout, H, W, axis): lines that are not multiples of 4, a single tile per line, two inputs, both axes
    (1, 32, 0, 128, 7, 9, 0), (2, 64, 0, 256, 20, 30, 1), (3, 16, 32, 256, 17, 33, 1), (2, 128, 128, 256, 6, 13, 0),
    (1, 128, 128, 128, 17, 30, 1), (1, 48, 0, 128, 1, 3, 0), (1, 48, 0, 128, 3, 1, 1), (2, 128, 128, 256, 68, 120, 0),
]


@pytest.mark.parametrize('case', WINO1D_CASES, ids=str)
def test_wino1d_conv_vs_oracle(case, oracle_dense):
    """camli_wino1d_conv (1-D Winograd F(4,5) of a 1x5 / 5x1 convolution, csrc/hip/wino1d.hip) against the numpy oracle, forward
    weights and -- through the transposed packing with reversed taps -- the data gradient.  Bound 5e-5 x max(1, max |want|): the
    transforms carry factors up to 21/4 and 8 (measured 8e-6 max abs at 256 channels on unit-variance data)."""
    from camliflow_amd.csrc import fused
    b, c0, c1, cout, h, w, axis = case
    kh, kw = (1, 5) if axis == 0 else (5, 1)
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((b, c0 + c1, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, c0 + c1, kh, kw)) * (5 * (c0 + c1)) ** -0.5).astype(np.float32)
    pad = (kh // 2, kw // 2)
    wp, wpt = fused.convcl_pack(dev(wt))
    xs = [nhwc(x[:, :c0])] + ([nhwc(x[:, c0:])] if c1 else [])
    got = fused.wino1d_conv(xs, fused.wino1d_weights(wp, False), axis)
    _close(got.permute(0, 3, 1, 2), oracle_dense.conv_taps_fwd(x, wt, pad), tol=5e-5, what='forward')
    if (c0 + c1) % 128 == 0:          # the data gradient's output width = the forward's input width
        gy = rng.standard_normal((b, cout, h, w), dtype=np.float32)
        base = rng.standard_normal((b, h, w, c0 + c1), dtype=np.float32)
        half = (c0 + c1) // 2
        g0, g1 = dev(base[..., :half]), dev(base[..., half:])
        fused.wino1d_conv([nhwc(gy)], fused.wino1d_weights(wpt, True), axis, split=half, out=(g0, g1), accumulate=(False, True))
        want_gx, _ = oracle_dense.conv_taps_bwd(gy, x, wt, pad)
        want = np.transpose(want_gx, (0, 2, 3, 1))
        _close(g0, want[..., :half], tol=5e-5, what='data gradient, first output (written)')
        _close(g1, want[..., half:] + base[..., half:], tol=5e-5, what='data gradient, second output (accumulated)')


@pytest.mark.parametrize('case', [(1, 128, 128, 256, 7, 9, 0), (2, 256, 0, 128, 20, 30, 1), (2, 128, 128, 128, 17, 33, 1), (1, 256, 0, 256, 6, 13, 0),
                                  (2, 128, 128, 256, 68, 120, 1)], ids=str)
def test_wino1d_weight_gradient_vs_oracle(case, oracle_dense):
    """camli_wino1d_wrw: the weight gradient of a 1x5 / 5x1 convolution contracted in the Winograd domain (the 8 planes laid end
    to end through the weight-gradient core of wrwcl.h, K splits aligned with the planes), written and accumulated, repeatable."""
    from camliflow_amd.csrc import fused
    b, c0, c1, cout, h, w, axis = case
    kh, kw = (1, 5) if axis == 0 else (5, 1)
    rng = np.random.default_rng(sum(case) + 5)
    x = rng.standard_normal((b, c0 + c1, h, w), dtype=np.float32)
    gy = rng.standard_normal((b, cout, h, w), dtype=np.float32)
    wt = np.zeros((cout, c0 + c1, kh, kw), np.float32)
    _, want = oracle_dense.conv_taps_bwd(gy, x, wt, (kh // 2, kw // 2))
    xs = [nhwc(x[:, :c0])] + ([nhwc(x[:, c0:])] if c1 else [])
    got = fused.wino1d_wrw(xs, nhwc(gy), axis)
    _close(got, want, tol=5e-5, what='weight gradient')
    assert torch.equal(got, fused.wino1d_wrw(xs, nhwc(gy), axis)), 'fixed summation order'
    base = rng.standard_normal(want.shape).astype(np.float32)
    acc = dev(base)
    fused.wino1d_wrw(xs, nhwc(gy), axis, out=acc)
    _close(acc, base + want, tol=5e-5, what='accumulated')


@pytest.mark.parametrize('wino', [True, False], ids=['winograd', 'taps'])
@pytest.mark.parametrize('tiles', ['narrow', 'wide'])
def test_gru2d_update_as_one_channels_last_node_vs_reference_module(golden, tiles, wino, monkeypatch):
    """cores/raft2d.GRU2D on the product path (fused._GRU2DStepCL: convolutions with the gate arithmetic in their epilogues, both
    adjoints) against what the REFERENCE's GRU2D recorded at the product's widths -- two updates, the new hidden state, both
    input gradients, fingerprints of all twelve parameter gradients (tests/golden/dense_gru2d_wide.npz from
    tests/golden/make_dense_golden.py, models/raft_core.py:110-140; weights name-hashed on both sides)."""
    import zlib
    from modelutils import hashed_fill_
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.raft2d import GRU2D
    runtime.set_backend('hip')
    monkeypatch.setenv('CAMLI_CONVCL_TILES', tiles)      # the gates / blend epilogues on the 256 | 128-channel tiles and on their halves
    from camliflow_amd.csrc import fused
    monkeypatch.setattr(fused, '_GRU_WINO', wino)        # r6: the half-step convolutions as 1-D Winograd F(4,5), or as tap convolutions
    g = golden('dense_gru2d_wide')
    gru = hashed_fill_(GRU2D(hidden_dim=128, input_dim=256)).cuda()
    h0 = dev(g['h0']).requires_grad_()
    x = dev(g['x']).requires_grad_()
    context, motion = x[:, :128], x[:, 128:]       # GRU2D.prepare: the first channels of x are the per-pass context
    state = gru.prepare(context)
    assert 'cl1' in state and 'cl2' in state
    from camliflow_amd.cores import runtime as rt
    rt.set_census(True)
    rt.reset_census()
    out = gru.step(gru.step(h0, motion, state), motion, state)
    _close(out, g['out'], tol=2e-5, what='new hidden state')
    out.backward(dev(g['gout']))
    census = rt.census()['fused']
    rt.set_census(False)
    assert (census.get('camli_wino1d_gru_gates', 0), census.get('camli_convcl_gru_gates', 0)) == ((4, 0) if wino else (0, 4)), census
    assert census.get('camli_wino1d_conv', 0) == (8 if wino else 0) and census.get('camli_wino1d_wrw', 0) == (8 if wino else 0)
    _close(h0.grad, g['gh0'], tol=5e-5, what='gradient of h')
    _close(x.grad, g['gx'], tol=5e-5, what='gradient of x')
    for name, p_ in gru.named_parameters():
        d = torch.randn(p_.shape, generator=torch.Generator().manual_seed(zlib.crc32(('dir.' + name).encode())))
        fp = np.array([float(p_.grad.double().norm()), float((p_.grad.double().cpu() * d.double()).sum())])
        want = g['fp_' + name]
        assert abs(fp[0] - want[0]) <= 1e-4 * want[0] and abs(fp[1] - want[1]) <= 2e-4 * want[0], (name, fp, want)


@pytest.mark.parametrize('shape', [(2, 19, 35), (1, 68, 120), (2, 16, 64)], ids=str)      # (2, 16, 64): the weight gradients contract the forward's kept transforms
def test_gru2d_update_one_node_vs_per_convolution_nodes(shape, monkeypatch):
    """The one-node channels-last update (CAMLI_GRU_CL, default) against the per-convolution formulation it replaces
    (cat -> library convolution -> gate kernels): new hidden state and every gradient (h, motion, context, the six weights and
    biases)."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.raft2d import GRU2D
    runtime.set_backend('hip')
    b, hh, ww = shape
    torch.manual_seed(7)
    gru = GRU2D(hidden_dim=128, input_dim=256).cuda()
    h0 = torch.tanh(torch.randn(b, 128, hh, ww, device='cuda'))
    context = torch.randn(b, 128, hh, ww, device='cuda')
    motion = torch.randn(b, 128, hh, ww, device='cuda')
    gout = torch.randn(b, 128, hh, ww, device='cuda')
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('CAMLI_GRU_CL', mode)
        monkeypatch.setenv('CAMLI_CONVCL', mode)
        from camliflow_amd.cores import blocks
        monkeypatch.setattr(blocks, '_CONVCL', mode == '1')
        for p_ in gru.parameters():
            p_.grad = None
        h, c, m = (t.clone().requires_grad_() for t in (h0, context, motion))
        state = gru.prepare(c)
        assert ('cl1' in state) == (mode == '1')
        out = gru.step(gru.step(h, m, state), m, state)         # two updates: gradients accumulate through the shared state
        out.backward(gout)
        res[mode] = [out.detach(), h.grad, c.grad, m.grad] + [p_.grad.clone() for p_ in gru.parameters()]
    names = ['output', 'gradient of h', 'gradient of context', 'gradient of motion'] + [n for n, _ in gru.named_parameters()]
    for a, bb, name in zip(res['1'], res['0'], names):
        scale = max(1.0, float(bb.abs().max()))
        assert float((a - bb).abs().max()) <= 1e-4 * scale, (name, float((a - bb).abs().max()), scale)


def test_gru2d_weight_gradients_from_the_kept_transforms(monkeypatch):
    """r6: the forward keeps the transformed inputs V of its four convolutions and camli_wino1d_wrw contracts them as they lie
    (v_in) instead of transforming [h | m] / [r h | m] again -- same weight gradients as with CAMLI_GRU_KEEP_V=0 up to the
    summation order of the K splits (the kept planes have no padding rows, so the split differs), and the shape below is one
    camli_wino1d_wrw_reuse accepts on both axes."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.raft2d import GRU2D
    from camliflow_amd.csrc import _lib, fused
    runtime.set_backend('hip')
    b, hh, ww = 2, 16, 64
    lib = _lib.load()
    for axis in (0, 1):
        for cout in (256, 128):
            assert lib.camli_wino1d_wrw_reuse(b, hh, ww, 256, cout, axis) == 1
    assert lib.camli_wino1d_wrw_reuse(1, 47, 156, 256, 256, 0) == 0          # 47 x 39 tiles: no 16-row K split divides the plane
    torch.manual_seed(11)
    gru = GRU2D(hidden_dim=128, input_dim=256).cuda()
    h0 = torch.tanh(torch.randn(b, 128, hh, ww, device='cuda'))
    context, motion, gout = (torch.randn(b, 128, hh, ww, device='cuda') for _ in range(3))
    res = {}
    for keep in (True, False):
        monkeypatch.setattr(fused, '_GRU_KEEP_V', keep)
        for p_ in gru.parameters():
            p_.grad = None
        h, m = h0.clone().requires_grad_(), motion.clone().requires_grad_()
        runtime.set_census(True)
        runtime.reset_census()
        state = gru.prepare(context)
        out = gru.step(gru.step(h, m, state), m, state)
        out.backward(gout)
        census = runtime.census()['fused']
        runtime.set_census(False)
        assert census.get('camli_wino1d_wrw', 0) == 8, census
        res[keep] = [out.detach(), h.grad, m.grad] + [p_.grad.clone() for p_ in gru.parameters()]
    for a, bb in zip(res[True][:3], res[False][:3]):
        # the forward and the data gradients do not depend on where V lives (not torch.equal: the context terms come from a
        # library convolution in each pass, and which algorithm it picks is not fixed from call to call)
        assert float((a - bb).abs().max()) <= 1e-5 * max(1.0, float(bb.abs().max()))
    for a, bb, (name, _) in zip(res[True][3:], res[False][3:], gru.named_parameters()):
        scale = max(1.0, float(bb.abs().max()))
        assert float((a - bb).abs().max()) <= 2e-5 * scale, name

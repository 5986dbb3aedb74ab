"""Winograd F(M x M, 3x3) convolution, M = 2 | 4, on the fp32 matrix cores (csrc/hip/winograd.hip: camli_wino_weights /
camli_wino_conv3x3 / camli_wino_wrw, round 6) -- the update block's 3x3 convolutions (models/raft_core.py:148-151, 173, 188):
forward, data gradient and weight gradient against oracle/dense.conv_taps_fwd / _bwd (numpy fp64 accumulation, pinned on the
reference's convolutions and on torch's conv2d: tests/test_dense_oracle.py).

Tolerance.  The Winograd form is not the direct form's summation order, so equality is not on offer.  With unit-variance
inputs and weights scaled to unit-variance outputs:
  tile 2  measured 1.2e-6 max abs at 256 input channels -- BELOW the direct fp32 fmaf chain's own 4.2e-6
          (profiles/r06a_winograd_microbench.txt).  Bound: 2e-5 x max(1, max |want|), the one tests/test_convcl_gpu.py uses for
          the direct kernels.
  tile 4  the interpolation points 0, +-1, +-2, inf put factors up to 8 and 1/24 into the transforms: measured ~4e-5 max abs /
          3e-6 relative L2 on the same data (profiles/r06_experiments.txt item 9).  Bound: 1e-4 x max(1, max |want|) and, for
          the gradients, 2e-4 (they contract over thousands of tiles)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {2: 2e-5, 4: 1e-4}
TOL_W = {2: 5e-5, 4: 2e-4}
TILES = [2, 4]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, want, tol=2e-5, what=''):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape and np.abs(got - want).max() <= tol * scale, (what, float(np.abs(got - want).max()), scale)


# (B, Cin, Cout, H, W): odd sizes (partial tiles at the right / bottom border, padded tile columns), a single tile row, Cout
# on each of the three GEMM tiles (<= 128, <= 192, 256 and beyond), Cout not a multiple of 4 / 16 (126), Cin = 192 (12 K steps),
# the product's own channel pairs at a small image, odd channel counts on odd planes (CamLiPWC's 629-channel estimator input:
# image stride not a multiple of 4 floats -> the 4-byte paths of the transforms)
CASES = [
    (2, 48, 52, 13, 21), (1, 64, 192, 16, 24), (3, 128, 126, 9, 40), (1, 256, 192, 5, 7), (1, 96, 256, 1, 9), (2, 128, 512, 6, 10),
    (1, 256, 126, 17, 30), (1, 192, 256, 17, 30), (1, 128, 256, 2, 2), (1, 112, 320, 7, 5), (1, 629, 128, 9, 15), (2, 101, 99, 3, 5),
]


def _case_data(case, seed=0):
    b, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + seed)
    x = rng.standard_normal((b, cin, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * (9 * cin) ** -0.5).astype(np.float32)
    return x, wt, rng


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', CASES, ids=str)
def test_wino_forward_vs_oracle(case, tile, oracle_dense):
    from camliflow_amd.csrc import fused
    x, wt, _ = _case_data(case)
    w_d = dev(wt)
    got = fused.wino_conv3x3(dev(x), fused.wino_transformed_weights(w_d, False, tile), wt.shape[0])
    _close(got, oracle_dense.conv_taps_fwd(x, wt, (1, 1)), tol=TOL[tile], what='forward')


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', CASES, ids=str)
def test_wino_data_gradient_vs_oracle(case, tile, oracle_dense):
    from camliflow_amd.csrc import fused
    x, wt, rng = _case_data(case, 1)
    gy = rng.standard_normal((case[0], case[2], case[3], case[4]), dtype=np.float32)
    w_d = dev(wt)
    got = fused.wino_conv3x3(dev(gy), fused.wino_transformed_weights(w_d, True, tile), wt.shape[1])
    want_gx, _ = oracle_dense.conv_taps_bwd(gy, x, wt, (1, 1))
    _close(got, want_gx, tol=TOL[tile], what='data gradient')


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', CASES + [(2, 256, 192, 13, 22), (1, 40, 24, 9, 11), (4, 128, 256, 20, 28)], ids=str)
def test_wino_weight_gradient_vs_oracle(case, tile, oracle_dense):
    """camli_wino_wrw: the contraction over the tiles in the transform domain, either operand on the row side (256 -> 192 and
    256 -> 126 keep the input channels there, 128 -> 256 the output channels), odd images, several K splits."""
    from camliflow_amd.csrc import fused
    x, wt, rng = _case_data(case, 3)
    gy = rng.standard_normal((case[0], case[2], case[3], case[4]), dtype=np.float32)
    got = fused.wino_wrw(dev(x), dev(gy), tile=tile)
    _, want_gw = oracle_dense.conv_taps_bwd(gy, x, wt, (1, 1))
    _close(got, want_gw, tol=TOL_W[tile], what='weight gradient')


@pytest.mark.parametrize('tile', TILES)
def test_wino_weight_gradient_mask_accumulate_slices_and_repeatability(tile, oracle_dense):
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(9)
    b, cin, cout, h, w = 2, 96, 128, 11, 20
    wide = rng.standard_normal((b, cin + 32, h, w), dtype=np.float32)
    x = np.ascontiguousarray(wide[:, 16:16 + cin])
    gy = rng.standard_normal((b, cout, h, w), dtype=np.float32)
    mask = rng.standard_normal((b, cout, h, w)).astype(np.float32)
    wt = np.zeros((cout, cin, 3, 3), np.float32)
    _, want = oracle_dense.conv_taps_bwd(np.where(mask > 0, gy, 0).astype(np.float32), x, wt, (1, 1))
    wide_d = dev(wide)
    bits = fused.wino_pack_bits(dev(mask) > 0)
    got = fused.wino_wrw(wide_d[:, 16:16 + cin], dev(gy), bits=bits, tile=tile)
    _close(got, want, tol=TOL_W[tile], what='masked, sliced input')
    again = fused.wino_wrw(wide_d[:, 16:16 + cin], dev(gy), bits=bits, tile=tile)
    assert torch.equal(got, again), 'the split contraction is summed in a fixed order'
    base = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    acc = dev(base)
    gb = torch.zeros(cout, device='cuda')
    fused.wino_wrw(dev(x), dev(gy), bits=bits, out=acc, gbias=gb, tile=tile)
    _close(acc, base + want, tol=TOL_W[tile], what='accumulate')
    _close(gb, np.where(mask > 0, gy, 0).sum((0, 2, 3)).astype(np.float32), tol=TOL_W[tile], what='bias gradient from the tile sums')


@pytest.mark.parametrize('tile', TILES)
def test_wino_epilogue_bias_relu_mask_accumulate_and_slices(tile, oracle_dense):
    """The transforms' fused forms: bias + ReLU on the way out (and the activation bits), += into an existing tensor, the ReLU
    adjoint's bits on the way in, input / output that are channel slices of wider NCHW tensors.  W = 20: 2.5 mask bytes per
    row (a partial last byte)."""
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(5)
    b, cin, cout, h, w = 2, 64, 128, 11, 20
    wide = rng.standard_normal((b, cin + 32, h, w), dtype=np.float32)
    x = wide[:, 16:16 + cin]
    wt = (rng.standard_normal((cout, cin, 3, 3)) * (9 * cin) ** -0.5).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    mask = rng.standard_normal((b, cin, h, w)).astype(np.float32)
    w_d = dev(wt)
    u = fused.wino_transformed_weights(w_d, False, tile)
    tol = TOL[tile]
    wide_d = dev(wide)
    want = oracle_dense.conv_taps_fwd(x, wt, (1, 1)) + bias[None, :, None, None]
    got = fused.wino_conv3x3(wide_d[:, 16:16 + cin], u, cout, bias=dev(bias), act='relu')
    _close(got, np.maximum(want, 0), tol=tol, what='bias + relu, sliced input')
    out_wide = torch.full((b, cout + 8, h, w), 7.0, device='cuda')
    fused.wino_conv3x3(wide_d[:, 16:16 + cin], u, cout, bias=dev(bias), out=out_wide[:, 4:4 + cout])
    _close(out_wide[:, 4:4 + cout], want, tol=tol, what='sliced output')
    assert float(out_wide[:, :4].min()) == 7.0 and float(out_wide[:, 4 + cout:].max()) == 7.0
    base = rng.standard_normal((b, cout, h, w)).astype(np.float32)
    acc = dev(base)
    fused.wino_conv3x3(dev(np.ascontiguousarray(x)), u, cout, out=acc, accumulate=True)
    _close(acc, base + oracle_dense.conv_taps_fwd(x, wt, (1, 1)), tol=tol, what='accumulate')
    got = fused.wino_conv3x3(dev(np.ascontiguousarray(x)), u, cout, bits=fused.wino_pack_bits(dev(mask) > 0))
    _close(got, oracle_dense.conv_taps_fwd(np.where(mask > 0, x, 0).astype(np.float32), wt, (1, 1)), tol=tol, what='masked input')
    # the activation bits the output transform leaves behind = (pre-activation > 0), in the format the adjoint's transforms read
    bits = fused.wino_mask_bits(b, cout, h, w, 'cuda')
    fused.wino_conv3x3(wide_d[:, 16:16 + cin], u, cout, bias=dev(bias), act='relu', bits_out=bits)
    # (a pre-activation within rounding distance of zero may fall on either side: none of the 56,320 does here for tile 2; tile 4
    # is compared on the elements that are not)
    safe = wino_safe = fused.wino_pack_bits(dev(np.abs(want) > 1e-3))
    assert torch.equal(bits & safe, fused.wino_pack_bits(dev(want) > 0) & wino_safe)


@pytest.mark.parametrize('case', [(2, 128, 192, 13, 21), (1, 256, 126, 17, 30), (2, 128, 256, 8, 12)], ids=str)
def test_wino_autograd_node_vs_oracle(case, oracle_dense):
    """fused.conv3x3_wino as the cores call it (cores/blocks.conv_bias_act): output, input gradient, weight gradient."""
    from camliflow_amd.csrc import fused
    x, wt, rng = _case_data(case, 2)
    gy = rng.standard_normal((case[0], case[2], case[3], case[4]), dtype=np.float32)
    x_d, w_d = dev(x).requires_grad_(True), dev(wt).requires_grad_(True)
    y = fused.conv3x3_wino(x_d, w_d)
    y.backward(dev(gy))
    want_gx, want_gw = oracle_dense.conv_taps_bwd(gy, x, wt, (1, 1))
    tile = fused._WINO_TILE
    _close(y, oracle_dense.conv_taps_fwd(x, wt, (1, 1)), tol=TOL[tile], what='forward')
    _close(x_d.grad, want_gx, tol=TOL[tile], what='input gradient')
    _close(w_d.grad, want_gw, tol=TOL_W[tile], what='weight gradient')


def test_wino_weights_follow_the_parameter():
    """The transformed weights are cached per value of the weight tensor: an in-place update (an optimiser step) must be seen."""
    from camliflow_amd.csrc import fused
    torch.manual_seed(0)
    w = torch.randn(128, 96, 3, 3, device='cuda') * 0.05
    x = torch.randn(1, 96, 6, 8, device='cuda')
    y0 = fused.conv3x3_wino(x, w)
    assert fused.wino_transformed_weights(w, False) is fused.wino_transformed_weights(w, False)
    assert fused.wino_transformed_weights(w, False, 2) is not fused.wino_transformed_weights(w, False, 4)
    w.mul_(2.0)
    y1 = fused.conv3x3_wino(x, w)
    torch.testing.assert_close(y1, 2 * y0, rtol=1e-4, atol=1e-4)


def test_update_block_convolutions_take_the_winograd_path():
    """MotionEncoder2D.conv_c2 / conv, FlowHead2D.conv1 and the mask head's 3x3 (models/raft_core.py:148-151,173,188) are
    eligible; conv_f2 (128 -> 64) and the 7x7 / 1x1 / two-channel convolutions are not."""
    from camliflow_amd.csrc import fused
    from camliflow_amd.cores.raft2d import MotionEncoder2D, FlowHead2D, ConvexUpsampler2D
    enc, head, up = MotionEncoder2D(4, 4).cuda(), FlowHead2D(128).cuda(), ConvexUpsampler2D(128).cuda()
    x256, x128 = torch.zeros(1, 256, 8, 8, device='cuda'), torch.zeros(1, 128, 8, 8, device='cuda')
    assert fused.wino_supported(enc.conv_c2, x256) and fused.wino_supported(enc.conv, x256)
    assert fused.wino_supported(head.conv1, x128) and fused.wino_supported(up.mask[0], x128)
    assert not fused.wino_supported(enc.conv_f2, x128) and not fused.wino_supported(enc.conv_c1, torch.zeros(1, 324, 8, 8, device='cuda'))
    assert not fused.wino_supported(head.conv2, x256) and not fused.wino_supported(enc.conv_f1, torch.zeros(1, 2, 8, 8, device='cuda'))
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert not fused.wino_supported(enc.conv_c2, x256)


def _close_through_relus(got, want, tol, what):
    """Gradients that passed ReLUs.  A pre-activation within rounding distance of zero may fall on the other side than the
    reference's did, and each such flip moves the gradients around it by one discrete term (tests/test_upsample_gpu.py has the
    same remark for the folded trunk).  Tile 2 reproduces every decision of the recorded maps; tile 4 (ten times the rounding
    error) flips ONE of the 16,128 activations of conv_c2 here -- the one whose fp64 pre-activation is 3.9e-7
    (tools/wino_flip_probe.py) -- and on this 7 x 12 map that single term is 1 % of the correlation gradient's norm and reaches
    the 3 x 3 neighbourhood of its pixel (9 of 84 pixels, every channel).  So: relative L2 within 2e-2, and all but 15 % of the
    elements within `tol`."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    scale = max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want)
    rel = float(np.sqrt((err ** 2).sum() / (want.astype(np.float64) ** 2).sum()))
    assert got.shape == want.shape and rel <= 2e-2 and float((err > tol * scale).mean()) <= 0.15, (what, rel, float((err > tol * scale).mean()))


@pytest.mark.parametrize('tile', TILES)
def test_update_block_modules_vs_reference_golden(golden, tile, monkeypatch):
    """cores/raft2d.MotionEncoder2D, FlowHead2D and the mask head on the product path (Winograd 3x3 convolutions, fused
    epilogues, cat-free concatenations) against what the REFERENCE's modules recorded with autograd
    (tests/golden/dense_update_block.npz from tests/golden/make_dense_golden.py, models/raft_core.py:142-190; weights
    name-hashed on both sides): outputs, input gradients, fingerprints of every parameter gradient."""
    import zlib
    from modelutils import hashed_fill_
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.blocks import conv_bias_act
    from camliflow_amd.cores.raft2d import ConvexUpsampler2D, FlowHead2D, MotionEncoder2D
    from camliflow_amd.csrc import _lib
    from camliflow_amd.csrc import fused
    runtime.set_backend('hip')
    monkeypatch.setattr(fused, '_WINO_TILE', tile)
    grad_close = (lambda got, want, tol, what: _close(got, want, tol=tol, what=what)) if tile == 2 else _close_through_relus
    g = golden('dense_update_block')
    enc = hashed_fill_(MotionEncoder2D(4, 4)).cuda()
    head = hashed_fill_(FlowHead2D(128, 256)).cuda()
    up = hashed_fill_(ConvexUpsampler2D(128)).cuda()
    flow, corr, hidden = (dev(g[k]).requires_grad_() for k in ('flow', 'corr', 'hidden'))
    runtime.set_census(True)
    runtime.reset_census()
    motion = enc(flow, corr)
    motion.backward(dev(g['gmotion']))
    delta = head(hidden)
    mask = conv_bias_act(up.mask[2], conv_bias_act(up.mask[0], hidden, 'relu'), None)
    (delta * dev(g['gdelta'])).sum().add((mask * dev(g['gmask'])).sum()).backward()
    census = runtime.census()['fused']
    runtime.set_census(False)
    # conv_c2, conv, FlowHead2D.conv1, mask[0]: forward + data gradient (hidden's two are both needed) and weight gradient each
    assert census.get('camli_wino_conv3x3', 0) == 8 and census.get('camli_wino_wrw', 0) == 4, census
    _close(motion, g['motion'], tol=TOL[tile], what='motion features')
    grad_close(flow.grad, g['gflow'], TOL_W[tile], 'gradient of the flow')
    grad_close(corr.grad, g['gcorr'], TOL_W[tile], 'gradient of the correlation window')
    _close(delta, g['delta'], tol=TOL[tile], what='flow update')
    _close(mask, g['mask'], tol=TOL[tile], what='up-sampling mask')
    grad_close(hidden.grad, g['ghidden'], TOL_W[tile], 'gradient of the hidden state')
    for prefix, module in (('enc.', enc), ('head.', head), ('up.', up)):
        for name, p_ in module.named_parameters():
            d = torch.randn(p_.shape, generator=torch.Generator().manual_seed(zlib.crc32(('dir.' + prefix + name).encode())))
            fp = np.array([float(p_.grad.double().norm()), float((p_.grad.double().cpu() * d.double()).sum())])
            want = g['fp_' + prefix + name]
            # (tile 4: the flipped activation's term is in every parameter gradient upstream of it -- 1 % of a bias gradient summed over
            # 84 pixels; tile 2 holds the strict bound on the same code path)
            bound = 1.0 if tile == 2 else 100.0
            assert abs(fp[0] - want[0]) <= bound * 1e-4 * want[0] and abs(fp[1] - want[1]) <= bound * 2e-4 * want[0], (prefix + name, fp, want)


@pytest.mark.parametrize('act', [None, 'relu', 'relu_nan_to_num'])
@pytest.mark.parametrize('deferred', [False, True])
def test_wino_conv_cat_node_vs_oracle(act, deferred, oracle_dense):
    """fused.wino_conv_cat: act(conv3x3(x) + bias) | act(raw + bias2) | tail as one node -- output, every gradient (input,
    weight, both biases, the raw part, the tail) against the numpy oracle composed with the activation's adjoint; with and
    without the per-pass parameter accumulators (runtime.PARAM_GRADS)."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(17)
    b, cin, cout, c2, h, w = 2, 96, 128, 24, 6, 10
    x = rng.standard_normal((b, cin, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * (9 * cin) ** -0.5).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) * 0.3
    raw = rng.standard_normal((b, c2, h, w), dtype=np.float32)
    bias2 = rng.standard_normal(c2).astype(np.float32) * 0.3
    tail = rng.standard_normal((b, 2, h, w), dtype=np.float32)
    gout = rng.standard_normal((b, cout + c2 + 2, h, w), dtype=np.float32)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    with torch.no_grad():
        conv.weight.copy_(dev(wt))
        conv.bias.copy_(dev(bias))
    bias2_p = torch.nn.Parameter(dev(bias2))
    x_d, raw_d, tail_d = dev(x).requires_grad_(), dev(raw).requires_grad_(), dev(tail).requires_grad_()
    runtime.set_deferred_param_grads(deferred)
    try:
        out = fused.wino_conv_cat(x_d, conv, act, others=[(raw_d, bias2_p, 'relu')], tail=tail_d)
        out.backward(dev(gout))
    finally:
        runtime.set_deferred_param_grads(False)
    pre = oracle_dense.conv_taps_fwd(x, wt, (1, 1)) + bias[None, :, None, None]
    want_a = pre if act is None else np.maximum(pre, 0)
    want_b = np.maximum(raw + bias2[None, :, None, None], 0)
    tile = fused._WINO_TILE
    _close(out, np.concatenate([want_a, want_b, tail], axis=1), tol=TOL[tile], what='output')
    # the activation decision of the product: the sign of ITS pre-activation (elements within rounding distance of zero may
    # fall on either side of the oracle's; their gradients are compared through the product's own mask)
    passed = (out[:, :cout] > 0).cpu().numpy() if act is not None else np.ones_like(pre, dtype=bool)
    assert act is None or np.mean(passed != (pre > 0)) < 1e-3
    g_a = (gout[:, :cout] * passed).astype(np.float32)
    want_gx, want_gw = oracle_dense.conv_taps_bwd(g_a, x, wt, (1, 1))
    _close(x_d.grad, want_gx, tol=TOL[tile], what='input gradient')
    _close(conv.weight.grad, want_gw, tol=TOL_W[tile], what='weight gradient')
    _close(conv.bias.grad, g_a.sum((0, 2, 3)), tol=TOL_W[tile], what='bias gradient')
    g_b = gout[:, cout:cout + c2] * (raw + bias2[None, :, None, None] > 0)
    _close(raw_d.grad, g_b.astype(np.float32), what='gradient of the raw part')
    _close(bias2_p.grad, g_b.sum((0, 2, 3)).astype(np.float32), tol=5e-5, what='bias gradient of the raw part')
    _close(tail_d.grad, gout[:, cout + c2:], what='gradient of the tail')

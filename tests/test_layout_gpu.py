"""camli_transpose_planes (channel-first <-> channels-last passes of the update block's convolutions) and the channels-last
convolution node built on it (cores/blocks.py:_CatConvCL) against torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(2, 128, 68 * 120), (3, 64, 36 * 60), (2, 5, 77), (1, 130, 63), (2, 4, 4), (1, 1, 1000)], ids=str)
def test_transpose_planes_is_an_exact_copy(shape):
    """[B,C,P] -> a channel slice of a wider [B,P,Ct] map and back: bit-exact, neighbouring channels untouched (vector and
    scalar paths: multiples of 4 or not)."""
    from camliflow_amd.csrc import fused
    b, c, p = shape
    g = torch.Generator().manual_seed(c + p)
    src = torch.randn(b, c, p, 1, generator=g).cuda()
    for c0, ct in ((0, c), (4, c + 12), (3, c + 5)):
        x_cl = torch.empty_strided((b, ct, p, 1), (p * ct, 1, ct, ct), device='cuda').fill_(float('nan'))
        fused.nchw_into_channels_last(src, x_cl, c0)
        got = x_cl[:, c0:c0 + c]
        assert torch.equal(got, src)
        rest = torch.cat([x_cl[:, :c0], x_cl[:, c0 + c:]], dim=1)
        assert bool(torch.isnan(rest).all())
        back = fused.channels_last_to_nchw(x_cl, c0, c)
        assert back.is_contiguous() and torch.equal(back, src)


def test_transpose_planes_reads_a_channel_slice_in_place():
    """The source may be a channel slice of a contiguous map (a gradient sliced by the adjoint of a cat)."""
    from camliflow_amd.csrc import fused
    g = torch.Generator().manual_seed(1)
    wide = torch.randn(2, 48, 20, 28, generator=g).cuda()
    part = wide[:, 16:48]
    x_cl = torch.empty((2, 32, 20, 28), device='cuda', memory_format=torch.channels_last)
    fused.nchw_into_channels_last(part, x_cl, 0)
    assert torch.equal(x_cl, part)


@pytest.mark.parametrize('ksize,padding', [((1, 5), (0, 2)), ((5, 1), (2, 0)), ((3, 3), (1, 1))], ids=str)
def test_cat_conv_channels_last_matches_the_library_on_nchw(ksize, padding):
    """conv2d(cat([h, m]), w) on channels-last operands: forward and the three gradients vs torch on NCHW (same library
    contraction, other layout: fp32 summation order only)."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.blocks import cat_conv_cl
    runtime.set_backend('hip')
    g = torch.Generator().manual_seed(7)
    h = torch.randn(2, 32, 20, 28, generator=g).cuda().requires_grad_(True)
    m = torch.randn(2, 24, 20, 28, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(40, 56, *ksize, generator=g) * 0.05).cuda().requires_grad_(True)
    gy = torch.randn(2, 40, 20, 28, generator=g).cuda()
    want = torch.nn.functional.conv2d(torch.cat([h, m], 1), w, None, padding=padding)
    want_g = torch.autograd.grad(want, [h, m, w], gy)
    got = cat_conv_cl([h, m], w, padding)
    assert got.is_contiguous()
    got_g = torch.autograd.grad(got, [h, m, w], gy)
    assert (got - want).abs().max() <= 1e-5 * want.abs().max()
    for a, b in zip(got_g, want_g):
        assert a.shape == b.shape and a.is_contiguous()
        assert (a - b).abs().max() <= 1e-5 * b.abs().max()
    # only the weight needs a gradient / only one part does
    gw_only = torch.autograd.grad(cat_conv_cl([h.detach(), m.detach()], w, padding), [w], gy)[0]
    assert (gw_only - want_g[2]).abs().max() <= 1e-5 * want_g[2].abs().max()
    gm_only = torch.autograd.grad(cat_conv_cl([h.detach(), m], w.detach(), padding), [m], gy)[0]
    assert (gm_only - want_g[1]).abs().max() <= 1e-5 * want_g[1].abs().max()

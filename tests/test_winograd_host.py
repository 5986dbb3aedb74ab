"""Host side of the Winograd path (no GPU): the Python wrappers' geometry agrees with the library's own workspace arithmetic,
the activation-bit packing round-trips, strided views are recognised, and ineligible convolutions stay with the library."""
import pytest
import torch


@pytest.mark.parametrize('tile', [2, 4])
@pytest.mark.parametrize('shape', [(8, 256, 192, 68, 120), (1, 256, 126, 47, 156), (2, 101, 99, 3, 5), (4, 128, 256, 1, 9)], ids=str)
def test_workspace_arithmetic_matches_the_library(shape, tile):
    """camli_wino_workspace_bytes (host function of the C-ABI) = P * NT * (Cp + Mp) * 4 with the wrappers' NT: the flop / byte
    figures the bench line declares for these launches are derived from the same geometry."""
    from camliflow_amd.csrc import _lib, fused
    lib = _lib.load()
    b, c, n, h, w = shape
    nt = fused._wino_tiles(b, h, w, tile)
    planes = (tile + 2) ** 2
    cp, mp = (c + 15) // 16 * 16, (n + 3) // 4 * 4
    assert nt % 16 == 0 and nt >= b * -(-h // tile) * -(-w // tile)
    assert lib.camli_wino_workspace_bytes(b, c, n, h, w, tile) == planes * nt * (cp + mp) * 4
    assert lib.camli_wino_weight_floats(c, n, tile) == planes * cp * mp
    assert lib.camli_wino_mask_bytes(b, n, h, w) == b * n * h * ((w + 7) // 8)
    assert lib.camli_wino_wrw_workspace_bytes(b, c, n, h, w, tile) > planes * nt * (c + n) * 4
    assert lib.camli_wino_workspace_bytes(b, c, n, h, w, 3) == 0 and lib.camli_wino_wrw_workspace_bytes(b, c, n, h, w, 8) == 0


@pytest.mark.parametrize('w', [8, 20, 13, 1])
def test_activation_bits_round_trip(w):
    from camliflow_amd.csrc import fused
    g = torch.Generator().manual_seed(w)
    mask = torch.rand(2, 3, 5, w, generator=g) > 0.5
    bits = fused.wino_pack_bits(mask)
    assert bits.dtype == torch.uint8 and bits.shape == (2, 3, 5, (w + 7) // 8)
    assert torch.equal(fused._wino_unpack_bits(bits, w), mask.float())
    if w % 8:      # the bits beyond the row's last pixel stay clear
        assert int(bits[..., -1].max()) < (1 << (w % 8))


def test_image_stride_recognises_channel_slices():
    from camliflow_amd.csrc import fused
    wide = torch.zeros(2, 10, 3, 5)
    assert fused._image_stride(wide) == 150 and fused._image_stride(wide[:, 2:6]) == 150
    assert fused._image_stride(wide[:1, 2:6]) == 60                    # one image: its own extent
    assert fused._image_stride(wide[:, :, :, 1:4]) is None            # rows are not dense
    assert fused._image_stride(wide.permute(0, 2, 3, 1)) is None and fused._image_stride(wide.double()) is None


def test_eligibility():
    """3x3 / stride 1 / padding 1 with at least 96 channels either side, fp32, on the GPU, outside autocast (unless
    CAMLI_AUTOCAST_OWN=1): everything else stays where it was."""
    from camliflow_amd.csrc import fused
    conv = torch.nn.Conv2d(128, 256, 3, padding=1)
    x = torch.zeros(1, 128, 8, 8)
    assert not fused.wino_supported(conv, x)                           # a CPU tensor
    assert fused.wino_shape_supported(128, 256) and fused.wino_shape_supported(629, 128)
    assert not fused.wino_shape_supported(128, 64) and not fused.wino_shape_supported(64, 128)
    assert fused._WINO_TILE in (2, 4)


@pytest.mark.parametrize('case', [(8, 68, 120), (4, 68, 120), (1, 47, 156), (2, 16, 64), (2, 19, 35)], ids=str)
def test_wino1d_workspace_and_kept_transform_arithmetic(case):
    """Host functions of the 1-D Winograd family (no launch): the forward's workspace is the two transform-domain planes
    [8][tiles][Cin + Cout], tiles = 4 pixels along the kernel's axis; the weight gradient may contract a transform the forward
    kept only where a 16-row K split that fills the device divides the unpadded planes (camli_wino1d_wrw_reuse)."""
    from camliflow_amd.csrc import _lib
    lib = _lib.load()
    b, h, w = case
    for axis in (0, 1):
        tiles = b * (h * -(-w // 4) if axis == 0 else w * -(-h // 4))
        for cout in (256, 128):
            assert lib.camli_wino1d_workspace_bytes(b, h, w, 256, cout, axis) == 8 * tiles * (256 + cout) * 4
            need = lib.camli_wino1d_wrw_workspace_bytes(b, h, w, 256, cout, axis)
            assert need >= 8 * tiles * (256 + cout) * 4            # both operands' planes (padded rows) + the K-split partial sums
            reuse = lib.camli_wino1d_wrw_reuse(b, h, w, 256, cout, axis)
            assert reuse in (0, 1) and (reuse == 0 or tiles % 16 == 0)
    assert lib.camli_wino1d_wrw_workspace_bytes(b, h, w, 384, 256, 0) == 0 and lib.camli_wino1d_wrw_reuse(b, h, w, 384, 256, 0) == 0   # Cin % 256
    if case == (8, 68, 120):          # the bench step's shape: 16,320 tiles = 30 splits of 34 K-steps per plane
        assert all(lib.camli_wino1d_wrw_reuse(b, h, w, 256, n, a) == 1 for n in (256, 128) for a in (0, 1))
    if case == (1, 47, 156):          # KITTI: 47 x 39 = 1,833 tiles along x -- no whole 16-row split
        assert lib.camli_wino1d_wrw_reuse(b, h, w, 256, 256, 0) == 0

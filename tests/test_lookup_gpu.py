"""All-pairs lookup kernel (fwd + adjoint) against the oracle and the golden vectors taken from the
reference's Correlation2D; fp32, tolerance stated per assert."""
import numpy as np
import pytest
import torch

from test_oracle_golden import build_levels_numpy

pytestmark = pytest.mark.gpu


def _levels(rng, b, h, w, n_levels=4):
    lv, hl, wl = [], h, w
    for _ in range(n_levels):
        lv.append(rng.standard_normal((b * h * w, hl, wl)).astype(np.float32))
        hl, wl = hl // 2, wl // 2
    return lv


class _Pyr:
    pass


def _run_hip(levels, coords, gout=None):
    from camliflow_amd.csrc import fused
    pyr = fused.AllPairsPyramid()
    pyr.levels = [torch.from_numpy(l).cuda() for l in levels]
    b, _, h, w = coords.shape
    pyr.shape = (b, h, w)
    pyr.token = torch.zeros(1, device='cuda', requires_grad=True)
    out = fused.allpairs_lookup(pyr, torch.from_numpy(coords).cuda(), 4)
    grads = None
    if gout is not None:
        out.backward(torch.from_numpy(gout).cuda())
        grads = [g.cpu().numpy() for g in pyr.grads]
    return out.detach().cpu().numpy(), grads


@pytest.mark.parametrize('shape', [(2, 16, 24), (1, 17, 30), (3, 20, 19), (1, 68, 120)])
def test_lookup_fwd_bwd_vs_oracle(shape, oracle_lib):
    b, h, w = shape
    rng = np.random.default_rng(b * 1000 + h)
    levels = _levels(rng, b, h, w)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    coords = np.stack([xs, ys])[None].repeat(b, 0) + rng.standard_normal((b, 2, h, w)).astype(np.float32) * 4
    coords[:, :, 0, 0] = -9.25
    coords[:, 0, 1, 1] = w + 3.5
    coords[:, :, 2, 2] = 5.0
    coords[:, :, 3, 3] = 1e7          # absurdly far: must give zeros, not crash
    coords = np.ascontiguousarray(coords.astype(np.float32))
    gout = rng.standard_normal((b, 324, h, w)).astype(np.float32)
    out, grads = _run_hip(levels, coords, gout)
    want = oracle_lib.allpairs_lookup_fwd(levels, coords, 4)
    assert np.allclose(out, want, rtol=1e-5, atol=2e-5), np.abs(out - want).max()
    want_g = oracle_lib.allpairs_lookup_bwd([l.shape for l in levels], coords, gout, 4)
    for g, wg in zip(grads, want_g):
        assert np.allclose(g, wg, rtol=1e-5, atol=5e-5), np.abs(g - wg).max()


def test_lookup_backward_accumulates_over_iterations(oracle_lib):
    """two lookups on one pyramid (= two GRU iterations): the gradient pyramid holds the sum"""
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(5)
    b, h, w = 1, 16, 16
    levels = _levels(rng, b, h, w)
    pyr = fused.AllPairsPyramid()
    pyr.levels = [torch.from_numpy(l).cuda() for l in levels]
    pyr.shape = (b, h, w)
    pyr.token = torch.zeros(1, device='cuda', requires_grad=True)
    cs = [rng.random((b, 2, h, w)).astype(np.float32) * 15 for _ in range(2)]
    gs = [rng.standard_normal((b, 324, h, w)).astype(np.float32) for _ in range(2)]
    total = sum((fused.allpairs_lookup(pyr, torch.from_numpy(c).cuda(), 4) * torch.from_numpy(g).cuda()).sum()
                for c, g in zip(cs, gs))
    total.backward()
    for lvl in range(4):
        want = sum(oracle_lib.allpairs_lookup_bwd([l.shape for l in levels], c, g, 4)[lvl] for c, g in zip(cs, gs))
        assert np.allclose(pyr.grads[lvl].cpu().numpy(), want, rtol=1e-5, atol=5e-5)


@pytest.mark.parametrize('name', ['allpairs_even', 'allpairs_odd'])
def test_correlation2d_module_vs_reference_golden(name, golden):
    """Correlation2D under the 'hip' backend (GEMM + pooling + fused lookup + token-routed backward)
    against outputs/gradients recorded from the reference's Correlation2D."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.raft2d import Correlation2D
    g = golden(name)
    corr = Correlation2D(4, 4).cuda()
    with torch.no_grad():
        corr.fnet_aligner.weight.copy_(torch.from_numpy(g['aligner_weight']))
        corr.fnet_aligner.bias.copy_(torch.from_numpy(g['aligner_bias']))
    f1 = torch.from_numpy(g['fmap1']).cuda().requires_grad_(True)
    f2 = torch.from_numpy(g['fmap2']).cuda().requires_grad_(True)
    coords = torch.from_numpy(g['coords']).cuda()
    res = {}
    for backend in ('hip', 'composed'):
        with runtime.use_backend(backend):
            corr.build_cost_volume_pyramid(f1, f2)
            out = corr(coords)
            gf1, gf2 = torch.autograd.grad(out, [f1, f2], torch.from_numpy(g['grad_out']).cuda())
        res[backend] = (out.detach().cpu().numpy(), gf1.cpu().numpy(), gf2.cpu().numpy())
    for backend, (out, gf1, gf2) in res.items():
        assert np.allclose(out, g['out'], rtol=1e-4, atol=3e-4), (backend, np.abs(out - g['out']).max())
        scale = np.abs(g['gfmap1']).max()
        assert np.abs(gf1 - g['gfmap1']).max() < 1e-4 * scale + 1e-4, backend
        assert np.abs(gf2 - g['gfmap2']).max() < 1e-4 * scale + 1e-4, backend


@pytest.mark.parametrize('shape', [(2, 256, 20, 24), (1, 64, 17, 30), (1, 256, 9, 13)], ids=str)
def test_allpairs_build_on_the_matrix_cores_vs_torch(shape):
    """camli_allpairs_build_fwd/bwd (fp32 MFMA, every level straight from the feature maps, pooled target features)
    against the reference composition matmul -> /sqrt(C) -> avg_pool2d chain (raft_core.py:52-68) in torch, fp32:
    levels to 1e-5 relative, feature gradients (random gradient on every level) to 1e-4."""
    import math
    from torch.nn.functional import avg_pool2d
    from camliflow_amd.csrc import fused
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    f1 = torch.randn(b, c, h, w, generator=g).cuda().requires_grad_(True)
    f2 = torch.randn(b, c, h, w, generator=g).cuda().requires_grad_(True)
    pyr = fused.allpairs_pyramid(f1, f2, 4)
    vol = torch.matmul(f1.view(b, c, h * w).transpose(1, 2), f2.view(b, c, h * w)) / math.sqrt(c)
    want = [vol.reshape(b * h * w, 1, h, w)]
    for _ in range(3):
        want.append(avg_pool2d(want[-1], 2, stride=2))
    assert len(pyr.levels) == 4
    grads = []
    for lvl, ref in zip(pyr.levels, want):
        ref = ref[:, 0]
        assert lvl.shape == ref.shape
        assert (lvl - ref.detach()).abs().max() <= 1e-5 * ref.detach().abs().max() + 1e-6
        grads.append(torch.randn(ref.shape, generator=g).cuda())
    want_g1, want_g2 = torch.autograd.grad([r[:, 0] for r in want], [f1, f2], grads)
    pyr.grads = [x.clone() for x in grads]
    g1, g2 = torch.autograd.grad(pyr.token, [f1, f2], torch.zeros(1, device='cuda'))
    for got, ref in ((g1, want_g1), (g2, want_g2)):
        assert (got - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-5


@pytest.mark.parametrize('shape', [(2, 64, 20, 24), (1, 256, 36, 60), (1, 32, 9, 13)], ids=str)
def test_pyramid_adjoint_following_visit_marks_equals_full_scan(shape):
    """camli_allpairs_lookup_bwd_marked + camli_allpairs_build_bwd_marked: the adjoint GEMMs skip every gradient tile no
    lookup ever wrote.  Only exact zeros are skipped, so the feature gradients must equal the unmarked path BIT FOR BIT;
    the marks themselves must cover every non-zero 32x32 block of the gradient pyramid."""
    from camliflow_amd.csrc import fused
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    f1 = torch.randn(b, c, h, w, generator=g).cuda().requires_grad_(True)
    f2 = torch.randn(b, c, h, w, generator=g).cuda().requires_grad_(True)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    base = torch.stack([xs, ys])[None].repeat(b, 1, 1, 1)
    coords = [(base + torch.randn(b, 2, h, w, generator=g) * s).cuda() for s in (0.5, 2.0, 6.0)]
    gouts = [torch.randn(b, 4 * 81, h, w, generator=g).cuda() for _ in coords]

    def run(use_marks, keep=None, splitk=False):
        saved = fused._USE_MARKS, fused._BUILD_SPLITK
        fused._USE_MARKS, fused._BUILD_SPLITK = use_marks, splitk
        try:
            pyr = fused.allpairs_pyramid(f1, f2, 4)
            outs = [fused.allpairs_lookup(pyr, cc, 4) for cc in coords]
            if keep is not None:       # snapshot gradient pyramid + marks right before the build node consumes them
                hook = pyr.token.register_hook(lambda _g: keep.update(grads=[x.clone() for x in pyr.grads],
                                                                       marks=[m.clone() for m in pyr.marks]))
            res = torch.autograd.grad(outs, [f1, f2], gouts)
            if keep is not None:
                hook.remove()
            return res
        finally:
            fused._USE_MARKS, fused._BUILD_SPLITK = saved

    keep = {}
    got = run(True, keep)
    want = run(False)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the split-K form (camli_allpairs_build_bwd_splitk, the default of the product path): g_f1 bit for bit, g_f2 up to the
    # fp32 summation order of the coarse levels' parts; and it is deterministic
    split = run(True, splitk=True)
    assert torch.equal(split[0], want[0])
    assert (split[1] - want[1]).abs().max() <= 1e-5 * want[1].abs().max()
    again = run(True, splitk=True)
    assert torch.equal(again[1], split[1])
    p = h * w
    for grad, mark in zip(keep['grads'], keep['marks']):
        pl = grad.shape[-2] * grad.shape[-1]
        nz = (grad.reshape(b, p, pl) != 0)
        sb, tb = mark.shape[1], mark.shape[2]
        pad = torch.zeros(b, sb * 32, tb * 32, dtype=torch.bool, device=nz.device)
        pad[:, :p, :pl] = nz
        blocks = pad.reshape(b, sb, 32, tb, 32).any(dim=4).any(dim=2)
        assert not (blocks & (mark == 0)).any(), 'a non-zero gradient block is not marked'
        assert (mark != 0).float().mean() < 1.0 or pl <= 128


def test_gradient_pyramid_kept_across_passes_is_clean_and_gives_the_same_gradients():
    """The gradient pyramid survives the backward pass (camli_allpairs_clear_marked zeroes exactly the blocks the lookups
    marked); three passes with different flow fields on the kept buffers == three passes on freshly zero-filled ones, bit
    for bit, and what is kept between passes is all zero, marks included."""
    from camliflow_amd.csrc import fused
    b, c, h, w = 2, 64, 20, 28
    g = torch.Generator().manual_seed(3)
    f1 = torch.randn(b, c, h, w, generator=g).cuda().requires_grad_(True)
    f2 = torch.randn(b, c, h, w, generator=g).cuda().requires_grad_(True)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    base = torch.stack([xs, ys])[None].repeat(b, 1, 1, 1)
    passes = [[(base + torch.randn(b, 2, h, w, generator=g) * s).cuda() for s in scales] for scales in ((0.5, 3.0), (8.0,), (1.0, 1.5, 5.0))]
    gouts = [[torch.randn(b, 4 * 81, h, w, generator=g).cuda() for _ in cs] for cs in passes]

    def run(keep):
        saved = fused._KEEP_GRAD_PYRAMID
        fused._KEEP_GRAD_PYRAMID = keep
        fused._clean_grad_pyramids.clear()
        try:
            results = []
            for cs, gs in zip(passes, gouts):
                pyr = fused.allpairs_pyramid(f1, f2, 4)
                outs = [fused.allpairs_lookup(pyr, cc, 4) for cc in cs]
                results.append(torch.autograd.grad(outs, [f1, f2], gs))
                if keep:
                    assert len(fused._clean_grad_pyramids) == 1
                    grads, marks, _ = next(iter(fused._clean_grad_pyramids.values()))
                    torch.cuda.synchronize()
                    assert all(int(t.count_nonzero()) == 0 for t in grads) and all(int(m.count_nonzero()) == 0 for m in marks)
            return results
        finally:
            fused._KEEP_GRAD_PYRAMID = saved
            fused._clean_grad_pyramids.clear()

    kept, fresh = run(True), run(False)
    for a, c_ in zip(kept, fresh):
        assert torch.equal(a[0], c_[0]) and torch.equal(a[1], c_[1])


@pytest.mark.parametrize('shape', [(1, 128, 40, 64), (2, 40, 33, 50)], ids=str)
def test_build_bwd_splitk_through_the_c_abi_matches_the_unsplit_adjoint(shape):
    """camli_allpairs_build_bwd_splitk with marks == NULL on a DENSE gradient pyramid (every K step live) against
    camli_allpairs_build_bwd: g_f1 and the level-0 g_f2 bit for bit, the split levels up to fp32 summation order; a NULL
    workspace runs the unsplit form (bit for bit everywhere)."""
    import ctypes
    import math
    from camliflow_amd.csrc import _lib
    lib = _lib.load()
    b, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    p = h * w
    sizes = [(h, w)]
    for _ in range(3):
        sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
    f1 = torch.randn(b, c, p, generator=g).cuda()
    f2 = [torch.randn(b, c, a * bb, generator=g).cuda() for a, bb in sizes]
    gv = [torch.randn(b, p, a * bb, generator=g).cuda() for a, bb in sizes]
    p_levels = (ctypes.c_int * 4)(*[a * bb for a, bb in sizes])
    ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])      # noqa: E731
    stream = torch.cuda.current_stream().cuda_stream
    scale = 1.0 / math.sqrt(c)

    def unsplit():
        g1, g2 = torch.empty_like(f1), [torch.empty_like(t) for t in f2]
        rc = lib.camli_allpairs_build_bwd(f1.data_ptr(), ptrs(f2), ptrs(gv), p_levels, 4, g1.data_ptr(), ptrs(g2), b, c, p, scale, stream)
        assert rc == 0
        return g1, g2

    def splitk(with_ws):
        g1, g2 = torch.empty_like(f1), [torch.empty_like(t) for t in f2]
        nbytes = int(lib.camli_allpairs_build_bwd_workspace_bytes(p_levels, 4, b, c, p))
        assert nbytes > 0
        ws = torch.full((nbytes // 4,), float('nan'), device='cuda') if with_ws else None
        rc = lib.camli_allpairs_build_bwd_splitk(f1.data_ptr(), ptrs(f2), ptrs(gv), p_levels, 4, g1.data_ptr(), ptrs(g2), b, c, p, scale,
                                                 None, ws.data_ptr() if with_ws else None, nbytes if with_ws else 0, stream)
        assert rc == 0
        return g1, g2

    want1, want2 = unsplit()
    got1, got2 = splitk(True)
    assert torch.equal(got1, want1) and torch.equal(got2[0], want2[0])
    for a, bb in zip(got2[1:], want2[1:]):
        assert (a - bb).abs().max() <= 1e-5 * bb.abs().max()          # K = h * w terms per sum, two summation orders
    assert not any(torch.equal(a, bb) for a, bb in zip(got2[1:2], want2[1:2])) or p < 1024      # level 1 really was split
    none1, none2 = splitk(False)
    assert torch.equal(none1, want1) and all(torch.equal(a, bb) for a, bb in zip(none2, want2))

"""Full-size checks (BASELINE.json configs[2] / configs[5] shapes) through size-independent
properties, for the sizes at which the scalar CPU oracle would take minutes:

  KNN   neighbour lists sorted by distance, every reported neighbour at least as close as any
        non-neighbour, bit-exact equivariance under a permutation of the queries, self-query hit
  FPS   prefix property (the first m picks of an n-pick run ARE the m-pick run), distinct picks,
        non-increasing coverage radius
  corr2d / all-pairs lookup / set-conv: bilinearity with power-of-two scaling (exact) and the
        adjoint identity <f(x), g> == <x, f^T(g)> in float64 accumulation (fp32 kernels: rel 1e-5)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from camliflow_amd import csrc
    from camliflow_amd.csrc import _lib
    _lib.load()
    return csrc


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


# ------------------------------------------------------------------------------------------ KNN

@pytest.mark.parametrize('case', [(8, 8192, 8192, 16), (2, 16384, 16384, 32), (8, 8192, 8192, 3)],
                         ids=lambda c: 'B%d_M%d_N%d_k%d' % c)
def test_knn_full_size_properties(case, ops):
    b, m, nq, k = case
    g = torch.Generator(device='cpu').manual_seed(sum(case))
    inp = (torch.rand(b, m, 3, generator=g) * 20 - 10).cuda()
    qry = (torch.rand(b, nq, 3, generator=g) * 20 - 10).cuda()
    idx = ops.k_nearest_neighbor(inp, qry, k)
    assert idx.shape == (b, nq, k) and idx.dtype == torch.int64
    assert int(idx.min()) >= 0 and int(idx.max()) < m

    def dist(points, q, index):     # the kernel's own unfused ((dx^2 + dy^2) + dz^2)
        nb = torch.gather(points.unsqueeze(1).expand(-1, q.shape[1], -1, -1), 2,
                          index.unsqueeze(-1).expand(-1, -1, -1, 3))
        d = nb - q.unsqueeze(2)
        return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]

    d = dist(inp, qry, idx)
    assert bool((d[..., 1:] >= d[..., :-1]).all()), 'neighbour lists must be sorted by distance'
    # no duplicates inside a list
    srt = idx.sort(dim=-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all())
    # optimality on a sample of queries: count of inputs strictly closer than the k-th neighbour is < k
    # and every input strictly closer than it is in the list
    sel = torch.randperm(nq, generator=g)[:256].cuda()
    qs = qry[:, sel]                                                    # [B,256,3]
    dx = inp[:, None, :, 0] - qs[:, :, None, 0]
    dy = inp[:, None, :, 1] - qs[:, :, None, 1]
    dz = inp[:, None, :, 2] - qs[:, :, None, 2]
    full = (dx * dx + dy * dy) + dz * dz                                # [B,256,M]
    kth = d[:, sel, -1:]
    closer = full < kth
    assert int(closer.sum(-1).max()) < k
    member = torch.zeros_like(closer)
    member.scatter_(2, idx[:, sel], True)
    assert bool((member | ~closer).all())
    # permutation equivariance over queries: bit-exact
    perm = torch.randperm(nq, generator=g).cuda()
    assert torch.equal(ops.k_nearest_neighbor(inp, qry[:, perm], k), idx[:, perm])
    # a cloud queried with itself finds each (distinct) point first
    own = ops.k_nearest_neighbor(inp, inp, k)
    assert torch.equal(own[..., 0], torch.arange(m, device='cuda').expand(b, -1))


# ------------------------------------------------------------------------------------------ FPS

@pytest.mark.parametrize('case', [(8, 8192, 4096), (2, 16384, 4096), (1, 24576, 2048)],
                         ids=lambda c: 'B%d_N%d_n%d' % c)
def test_fps_full_size_properties(case, ops):
    b, n, ns = case
    g = torch.Generator(device='cpu').manual_seed(sum(case))
    xyz = (torch.rand(b, n, 3, generator=g) * 30).cuda()
    idx = ops.furthest_point_sampling(xyz, ns)
    assert idx.shape == (b, ns) and bool((idx[:, 0] == 0).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    srt = idx.sort(dim=-1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), 'distinct points -> distinct picks'
    # prefix property: bit-exact
    for m in (1, 2, 257, ns // 2):
        assert torch.equal(ops.furthest_point_sampling(xyz, m), idx[:, :m])
    # coverage radius (distance of pick s to picks 0..s-1) never increases, checked on a 512-pick prefix
    p = torch.gather(xyz, 1, idx[:, :512].unsqueeze(-1).expand(-1, -1, 3))
    dmat = torch.cdist(p.double(), p.double())
    mask = torch.tril(torch.ones(512, 512, device='cuda', dtype=torch.bool), diagonal=-1)
    radius = dmat.masked_fill(~mask, float('inf')).min(dim=-1).values[:, 1:]        # pick s vs earlier picks
    assert bool((radius[:, 1:] <= radius[:, :-1] * (1 + 1e-5)).all())
    # and pick s really is the furthest remaining point, for a few s
    for s in (1, 100, 511):
        prev = p[:, :s]
        to_set = torch.cdist(xyz.double(), prev.double()).min(dim=-1).values        # [B,N]
        assert torch.allclose(to_set.max(dim=-1).values, radius[:, s - 1], rtol=1e-5)


# --------------------------------------------------------------------------------------- corr2d

def test_corr2d_full_size_bilinear_and_adjoint(ops):
    """The reference's own self-check shape (correlation_test.cpp:45-60): B=32, C=128, 144x240, md=4."""
    b, c, h, w, md = 32, 128, 144, 240, 4
    g = torch.Generator(device='cuda').manual_seed(3)
    x1 = torch.randn(b, c, h, w, device='cuda', generator=g).requires_grad_(True)
    x2 = torch.randn(b, c, h, w, device='cuda', generator=g).requires_grad_(True)
    go = torch.randn(b, 81, h, w, device='cuda', generator=g)
    out = ops.correlation2d(x1, x2, md)
    out.backward(go)
    # centre tap == channel mean of the product
    centre = (x1.detach() * x2.detach()).mean(1)
    assert (out[:, 40].detach() - centre).abs().max().item() < 2e-5
    # displaced tap against a shifted product (dy=-3, dx=+2 -> channel (1)*9 + 6)
    tap = (x1.detach()[:, :, 3:, :-2] * x2.detach()[:, :, :-3, 2:]).mean(1)
    assert (out[:, 1 * 9 + 6, 3:, :-2].detach() - tap).abs().max().item() < 2e-5
    # bilinear: power-of-two scaling is exact in fp32
    with torch.no_grad():
        assert torch.equal(ops.correlation2d(x1 * 4, x2 * 0.5, md), out * 2)
    # adjoint identity: <corr(x1,x2), go> = <x1, g1> = <x2, g2>
    lhs = _dot(out.detach(), go)
    tol = 1e-5 * abs(lhs) + 1e-8 * out.detach().norm().item() * go.norm().item()
    assert abs(_dot(x1.detach(), x1.grad) - lhs) <= tol
    assert abs(_dot(x2.detach(), x2.grad) - lhs) <= tol


# ---------------------------------------------------------------------------- all-pairs lookup

def test_allpairs_lookup_full_size_linear_and_adjoint():
    """configs[2] working resolution: 68x120 at 1/8 of 540x960 (padded to 544), batch 2, 4 levels."""
    from camliflow_amd.csrc import fused
    b, h, w = 2, 68, 120
    g = torch.Generator(device='cuda').manual_seed(5)
    f1 = torch.randn(b, 128, h, w, device='cuda', generator=g)
    f2 = torch.randn(b, 128, h, w, device='cuda', generator=g)
    with torch.no_grad():
        built = fused.allpairs_pyramid(f1, f2, 4)
    pyr = fused.AllPairsPyramid()           # same levels, leaf token: the accumulated gradient pyramid stays readable
    pyr.levels, pyr.shape = built.levels, built.shape
    pyr.token = torch.zeros(1, device='cuda', requires_grad=True)
    ys, xs = torch.meshgrid(torch.arange(h, device='cuda', dtype=torch.float32),
                            torch.arange(w, device='cuda', dtype=torch.float32), indexing='ij')
    base = torch.stack([xs, ys])[None].expand(b, -1, -1, -1)
    coords = (base + torch.randn(b, 2, h, w, device='cuda', generator=g) * 6).contiguous()
    out = fused.allpairs_lookup(pyr, coords, 4)
    assert out.shape == (b, 4 * 81, h, w)
    # integer coordinates read the volume itself: level 0, centre tap at the identity grid is the diagonal
    ident = fused.allpairs_lookup(pyr, base.contiguous(), 4)
    diag = pyr.levels[0].reshape(b, h * w, h * w).diagonal(dim1=1, dim2=2).reshape(b, h, w)
    assert torch.equal(ident[:, 4 * 9 + 4].detach(), diag)
    # linear in the volume: power-of-two scaling is exact
    scaled = fused.AllPairsPyramid()
    scaled.levels, scaled.shape, scaled.token = [v * 8 for v in pyr.levels], pyr.shape, pyr.token
    assert torch.equal(fused.allpairs_lookup(scaled, coords, 4).detach(), out.detach() * 8)
    # adjoint identity: <lookup(V), go> = sum_l <V_l, gV_l>
    go = torch.randn_like(out)
    out.backward(go)
    lhs = _dot(out.detach(), go)
    rhs = sum(_dot(v, gv) for v, gv in zip(pyr.levels, pyr.grads))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs) + 1e-8 * out.detach().norm().item() * go.norm().item()


# ------------------------------------------------------------------------------------ set-conv

def test_pointconv_dw_full_size_adjoint():
    """GRU3D / motion-encoder shape of configs[2]: batch 8, 2048 points, 384 channels, k = 16 of 32."""
    from camliflow_amd.csrc import fused
    from camliflow_amd.csrc import k_nearest_neighbor
    b, c, n, k = 8, 384, 2048, 16
    g = torch.Generator(device='cuda').manual_seed(7)
    xyz = torch.rand(b, 3, n, device='cuda', generator=g) * 8
    knn = k_nearest_neighbor(xyz, xyz, 32)
    feat = torch.randn(b, c, n, device='cuda', generator=g).requires_grad_(True)
    weight = torch.relu(torch.randn(b, c, n, k, device='cuda', generator=g)).requires_grad_(True)
    shared = fused.SharedSetConvWeights(weight)
    out = fused.pointconv_dw(feat, shared, knn, k)
    ref = (torch.gather(feat.detach().unsqueeze(2).expand(-1, -1, n, -1), 3,
                        knn[:, None, :, :k].expand(-1, c, -1, -1)) * weight.detach()).max(dim=-1).values
    assert torch.equal(out.detach(), ref)
    go = torch.randn_like(out)
    out.backward(go)
    # out is linear in feat for fixed arg-max selection and homogeneous of degree 1 in (feat) and in (weight)
    lhs = _dot(out.detach(), go)
    tol = 1e-5 * abs(lhs) + 1e-8 * out.detach().norm().item() * go.norm().item()
    assert abs(_dot(feat.detach(), feat.grad) - lhs) <= tol
    assert abs(_dot(weight.detach(), weight.grad) - lhs) <= tol

"""HIP kernels (through the C-ABI / the operator boundary) against the CPU oracle and the golden
vectors.  Index ops are compared BIT-EXACT; correlation within the reference's own tolerance
(mean-abs < 1e-6, correlation_test.cpp:82-89)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from camliflow_amd import csrc
    from camliflow_amd.csrc import _lib
    _lib.load()  # must be the HIP library; no fallback exists
    return csrc


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cloud(rng, b, n, d, kind):
    if kind == 'uniform':
        return (rng.random((b, n, d), dtype=np.float32) * 10).astype(np.float32)
    if kind == 'lattice':  # many exactly equal distances
        return rng.integers(0, 6, size=(b, n, d)).astype(np.float32)
    if kind == 'dup':      # 25 % exact duplicates (datasets sample with replacement, flyingthings3d.py:75-76)
        x = (rng.random((b, n, d), dtype=np.float32) * 10).astype(np.float32)
        q = n // 4
        x[:, n - q:] = x[:, :q]
        return x
    raise ValueError(kind)


KNN_CASES = [
    # (B, M, Nq, D, k)
    (2, 2048, 2048, 3, 16), (2, 2048, 2048, 3, 32), (1, 8192, 4096, 3, 16), (2, 2048, 1024, 3, 3),
    (2, 512, 256, 3, 3), (1, 2048, 8192, 3, 3), (2, 2048, 8160, 2, 1), (1, 4096, 34560, 2, 1),
    (1, 256, 2048, 3, 16), (8, 2048, 2048, 3, 16), (3, 1000, 777, 3, 16), (2, 77, 130, 2, 3),
    (1, 5, 9, 3, 16), (1, 300, 300, 3, 8), (1, 300, 300, 3, 4), (1, 300, 64, 3, 5), (1, 100, 70, 2, 64),
    (1, 16384, 4096, 3, 16),
    # round 4: the slot-count / team boundaries of the cross-lane kernels (M = 64 J, 2048 per team wave), k = 1 in 3-D,
    # wide k in 2-D, fewer candidates than lanes
    (2, 257, 100, 3, 16), (1, 2049, 300, 3, 16), (2, 3000, 500, 3, 32), (1, 5000, 333, 3, 3), (1, 12000, 200, 3, 8),
    (2, 640, 640, 2, 16), (1, 2048, 2048, 3, 1), (1, 40, 64, 3, 32), (1, 4096, 4096, 3, 4),
]


@pytest.mark.parametrize('case', KNN_CASES, ids=lambda c: 'B%d_M%d_N%d_D%d_k%d' % c)
@pytest.mark.parametrize('kind', ['uniform', 'dup', 'lattice'])
def test_knn_bit_exact(case, kind, ops, oracle_lib):
    b, m, nq, d, k = case
    if kind == 'lattice' and m * nq * b > 2048 * 2048 * 2:
        pytest.skip('lattice case kept small (every query takes the in-order recompute path)')
    rng = np.random.default_rng(hash((case, kind)) % (2 ** 32))
    inp = _cloud(rng, b, m, d, kind)
    qry = _cloud(rng, b, nq, d, kind)
    if kind == 'dup':
        qry[:, :min(nq, m) // 2] = inp[:, :min(nq, m) // 2]
    got = ops.k_nearest_neighbor(dev(inp), dev(qry), k).cpu().numpy()
    want = oracle_lib.knn(inp, qry, k)
    assert got.dtype == np.int64 and got.shape == (b, nq, k)
    assert np.array_equal(got, want), 'mismatching entries: %d of %d' % ((got != want).sum(), got.size)


@pytest.mark.parametrize('case', [(2, 2048, 1024, 3, 16), (1, 8192, 512, 3, 16), (2, 2048, 700, 3, 3), (2, 1000, 300, 2, 1),
                                  (1, 4096, 300, 3, 32)], ids=lambda c: 'B%d_M%d_N%d_D%d_k%d' % c)
@pytest.mark.parametrize('mode', ['lane', 'xlane'])
def test_knn_kernel_families_agree_with_the_oracle(case, mode, ops, oracle_lib, monkeypatch):
    """The kernel families behind k_nearest_neighbor (lane-per-query, candidates-across-lanes) on a cloud
    with 25 % duplicates: bit-exact vs the oracle whichever one the dispatcher is told to take."""
    b, m, nq, d, k = case
    monkeypatch.setenv('CAMLI_KNN', mode)
    rng = np.random.default_rng(hash((case, 'fam')) % (2 ** 32))
    inp, qry = _cloud(rng, b, m, d, 'dup'), _cloud(rng, b, nq, d, 'dup')
    qry[:, :min(nq, m) // 2] = inp[:, :min(nq, m) // 2]
    got = ops.k_nearest_neighbor(dev(inp), dev(qry), k).cpu().numpy()
    assert np.array_equal(got, oracle_lib.knn(inp, qry, k))


def _shaped_cloud(rng, kind, b, m, d):
    if kind == 'uniform':
        return (rng.random((b, m, d), dtype=np.float32) * 10).astype(np.float32)
    if kind == 'clustered':       # twenty tight blobs: dense next to empty
        centres = rng.random((b, 20, d)) * 10
        pick = rng.integers(0, 20, size=(b, m))
        return (np.take_along_axis(centres, pick[..., None].repeat(d, axis=2), axis=1) + rng.normal(0, 0.3, (b, m, d))).astype(np.float32)
    if kind == 'lattice':         # integer grid: exact distance ties everywhere, many of them at the k-th distance
        return rng.integers(0, 12, size=(b, m, d)).astype(np.float32)
    ang, r = rng.random((b, m)) * 2 * np.pi, rng.exponential(8.0, (b, m)) + 2      # a disk with a dense centre, like a LiDAR sweep
    cols = [r * np.cos(ang), r * np.sin(ang), rng.normal(0, 0.3, (b, m)) + 0.02 * r]
    return np.stack(cols[:d], axis=2).astype(np.float32)


@pytest.mark.parametrize('k', [1, 2, 5, 16, 20, 32])
def test_knn_initial_distance_and_far_candidates(k, ops, oracle_lib):
    """The reference's list starts as k entries (1e9, index 0): candidates farther than 1e9 never enter, one at EXACTLY
    1e9 ties the initial entries (start-slot rule of k_nearest_neighbor_kernel.cu:80), fewer than k candidates inside
    leave index 0 in the tail.  sqrt(1e9) is not representable, 31622.7765^2 lands on both sides of 1e9 in fp32."""
    rng = np.random.default_rng(k)
    m, nq = 300, 130
    inp = (rng.random((2, m, 3), dtype=np.float32) * 10).astype(np.float32)
    qry = (rng.random((2, nq, 3), dtype=np.float32) * 10).astype(np.float32)
    inp[0, ::3] += 1e6                       # farther than 1e9 (squared) from every query
    inp[1, 5:] += 1e6                        # only five candidates inside: fewer than k for most k
    qry[0, :16] = 0.0
    inp[0, 1] = (31622.0, 6.0, 0.0)          # squared distance from the origin = 999950884 + 36 ... near 1e9
    inp[0, 2] = (np.float32(np.sqrt(np.float32(1e9))), 0.0, 0.0)
    inp[0, 4] = (0.0, np.float32(np.sqrt(np.float32(1e9))), 0.0)
    got = ops.k_nearest_neighbor(dev(inp), dev(qry), k).cpu().numpy()
    assert np.array_equal(got, oracle_lib.knn(inp, qry, k))


def test_knn_channel_first_layout_is_sniffed_like_the_reference(ops, oracle_lib):
    rng = np.random.default_rng(1)
    inp, qry = _cloud(rng, 2, 500, 3, 'uniform'), _cloud(rng, 2, 200, 3, 'uniform')
    a = ops.k_nearest_neighbor(dev(inp), dev(qry), 16)
    b = ops.k_nearest_neighbor(dev(inp).transpose(1, 2), dev(qry).transpose(1, 2), 16)
    assert torch.equal(a, b)
    assert np.array_equal(a.cpu().numpy(), oracle_lib.knn(inp, qry, 16))


@pytest.mark.parametrize('name', sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, 'knn_d*.npz'))))
def test_knn_golden(name, ops, golden):
    g = golden(name)
    got = ops.k_nearest_neighbor(dev(g['input']), dev(g['query']), int(g['k'])).cpu().numpy()
    safe = g['safe']
    assert np.array_equal(got[safe], g['indices'][safe])


FPS_CASES = [(2, 8192, 4096), (16, 8192, 4096), (1, 16384, 4096), (3, 4100, 4096), (2, 1024, 256), (1, 1500, 700),
             (1, 65, 64), (2, 20000, 512)]


@pytest.mark.parametrize('case', FPS_CASES, ids=lambda c: 'B%d_N%d_n%d' % c)
@pytest.mark.parametrize('kind', ['uniform', 'dup'])
def test_fps_bit_exact(case, kind, ops, oracle_lib):
    b, n, ns = case
    rng = np.random.default_rng(hash((case, kind)) % (2 ** 32))
    xyz = _cloud(rng, b, n, 3, kind)
    got = ops.furthest_point_sampling(dev(xyz), ns).cpu().numpy()
    want = oracle_lib.fps(xyz, ns)
    assert got.dtype == np.int64
    assert np.array_equal(got, want), 'first mismatch at %s' % (np.argwhere(got != want)[:1],)


@pytest.mark.parametrize('name', ['fps_a', 'fps_b', 'fps_dup'])
def test_fps_golden(name, ops, golden):
    g = golden(name)
    got = ops.furthest_point_sampling(dev(g['xyz']), int(g['n_samples'])).cpu().numpy()
    assert np.array_equal(got, g['indices'])


def test_fps_all_points_identical(ops, oracle_lib):
    xyz = np.ones((1, 300, 3), dtype=np.float32)
    got = ops.furthest_point_sampling(dev(xyz), 10).cpu().numpy()
    assert np.array_equal(got, oracle_lib.fps(xyz, 10))


PRUNED_CASES = [(2, 8192, 8191), (1, 16384, 2048), (3, 4100, 4096), (2, 1024, 1023), (1, 65, 64), (1, 64, 63), (1, 7, 6), (1, 2, 1),
                (4, 777, 300)]


@pytest.mark.parametrize('variant', ['p16', 'p32', 'p8t1024', 'p16t1024'])
@pytest.mark.parametrize('kind', ['uniform', 'dup', 'lattice', 'plane'])
def test_fps_bucket_pruned_variants_bit_exact(variant, kind, ops, oracle_lib, monkeypatch):
    """The bucket-pruned kernel (round 3: Morton-sorted buckets, exact box test) in every register / thread shape, forced
    through CAMLI_FPS on clouds its default dispatch would not see: tiny, ragged, all but one point picked (the wrapper requires n_samples < N like wrapper.py:98), many exact
    ties (lattice: lowest ORIGINAL index must win although the points are permuted) and a degenerate planar cloud."""
    monkeypatch.setenv('CAMLI_FPS', variant)
    cap = {'p16': 16 * 512, 'p32': 32 * 512, 'p8t1024': 8 * 1024, 'p16t1024': 16 * 1024}[variant]
    for case in PRUNED_CASES:
        b, n, ns = case
        if n > cap or (kind == 'lattice' and n > 4100):
            continue
        rng = np.random.default_rng(hash((case, kind)) % (2 ** 32))
        xyz = _cloud(rng, b, n, 3, 'uniform' if kind == 'plane' else kind)
        if kind == 'plane':
            xyz[:, :, 2] = 3.0
        got = ops.furthest_point_sampling(dev(xyz), ns).cpu().numpy()
        want = oracle_lib.fps(xyz, ns)
        assert np.array_equal(got, want), (case, 'first mismatch at %s' % (np.argwhere(got != want)[:1],))


CORR_CASES = [(2, 32, 24, 40, 4), (1, 64, 18, 30, 4), (1, 96, 36, 60, 4), (1, 128, 18, 30, 4), (1, 192, 9, 15, 4),
              (2, 20, 7, 70, 3), (1, 300, 5, 9, 2), (1, 8, 5, 6, 1), (1, 16, 6, 6, 5)]


@pytest.mark.parametrize('case', CORR_CASES, ids=lambda c: 'B%d_C%d_H%d_W%d_md%d' % c)
def test_correlation_fwd_bwd_vs_oracle(case, ops, oracle_lib):
    b, c, h, w, md = case
    rng = np.random.default_rng(sum(case))
    x1 = rng.standard_normal((b, c, h, w)).astype(np.float32)
    x2 = rng.standard_normal((b, c, h, w)).astype(np.float32)
    go = rng.standard_normal((b, (2 * md + 1) ** 2, h, w)).astype(np.float32)
    t1, t2 = dev(x1).requires_grad_(True), dev(x2).requires_grad_(True)
    out = ops.correlation2d(t1, t2, md)
    out.backward(dev(go))
    in1 = np.ascontiguousarray(x1.transpose(0, 2, 3, 1))
    in2 = np.ascontiguousarray(x2.transpose(0, 2, 3, 1))
    want = oracle_lib.corr2d_fwd(in1, in2, md)
    g1, g2 = oracle_lib.corr2d_bwd(go, in1, in2, md)
    assert np.abs(out.detach().cpu().numpy() - want).mean() < 1e-6
    assert np.abs(out.detach().cpu().numpy() - want).max() < 2e-5
    assert np.abs(t1.grad.cpu().numpy() - g1.transpose(0, 3, 1, 2)).mean() < 1e-6
    assert np.abs(t2.grad.cpu().numpy() - g2.transpose(0, 3, 1, 2)).mean() < 1e-6
    assert np.abs(t1.grad.cpu().numpy() - g1.transpose(0, 3, 1, 2)).max() < 5e-5


@pytest.mark.parametrize('name', ['corr2d_a', 'corr2d_b', 'corr2d_c'])
def test_correlation_golden(name, ops, golden):
    g = golden(name)
    t1, t2 = dev(g['input1']).requires_grad_(True), dev(g['input2']).requires_grad_(True)
    out = ops.correlation2d(t1, t2, int(g['md']))
    out.backward(dev(g['grad_output']))
    assert np.abs(out.detach().cpu().numpy() - g['output']).mean() < 1e-6
    assert np.abs(t1.grad.cpu().numpy() - g['grad1']).mean() < 1e-6
    assert np.abs(t2.grad.cpu().numpy() - g['grad2']).mean() < 1e-6


def test_correlation_recipe_of_reference_self_check(ops):
    """correlation_test.cpp:45-60 recipe (rand, C=128, 144x240, md=4) at B=2: native vs the composed
    formulation (the reference's 'naive' arm), mean-abs < 1e-6 for out and both grads."""
    g = torch.Generator(device='cpu').manual_seed(0)
    x1 = torch.rand(2, 128, 144, 240, generator=g).cuda().requires_grad_(True)
    x2 = torch.rand(2, 128, 144, 240, generator=g).cuda().requires_grad_(True)
    go = torch.rand(2, 81, 144, 240, generator=g).cuda()
    out = ops.correlation2d(x1, x2, 4)
    out.backward(go)
    g1, g2 = x1.grad.clone(), x2.grad.clone()
    x1.grad = x2.grad = None
    ref = ops.correlation2d(x1, x2, 4, cpp_impl=False)
    ref.backward(go)
    assert (out - ref).abs().mean().item() < 1e-6
    assert (g1 - x1.grad).abs().mean().item() < 1e-6
    assert (g2 - x2.grad).abs().mean().item() < 1e-6


def test_bad_arguments_raise(ops):
    from camliflow_amd.csrc._lib import CamliHipError
    x = torch.rand(1, 100, 3).cuda()
    with pytest.raises(CamliHipError):
        ops.k_nearest_neighbor(x, x, 65)
    with pytest.raises(CamliHipError):
        ops.k_nearest_neighbor(torch.rand(1, 100, 4).cuda(), torch.rand(1, 10, 4).cuda(), 3)

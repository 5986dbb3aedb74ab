"""gather / knn-interpolation / point-cost-volume gather kernels against the oracle (fp32).
Forward results are pure copies or short fixed-order sums -> compared exactly or to 1e-6;
backward uses float atomics -> 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# the adjoint stages 8 / 4 / 2 / 1 rows of gout in LDS depending on the row length (I*4 bytes), falls back to direct
# gathers above 128 KiB rows, and takes the scalar staging path when I % 4 != 0: one case per branch
@pytest.mark.parametrize('case', [(2, 5, 40, (17, 3)), (8, 130, 2048, (8160, 1)), (2, 2048, 2048, (1024, 3)), (1, 3, 8192, (4096,)),
                                  (2, 19, 300, (1500,)), (1, 4, 50, (33000,))],
                         ids=str)
def test_gather_points(case, oracle_lib):
    from camliflow_amd.csrc import fused
    b, c, m, ishape = case
    rng = np.random.default_rng(c)
    data = rng.standard_normal((b, c, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(b,) + ishape).astype(np.int64)
    t = dev(data).requires_grad_(True)
    out = fused.gather_points(t, dev(idx))
    assert out.shape == (b, c) + ishape
    flat = idx.reshape(b, -1)
    want = oracle_lib.gather_cf(data, flat).reshape(out.shape)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(g))
    assert np.allclose(t.grad.cpu().numpy(), oracle_lib.scatter_add_cf(g.reshape(b, c, -1), flat, m), rtol=1e-5, atol=1e-5)


def test_gather_golden(golden):
    from camliflow_amd.csrc import fused
    g = golden('batch_indexing')
    out = fused.gather_points(dev(g['data']), dev(g['indices']))
    assert np.array_equal(out.cpu().numpy(), g['out_cf'])


@pytest.mark.parametrize('invariant', [False, True], ids=['atomic_adjoint', 'sorted_adjoint'])
@pytest.mark.parametrize('case', [(2, 3, 2048, 8192, 3), (8, 3, 2048, 256, 3), (1, 67, 512, 1024, 3), (2, 7, 200, 90, 5)], ids=str)
def test_knn_interpolate(case, invariant, oracle_lib):
    """forward and feature adjoint vs the oracle; ``invariant`` (the GRU loops' declaration: same clouds every call) selects the
    atomic-free adjoint on per-pass geometry (camli_knn_interp_weights + camli_knn_interp_bwd_sorted), which must also give the
    same bits on a second backward pass through the cached geometry"""
    from camliflow_amd.csrc import fused, k_nearest_neighbor
    b, c, m, nq, k = case
    rng = np.random.default_rng(nq)
    in_xyz = rng.standard_normal((b, 3, m)).astype(np.float32)
    q_xyz = rng.standard_normal((b, 3, nq)).astype(np.float32)
    q_xyz[:, :, :4] = in_xyz[:, :, :4]       # coincident points -> clamp(1e-8)
    feat = rng.standard_normal((b, c, m)).astype(np.float32)
    knn = k_nearest_neighbor(dev(in_xyz), dev(q_xyz), k)
    tf = dev(feat).requires_grad_(True)
    d_in, d_q = dev(in_xyz), dev(q_xyz)
    out = fused.knn_interpolate(d_in, tf, d_q, knn, k, invariant=invariant)
    want = oracle_lib.knn_interp_fwd(in_xyz, feat, q_xyz, knn.cpu().numpy())
    assert np.allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(g))
    assert np.allclose(tf.grad.cpu().numpy(), oracle_lib.knn_interp_bwd(in_xyz, g, q_xyz, knn.cpu().numpy(), m),
                       rtol=1e-4, atol=1e-5)
    if invariant:       # a second call on the same clouds (cached geometry): bit-identical gradient, fixed summation order
        first = tf.grad.clone()
        tf.grad = None
        fused.knn_interpolate(d_in, tf, d_q, knn, k, invariant=True).backward(dev(g))
        assert torch.equal(tf.grad, first)


@pytest.mark.parametrize('case', [(2, 3, 2048, 1024, 3), (1, 67, 512, 1024, 3), (2, 7, 200, 90, 5)], ids=str)
def test_knn_interpolate_coordinate_adjoint(case, oracle_lib):
    """CamLiPWC back-warps with a live flow (camlipwc_core.py:172-179): gradients reach both coordinate sets.
    Checked against the oracle's adjoint (itself pinned on the reference's autograd, tests/test_oracle_golden.py);
    coincident points exercise the clamp / norm-at-zero subgradients."""
    from camliflow_amd.csrc import fused, k_nearest_neighbor
    b, c, m, nq, k = case
    rng = np.random.default_rng(nq + 1)
    in_xyz = rng.standard_normal((b, 3, m)).astype(np.float32)
    q_xyz = rng.standard_normal((b, 3, nq)).astype(np.float32)
    q_xyz[:, :, :4] = in_xyz[:, :, :4]
    feat = rng.standard_normal((b, c, m)).astype(np.float32)
    knn = k_nearest_neighbor(dev(in_xyz), dev(q_xyz), k)
    ti, tf, tq = dev(in_xyz).requires_grad_(True), dev(feat).requires_grad_(True), dev(q_xyz).requires_grad_(True)
    out = fused.knn_interpolate(ti, tf, tq, knn, k)
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(g))
    want_in, want_q = oracle_lib.knn_interp_bwd_xyz(in_xyz, feat, g, q_xyz, knn.cpu().numpy())
    scale = max(np.abs(want_in).max(), np.abs(want_q).max())
    assert np.abs(ti.grad.cpu().numpy() - want_in).max() <= 2e-4 * scale
    assert np.abs(tq.grad.cpu().numpy() - want_q).max() <= 2e-4 * scale
    assert np.allclose(tf.grad.cpu().numpy(), oracle_lib.knn_interp_bwd(in_xyz, g, q_xyz, knn.cpu().numpy(), m),
                       rtol=1e-4, atol=1e-5)


def test_backwarp_3d_with_live_flow_stays_on_hip():
    """the CamLiPWC call pattern: backwarp_3d(xyz1, xyz2, flow) with flow requiring grad -- under CAMLI strict mode
    nothing may drop to the composed formulation, and the flow gradient must match the composed one"""
    from camliflow_amd.cores import geometry, runtime
    torch.manual_seed(0)
    xyz1, xyz2 = torch.rand(2, 3, 512, device='cuda') * 4, torch.rand(2, 3, 512, device='cuda') * 4
    flow = (torch.randn(2, 3, 512, device='cuda') * 0.1).requires_grad_(True)
    g = torch.randn(2, 3, 512, device='cuda')
    res = {}
    for backend in ('hip', 'composed'):
        with runtime.use_backend(backend):
            runtime.set_strict(backend == 'hip')
            try:
                out = geometry.backwarp_3d(xyz1, xyz2, flow)
                res[backend] = (out.detach(), torch.autograd.grad(out, flow, g)[0])
            finally:
                runtime.set_strict(False)
    assert torch.allclose(res['hip'][0], res['composed'][0], rtol=1e-5, atol=1e-5)
    err = (res['hip'][1] - res['composed'][1]).abs().max() / res['composed'][1].abs().max()
    assert err < 1e-3, err


def test_knn_interpolation_golden_through_the_core_function(golden):
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.geometry import knn_interpolation
    g = golden('knn_interpolation')
    with runtime.use_backend('hip'):
        out = knn_interpolation(dev(g['in_xyz']), dev(g['feat']), dev(g['q_xyz']), k=3)
    assert np.allclose(out.cpu().numpy(), g['out'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', [(2, 2048, 2048, 16), (8, 2048, 256, 16), (1, 300, 77, 5)], ids=str)
def test_corr3d_lookup_input(case, oracle_lib):
    from camliflow_amd.csrc import fused
    b, n, m, k = case
    rng = np.random.default_rng(m)
    xyz1 = rng.standard_normal((b, 3, n)).astype(np.float32)
    xyz2 = rng.standard_normal((b, 3, m)).astype(np.float32)
    cost = rng.standard_normal((b, n, m)).astype(np.float32)
    knn = np.stack([np.stack([rng.permutation(m)[:k] for _ in range(n)]) for _ in range(b)]).astype(np.int64)
    tc = dev(cost).requires_grad_(True)
    out = fused.corr3d_lookup_input(tc, dev(xyz1), dev(xyz2), dev(knn))
    assert np.array_equal(out.detach().cpu().numpy(), oracle_lib.corr3d_gather_fwd(xyz1, xyz2, cost, knn))
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(g))
    want = np.zeros_like(cost)
    np.add.at(want, (np.arange(b)[:, None, None], np.arange(n)[None, :, None], knn), g[:, 3])
    assert np.allclose(tc.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_correlation3d_module_batched_levels_vs_composed():
    """raft3d.Correlation3D under the 'hip' backend (four levels through cost_mlp in one call) vs the
    per-level composed formulation: output and parameter / feature gradients, fp32."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.geometry import build_pc_pyramid
    from camliflow_amd.cores.raft3d import Correlation3D
    from modelutils import hashed_fill_
    torch.manual_seed(3)
    corr = hashed_fill_(Correlation3D(out_channels=128, k=16)).cuda()
    pc1 = torch.rand(2, 3, 4200, device='cuda') * 6
    pc2 = pc1 + 0.05 * torch.randn_like(pc1)
    xyzs1, xyzs2, _, _ = build_pc_pyramid(pc1, pc2, [4096, 2048, 1024, 512, 256])
    xyz1, targets = xyzs1[2], xyzs2[2:]
    gout = torch.randn(2, 128, 2048, device='cuda')
    res = {}
    for backend in ('hip', 'composed'):
        f1 = torch.randn(2, 128, 2048, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)).requires_grad_(True)
        f2 = torch.randn(2, 128, 2048, device='cuda', generator=torch.Generator(device='cuda').manual_seed(6)).requires_grad_(True)
        corr.zero_grad()
        with runtime.use_backend(backend):
            corr.build_cost_volume_pyramid(f1, f2, targets)
            out = corr(xyz1, targets)
        out.backward(gout)
        res[backend] = (out.detach(), f1.grad, f2.grad, [p.grad.clone() for p in corr.parameters()])
    a, b = res['hip'], res['composed']
    assert torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5), (a[0] - b[0]).abs().max()
    # gradients: a handful of the 4M ReLU decisions inside cost_mlp flip with the GEMM's summation order
    # (one batched call vs four), each flip moves a gradient by one term -> norm-relative 5e-3
    errs = [((x - y).norm() / y.norm()).item() for x, y in zip([a[1], a[2]] + a[3], [b[1], b[2]] + b[3])]
    assert max(errs) <= 5e-3, errs


@pytest.mark.parametrize('case', [(8, 128, 68, 120, 2048), (2, 7, 9, 13, 100), (1, 64, 47, 156, 2048), (3, 1, 2, 2, 65)],
                         ids=lambda c: 'B%d_C%d_%dx%d_N%d' % c)
def test_bilinear_sample_vs_oracle(case, oracle_lib):
    from camliflow_amd.csrc import fused
    b, c, h, w, n = case
    rng = np.random.default_rng(sum(case))
    feat = rng.standard_normal((b, c, h, w)).astype(np.float32)
    uv = (rng.random((b, 2, n), dtype=np.float32) * np.array([w + 6, h + 6], dtype=np.float32)[None, :, None] - 3).astype(np.float32)
    uv[:, :, :5] = np.floor(uv[:, :, :5])                                        # exact pixel centres
    got = fused.bilinear_sample(torch.from_numpy(feat).cuda(), torch.from_numpy(uv).cuda()).cpu().numpy()
    want = oracle_lib.bilinear_sample_fwd(feat, uv)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)


def test_bilinear_sample_golden_and_wrapper_dispatch(golden):
    """The reference's grid_sample_wrapper output, through geometry.grid_sample_wrapper under both backends."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.geometry import grid_sample_wrapper
    g = golden('grid_sample')
    feat, uv = torch.from_numpy(g['feat']).cuda(), torch.from_numpy(g['uv']).cuda()
    for backend in ('hip', 'composed'):
        with runtime.use_backend(backend):
            out = grid_sample_wrapper(feat, uv)
        assert np.allclose(out.cpu().numpy(), g['out'], rtol=1e-5, atol=5e-6), backend   # the library's GPU kernel itself is 7e-7 off its CPU result
    # a differentiable input keeps the differentiable (library) formulation
    with runtime.use_backend('hip'):
        out = grid_sample_wrapper(feat.clone().requires_grad_(True), uv)
    assert out.requires_grad


def test_ids_flow_vs_oracle_golden_and_autograd(golden, oracle_lib):
    """camli_ids_flow_fwd/bwd: the reference's paral2persp read-out (golden), the oracle at a larger size, and
    the adjoint against autograd through geometry.paral2persp."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.geometry import flows_paral2persp, paral2persp
    from test_oracle_golden import _ids_consts
    g = golden('ids_flow')
    intr = torch.from_numpy(g['intrinsics']).cuda()
    (ph, pw), (qh, qw) = g['persp_hw'], g['paral_hw']
    persp = {'sensor_h': int(ph), 'sensor_w': int(pw), 'f': intr[:, 0], 'cx': intr[:, 1], 'cy': intr[:, 2]}
    paral = {'sensor_h': int(qh), 'sensor_w': int(qw)}
    pc1 = torch.from_numpy(g['pc1']).cuda()
    with runtime.use_backend('hip'):
        got = flows_paral2persp(pc1, [torch.from_numpy(g['flow']).cuda()], persp, paral)[0]
    # z = exp(.) is 5..35 and the flow is the DIFFERENCE of two such values: one ulp of exp is ~4e-6 here
    assert np.allclose(got.cpu().numpy(), g['out'], rtol=1e-5, atol=5e-5)
    # larger, with gradients
    rng = np.random.default_rng(9)
    b, n = 8, 8192
    big_pc1 = np.concatenate([rng.uniform(-14, 14, (b, 1, n)), rng.uniform(-8, 8, (b, 1, n)), rng.uniform(30, 120, (b, 1, n))], 1).astype(np.float32)
    flow0 = (rng.standard_normal((b, 3, n)) * 0.2).astype(np.float32)
    intr8 = intr[:1].expand(b, -1).contiguous()
    persp8 = dict(persp, f=intr8[:, 0], cx=intr8[:, 1], cy=intr8[:, 2])
    t_pc1 = torch.from_numpy(big_pc1).cuda()
    gout = torch.randn(b, 3, n, device='cuda')
    res = []
    for backend in ('hip', 'composed'):
        flow = torch.from_numpy(flow0).cuda().requires_grad_(True)
        with runtime.use_backend(backend):
            out = flows_paral2persp(t_pc1, [flow], persp8, paral)[0]
        out.backward(gout)
        res.append((out.detach(), flow.grad))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=5e-5)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-6)
    origin = paral2persp(t_pc1, persp8, paral).cpu().numpy()
    f = intr8.cpu().numpy()
    want = oracle_lib.ids_flow_fwd(big_pc1, flow0, origin, f[:, 0], f[:, 1], f[:, 2], *_ids_consts(g))
    assert np.allclose(res[0][0].cpu().numpy(), want, rtol=1e-6, atol=5e-5)


@pytest.mark.parametrize('case', [(2, 2, (64, 81), True), (8, 2, (540, 960), True), (3, 3, (4097,), False), (2, 3, (2048,), True)],
                         ids=lambda c: 'B%d_C%d_%s_%s' % (c[0], c[1], 'x'.join(map(str, c[2])), 'mask' if c[3] else 'nomask'))
def test_sequence_loss_l2_hip_vs_composed(case):
    """objectives._sequence_loss (order l2-norm) through camli_masked_l2_fwd/bwd vs the torch formulation of
    models/losses.py:64-119: value and the gradient of every iterate."""
    from types import SimpleNamespace
    from camliflow_amd.cores import objectives, runtime
    b, c, sp, masked = case
    torch.manual_seed(sum(sp) + b)
    target = torch.randn(b, c + (1 if masked else 0), *sp, device='cuda')
    if masked:
        target[:, c] = (torch.rand(b, *sp, device='cuda') > 0.3).float()
    preds0 = [torch.randn(b, c, *sp, device='cuda') for _ in range(4)]
    preds0[1][0, :, ..., :3] = target[0, :c, ..., :3]          # exact hits: zero error, zero gradient
    cfgs = SimpleNamespace(gamma=0.8, order='l2-norm')
    res = []
    for backend in ('hip', 'composed'):
        preds = [q.clone().requires_grad_(True) for q in preds0]
        with runtime.use_backend(backend):
            loss = objectives._sequence_loss(preds, target, cfgs, c)
        loss.backward()
        res.append((loss.item(), [q.grad for q in preds]))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    for g1, g2 in zip(res[0][1], res[1][1]):
        assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-9), (g1 - g2).abs().max()


@pytest.mark.parametrize('case', [(8, 128, 2048, 8160), (2, 7, 50, 301), (1, 64, 16384, 7332)], ids=lambda c: 'B%d_C%d_M%d_P%d' % c)
def test_gather_scale_vs_torch(case):
    """camli_gather_scale_fwd and its use as the score adjoint vs gather * score in torch (exact: one product)."""
    from camliflow_amd.csrc import fused
    b, c, m, p = case
    torch.manual_seed(sum(case))
    data = torch.randn(b, c, m, device='cuda')
    score = torch.rand(b, c, p, device='cuda').requires_grad_(True)
    idx = torch.randint(0, m, (b, p), device='cuda')
    gout = torch.randn(b, c, p, device='cuda')
    out = fused.gather_scale(data, score, idx)
    out.backward(gout)
    gathered = torch.gather(data, 2, idx[:, None, :].expand(-1, c, -1))
    assert torch.equal(out.detach(), score.detach() * gathered)
    assert torch.equal(score.grad, gout * gathered)


@pytest.mark.parametrize('case', [(8, 2048, 2048, 16, (2048, 1024, 512, 256)), (2, 1024, 300, 32, (1024, 512, 256)),
                                  (2, 2048, 700, 3, (2048, 1024)), (1, 2048, 64, 16, (2048, 1024, 512)), (2, 512, 300, 8, (512, 256)),
                                  (2, 2048, 500, 16, (2048, 1536, 512)), (3, 900, 200, 16, (900, 450)), (2, 512, 128, 3, (512, 256))],
                         ids=str)
@pytest.mark.parametrize('mode', ['lane', 'xlane', 'lane4'])
def test_knn_prefixes_equal_separate_searches(case, mode, oracle_lib, monkeypatch):
    """camli_knn_prefixes: the neighbours among the first m inputs for every nested prefix size, from one scan --
    bit-identical to one search per prefix (oracle), on a cloud with 25 % exact duplicates (ties at the k-th distance
    exercise the per-snapshot redo) and for shapes that take the one-launch kernel as well as the per-level fallback."""
    from camliflow_amd.csrc import wrapper
    b, m, nq, k, sizes = case
    if mode == 'lane4':     # round 5: at most four waves per query group -- levels below a chunk are snapshots inside wave 0's scan
        mode = 'lane'
        monkeypatch.setenv('CAMLI_KNN_PREFIX_NW', '4')
    monkeypatch.setenv('CAMLI_KNN', mode)        # lane-per-query prefix kernel / cross-lane chain kernel (round 4)
    rng = np.random.default_rng(m + k)
    inp = (rng.random((b, m, 3), dtype=np.float32) * 4).astype(np.float32)
    dup = rng.integers(0, m, size=m // 4)
    inp[:, rng.integers(0, m, size=m // 4)] = inp[:, dup]
    qry = (rng.random((b, nq, 3), dtype=np.float32) * 4).astype(np.float32)
    qry[:, :8] = inp[:, :8]
    got = wrapper.k_nearest_neighbor_prefixes(dev(inp), dev(qry), sizes, k)
    for size, idx in zip(sizes, got):
        want = oracle_lib.knn(np.ascontiguousarray(inp[:, :size]), qry, k)
        assert np.array_equal(idx.cpu().numpy(), want), size


@pytest.mark.parametrize('case', [(8, 2048, 2048, 16, (2048, 1024, 512, 256)), (2, 1024, 300, 32, (1024, 512, 256)),
                                  (2, 2048, 500, 16, (2048, 1536, 512)), (1, 2048, 64, 16, (2048, 1024, 512))], ids=str)
@pytest.mark.parametrize('prior_kind', ['same', 'moved', 'permuted', 'out_of_range', 'zeros', 'repeated'])
def test_knn_prefixes_with_prior_are_unchanged(case, prior_kind, oracle_lib):
    """camli_knn_prefixes_prior: an earlier result bounds every level's k-th distance and the scan queues nothing beyond the
    bound -- the indices must be those of the plain search (oracle) whatever the prior holds: the result of the same search,
    of the search on clouds moved since (the GRU loop's case), arbitrary different in-range candidates, indices outside the
    level, or in-range indices that REPEAT (a zeros placeholder; the nearest neighbour k times over) -- K equal candidates
    bound nothing, the kernel must notice and search unbounded.  25 % exact duplicates in the cloud, so ties at the k-th
    distance meet the bound."""
    from camliflow_amd.csrc import wrapper
    b, m, nq, k, sizes = case
    rng = np.random.default_rng(m + k + len(prior_kind))
    inp = (rng.random((b, m, 3), dtype=np.float32) * 4).astype(np.float32)
    inp[:, rng.integers(0, m, size=m // 4)] = inp[:, rng.integers(0, m, size=m // 4)]
    qry = (rng.random((b, nq, 3), dtype=np.float32) * 4).astype(np.float32)
    qry[:, :8] = inp[:, :8]
    if prior_kind == 'same':
        prior = wrapper.k_nearest_neighbor_prefixes(dev(inp), dev(qry), sizes, k)
    elif prior_kind == 'moved':
        earlier = (inp + rng.normal(0, 0.05, inp.shape)).astype(np.float32)
        prior = wrapper.k_nearest_neighbor_prefixes(dev(earlier), dev(qry), sizes, k)
    elif prior_kind == 'permuted':
        prior = [dev(np.stack([np.stack([rng.permutation(size)[:k] for _ in range(nq)]) for _ in range(b)]).astype(np.int64))
                 for size in sizes]
    elif prior_kind == 'zeros':
        prior = [dev(np.zeros((b, nq, k), dtype=np.int64)) for size in sizes]
    elif prior_kind == 'repeated':
        # the true nearest neighbour in every slot but one: the largest "bound" is then far below the k-th distance
        near = [oracle_lib.knn(np.ascontiguousarray(inp[:, :size]), qry, 2) for size in sizes]
        prior = [dev(np.concatenate([np.repeat(n2[:, :, :1], k - 1, axis=2), n2[:, :, 1:2]], axis=2).astype(np.int64)) for n2 in near]
    else:
        prior = [dev(np.full((b, nq, k), size + 5, dtype=np.int64)) for size in sizes]
    prior = [p.contiguous() for p in prior]
    got = wrapper.k_nearest_neighbor_prefixes(dev(inp), dev(qry), sizes, k, prior=prior)
    for size, idx in zip(sizes, got):
        want = oracle_lib.knn(np.ascontiguousarray(inp[:, :size]), qry, k)
        assert np.array_equal(idx.cpu().numpy(), want), (prior_kind, size)


def test_nested_pyramid_paths_match_per_level_paths():
    """Correlation3D with nested target levels (one prefix search, one multi-level gather, persistent gradient volumes)
    and the single back-warp against the per-level forms: identical forward values, gradients to float noise."""
    from camliflow_amd.cores import geometry, runtime
    from camliflow_amd.cores.raft3d import Correlation3D
    from camliflow_amd.cores.setconv import pass_cache
    from modelutils import hashed_fill_
    torch.manual_seed(0)
    mod = hashed_fill_(Correlation3D(out_channels=128, k=16)).cuda()
    b, n = 2, 1024
    xyz1 = torch.rand(b, 3, n, device='cuda') * 4
    base2 = xyz1 + torch.randn(b, 3, n, device='cuda') * 0.2
    xyzs2 = [base2[:, :, :m].contiguous() for m in (1024, 512, 256, 128)]
    flow = torch.randn(b, 3, n, device='cuda') * 0.1
    f1 = torch.randn(b, 128, n, device='cuda', requires_grad=True)
    f2 = torch.randn(b, 128, n, device='cuda', requires_grad=True)
    g = torch.randn(b, 128, n, device='cuda')
    res = {}
    with runtime.use_backend('hip'):
        for nested in (True, False):
            with pass_cache():
                mod.zero_grad()
                mod.build_cost_volume_pyramid(f1, f2, xyzs2, nested=nested)
                warped = geometry.backwarp_3d_levels(xyz1, xyzs2, flow, nested=nested)
                outs = [mod(xyz1, xyzs2), mod(xyz1, warped)]        # two lookups accumulate into the same volumes
                grads = torch.autograd.grad(outs, [f1, f2] + list(mod.parameters()), [g, 0.5 * g])
            res[nested] = ([w.clone() for w in warped], [o.detach() for o in outs], grads)
    for a, c in zip(res[True][0], res[False][0]):
        assert torch.equal(a, c)
    for a, c in zip(res[True][1], res[False][1]):
        assert torch.allclose(a, c, rtol=1e-6, atol=1e-6)
    for a, c in zip(res[True][2], res[False][2]):
        assert (a - c).norm() <= 1e-5 * c.norm() + 1e-7


def test_gru_loop_lookups_with_the_previous_neighbours_as_prior_are_unchanged(monkeypatch):
    """Correlation3D over a sequence of back-warped target clouds (what the GRU loop does): from the second lookup on the
    prefix search is bounded by the previous lookup's neighbours (camli_knn_prefixes_prior).  CAMLI_KNN_PRIOR=0 searches from
    scratch every time; the lookups must be bit-identical either way, jumpy flows included (a prior from far-away clouds only
    costs the speed-up)."""
    from camliflow_amd.cores import geometry, runtime
    from camliflow_amd.cores.raft3d import Correlation3D
    from camliflow_amd.cores.setconv import pass_cache
    from modelutils import hashed_fill_
    torch.manual_seed(1)
    mod = hashed_fill_(Correlation3D(out_channels=128, k=16)).cuda()
    b, n = 2, 2048
    xyz1 = torch.rand(b, 3, n, device='cuda') * 4
    base2 = xyz1 + torch.randn(b, 3, n, device='cuda') * 0.2
    xyzs2 = [base2[:, :, :m].contiguous() for m in (2048, 1024, 512, 256)]
    f1, f2 = torch.randn(b, 128, n, device='cuda'), torch.randn(b, 128, n, device='cuda')
    flows = [torch.randn(b, 3, n, device='cuda') * s for s in (0.0, 0.02, 0.05, 1.5, 0.05)]     # small steps and one jump
    res = {}
    with runtime.use_backend('hip'), torch.no_grad():
        for prior in ('1', '0'):
            monkeypatch.setenv('CAMLI_KNN_PRIOR', prior)
            with pass_cache():
                mod.build_cost_volume_pyramid(f1, f2, xyzs2, nested=True)
                outs = [mod(xyz1, geometry.backwarp_3d_levels(xyz1, xyzs2, fl, nested=True)) for fl in flows]
                assert (mod._prior_crosses is not None)
                mod.release()
            res[prior] = outs
    for a, c in zip(res['1'], res['0']):
        assert torch.equal(a, c)


def test_input_side_kernels_vs_oracle_golden_and_torch(golden, oracle_lib):
    """camli_pad_normalize / camli_persp2paral: exact against the oracle and the reference golden for
    the padding + normalisation; the IDS transform bit-identical to the torch composition on the same device (FPS
    downstream needs identical inputs) and within 1e-6 of the CPU values (log / divide differ in the last ulp)."""
    from camliflow_amd.csrc import fused
    from camliflow_amd.cores import geometry, runtime
    from camliflow_amd.cores.camliraft import _camera_pair, _IMAGENET_MEAN, _IMAGENET_STD
    g = golden('input_side')
    pad = [int(v) for v in g['pad']]
    i1, i2 = fused.pad_normalize(dev(g['images']), pad, _IMAGENET_MEAN, _IMAGENET_STD)
    assert np.array_equal(i1.cpu().numpy(), g['image1']) and np.array_equal(i2.cpu().numpy(), g['image2'])
    w1, w2 = oracle_lib.pad_normalize(g['images'], pad, _IMAGENET_MEAN, _IMAGENET_STD)
    assert np.array_equal(i1.cpu().numpy(), w1) and np.array_equal(i2.cpu().numpy(), w2)
    pcs, intr = dev(g['pcs']), dev(g['intrinsics'])
    persp, paral = _camera_pair(int(g['persp_hw'][0]), int(g['persp_hw'][1]), intr)
    with runtime.use_backend('hip'):
        p1, p2 = geometry.persp2paral_both(pcs, persp, paral)
    with runtime.use_backend('composed'):
        c1, c2 = geometry.persp2paral_both(pcs, persp, paral)
    assert torch.equal(p1, c1) and torch.equal(p2, c2)
    assert np.allclose(p1.cpu().numpy(), g['pc1'], rtol=1e-6, atol=1e-6) and np.allclose(p2.cpu().numpy(), g['pc2'], rtol=1e-6, atol=1e-6)


def test_projection_kernel_vs_oracle_golden_and_composition(golden, oracle_lib):
    """camli_project_pc2image (round 3; utils.py:234-259 + grid rescale): bit-exact against the oracle and the
    reference golden for the parallel camera (add, multiply), within 1 ulp-level tolerance of the CPU golden for the
    perspective camera (its divide is the one op that may differ CPU <-> GPU) and bit-identical to the torch
    composition on the same device; the cores' call sites go through the kernel under strict mode."""
    from camliflow_amd.cores import geometry, runtime
    g = golden('project_pc2image')
    gh, gw = [int(v) for v in g['grid_hw']]
    intr = dev(g['intrinsics'])
    cams = {
        'persp': {'projection_mode': 'perspective', 'sensor_h': int(g['persp_hw'][0]), 'sensor_w': int(g['persp_hw'][1]),
                  'f': intr[:, 0].contiguous(), 'cx': intr[:, 1].contiguous(), 'cy': intr[:, 2].contiguous()},
        'paral': {'projection_mode': 'parallel', 'sensor_h': int(g['paral_hw'][0]), 'sensor_w': int(g['paral_hw'][1]),
                  'cx': float(g['paral_c'][0]), 'cy': float(g['paral_c'][1])}}
    for name, cam in cams.items():
        pc = dev(g['pc_' + name])
        for grid_hw, key in ((None, 'uv_'), ((gh, gw), 'uv_grid_')):
            with runtime.use_backend('hip'):
                runtime.set_strict(True)
                try:
                    got = geometry.project_pc2image(pc, cam, grid_hw=grid_hw)
                finally:
                    runtime.set_strict(False)
            with runtime.use_backend('composed'):
                same_device = geometry.project_pc2image(pc, cam, grid_hw=grid_hw)
            assert torch.equal(got, same_device), (name, key)
            want = g[key + name]
            if name == 'paral':
                assert np.array_equal(got.cpu().numpy(), want)
            else:
                assert np.allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-5)
    # a cloud that requires grad keeps the composition (and says so)
    pc = dev(g['pc_paral']).requires_grad_(True)
    with runtime.use_backend('hip'):
        uv = geometry.project_pc2image(pc, cams['paral'], grid_hw=(gh, gw))
    assert uv.requires_grad

"""Edge cases of the C-ABI entry points: empty batches / queries, minimum and maximum sizes, bad
arguments (status code + message, never a crash), non-default streams."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_empty_batch_and_empty_queries_are_noops():
    from camliflow_amd import csrc
    x = torch.rand(0, 100, 3, device='cuda')
    assert csrc.k_nearest_neighbor(x, x, 3).shape == (0, 100, 3)
    assert csrc.furthest_point_sampling(x, 10).shape == (0, 10)
    inp = torch.rand(2, 50, 3, device='cuda')
    qry = torch.rand(2, 0, 3, device='cuda')
    # the reference's layout sniffing (shape[1] <= 3) treats a 0-row tensor as channel-first; feed channel-first
    assert csrc.k_nearest_neighbor(inp.transpose(1, 2), qry.transpose(1, 2), 4).shape == (2, 0, 4)
    out = csrc.correlation2d(torch.rand(0, 8, 5, 6, device='cuda'), torch.rand(0, 8, 5, 6, device='cuda'), 2)
    assert out.shape == (0, 25, 5, 6)


def test_knn_fewer_inputs_than_k_pads_with_index_zero(oracle_lib):
    """k_nearest_neighbor_kernel.cu:68-72,93-94: unfilled slots keep index 0"""
    from camliflow_amd import csrc
    rng = np.random.default_rng(0)
    inp = rng.random((1, 5, 3), dtype=np.float32)
    qry = rng.random((1, 9, 3), dtype=np.float32)
    got = csrc.k_nearest_neighbor(dev(inp), dev(qry), 16).cpu().numpy()
    assert np.array_equal(got, oracle_lib.knn(inp, qry, 16))
    assert (got[:, :, 5:] == 0).all()


def test_k_max_64_and_rejects_65(oracle_lib):
    from camliflow_amd import csrc
    from camliflow_amd.csrc._lib import CamliHipError
    rng = np.random.default_rng(1)
    inp = rng.random((1, 200, 3), dtype=np.float32)
    got = csrc.k_nearest_neighbor(dev(inp), dev(inp), 64).cpu().numpy()
    assert np.array_equal(got, oracle_lib.knn(inp, inp, 64))
    with pytest.raises(CamliHipError, match='k'):
        csrc.k_nearest_neighbor(dev(inp), dev(inp), 65)


def test_fps_limits(oracle_lib):
    from camliflow_amd import csrc
    from camliflow_amd.csrc._lib import CamliHipError
    rng = np.random.default_rng(2)
    xyz = rng.random((1, 24576, 3), dtype=np.float32)          # the documented maximum
    got = csrc.furthest_point_sampling(dev(xyz), 300).cpu().numpy()
    assert np.array_equal(got, oracle_lib.fps(xyz, 300))
    with pytest.raises(CamliHipError, match='register-resident'):
        csrc.furthest_point_sampling(torch.rand(1, 30000, 3, device='cuda'), 16)
    two = rng.random((1, 2, 3), dtype=np.float32)              # smallest legal cloud (N > n_samples)
    assert np.array_equal(csrc.furthest_point_sampling(dev(two), 1).cpu().numpy(), [[0]])


def test_kernels_honour_the_current_stream(oracle_lib):
    from camliflow_amd import csrc
    rng = np.random.default_rng(3)
    inp = rng.random((2, 1000, 3), dtype=np.float32)
    side = torch.cuda.Stream()
    tin = dev(inp)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        busy = torch.randn(4096, 4096, device='cuda') @ torch.randn(4096, 4096, device='cuda')   # keep `side` busy
        scaled = tin * 2.0                                       # produced on `side` ...
        got = csrc.k_nearest_neighbor(scaled, scaled, 8)         # ... and consumed by a kernel on `side`
    side.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle_lib.knn(inp * 2.0, inp * 2.0, 8))
    del busy


def test_lookup_rejects_other_radius():
    from camliflow_amd.csrc import fused
    from camliflow_amd.csrc._lib import CamliHipError
    pyr = fused.AllPairsPyramid()
    pyr.levels = [torch.randn(64, 8, 8, device='cuda')]
    pyr.shape = (1, 8, 8)
    pyr.token = torch.zeros(1, device='cuda')
    with pytest.raises(CamliHipError, match='radius'):
        fused.allpairs_lookup(pyr, torch.zeros(1, 2, 8, 8, device='cuda'), 3)


def test_kitti_shape_bf16_autocast_config5():
    """BASELINE configs[4]: KITTI-shape 1242x375 + 16384 points under bf16 autocast (convs / GEMMs only;
    KNN, FPS, correlation and CLFM stay fp32 as in the reference).  bf16 is not the fp32 parity bar:
    the flows must stay finite and within 0.5 px / 0.05 of the fp32 run (4 iterations)."""
    from camliflow_amd.cores import CamLiRAFT, runtime
    from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=4)), scale=0.5).cuda().eval()
    inputs = {k: v.cuda() for k, v in synthetic_inputs(1, 375, 1242, 16384, f=721.5, with_targets=False, zmax=90.0).items()}
    with torch.no_grad(), runtime.use_backend('hip'):
        ref = model(inputs)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            low = model(inputs)
    assert ref['flow_2d'].shape == (1, 2, 375, 1242) and ref['flow_3d'].shape == (1, 3, 16384)
    for key, tol in (('flow_2d', 0.5), ('flow_3d', 0.05)):
        assert torch.isfinite(low[key]).all()
        assert torch.linalg.norm(low[key].float() - ref[key], dim=1).mean().item() < tol, key


def test_bf16_autocast_training_step():
    """configs[4] trains under autocast: forward + backward through every custom adjoint with reduced-precision
    convolution outputs (the 1x1-convolution function sees bf16 gradients for fp32 operands); gradients are fp32,
    finite, and close in norm to the fp32 step."""
    from camliflow_amd.cores import CamLiRAFT, runtime
    from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2)), scale=0.5).cuda().train()
    inputs = {k: v.cuda() for k, v in synthetic_inputs(1, 128, 160, 4608).items()}

    def grads(autocast):
        model.zero_grad()
        with runtime.use_backend('hip'), torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            model(inputs)
            loss = model.get_loss()
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    loss32, g32 = grads(False)
    loss16, g16 = grads(True)
    assert abs(loss16 - loss32) < 0.05 * abs(loss32) + 1e-3
    assert set(g16) == set(g32)
    n32 = torch.sqrt(sum((g.double() ** 2).sum() for g in g32.values())).item()
    diff = torch.sqrt(sum(((g16[n].double() - g32[n].double()) ** 2).sum() for n in g32)).item()
    for n, g in g16.items():
        assert g.dtype == torch.float32 and torch.isfinite(g).all(), n
    assert diff < 0.25 * n32, (diff, n32)


def test_deterministic_algorithms_route_atomic_adjoints_to_torch():
    """torch.use_deterministic_algorithms is honoured -- the fused ops whose adjoints use float atomics
    (bias gradient, masked-L2 sums, SK gate, interpolation / max-pool / up-sampling scatters) switch to the torch
    composition (recorded in the census), the atomic-free kernels stay on HIP, and the step still matches."""
    from camliflow_amd.cores import CamLiRAFT, runtime
    from modelutils import camliraft_cfg, hashed_fill_, synthetic_inputs
    torch.manual_seed(0)
    model = hashed_fill_(CamLiRAFT(camliraft_cfg(n_iters=2)), scale=0.5).cuda().train()
    inputs = {k: v.cuda() for k, v in synthetic_inputs(1, 128, 160, 4608).items()}

    def run():
        model.zero_grad()
        out = model(inputs)
        loss = model.get_loss()
        loss.backward()
        return out, loss.item()

    with runtime.use_backend('hip'):
        base_out, base_loss = run()
        runtime.set_census(True)
        runtime.reset_census()
        torch.use_deterministic_algorithms(True, warn_only=True)
        try:
            det_out, det_loss = run()
        finally:
            torch.use_deterministic_algorithms(False)
            census = runtime.census()
            runtime.set_census(False)
    switched = [k for k in census['composed'] if 'deterministic' in k]
    assert any(k.startswith('bias_act') for k in switched), switched
    assert 'camli_bias_act_bwd' not in census['fused']
    # round 4: PointConvDW's adjoint is atomic-free (ordered row kernel) and stays on HIP
    assert not any(k.startswith('PointConvDW') for k in switched) and census['fused'].get('camli_pointconv_dw_bwd', 0) > 0
    assert census['fused'].get('camli_knn', 0) > 0 and census['fused'].get('camli_allpairs_lookup_bwd', 0) > 0
    assert abs(det_loss - base_loss) <= 1e-4 * max(1.0, abs(base_loss))
    for key in base_out:
        assert torch.linalg.norm(det_out[key] - base_out[key], dim=1).mean().item() <= 1e-4


def _twice(fn):
    a = fn()
    torch.cuda.synchronize()
    b = fn()
    torch.cuda.synchronize()
    return a, b


def test_adjoints_left_on_hip_under_deterministic_mode_are_bit_reproducible():
    """the kernels that STAY on HIP under torch.use_deterministic_algorithms(True) must be run-to-run
    bit-identical.  Three of them were not in round 2 -- the weight network's cross-wave merge (LDS float atomics), the
    single-level point cost-volume gather adjoint (global float atomics) and PointConv mixing at k != 16 (now routed to
    torch) -- so each atomic-free adjoint is run twice on the same inputs and compared with torch.equal."""
    import numpy as np
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.blocks import MLP2d
    from camliflow_amd.cores.setconv import PointConv
    from camliflow_amd.csrc import fused
    g = torch.Generator(device='cpu').manual_seed(5)
    b, c, m, n, k = 2, 128, 2048, 2048, 16

    # weight network adjoint: 2 x 2048 x 16 columns over many workgroups x 4 waves
    mlp = MLP2d(3, [8, 32, c], act='relu').cuda()
    xyz = (torch.rand(b, 3, m, generator=g) * 4).cuda()
    idx = torch.randint(0, m, (b, n, k), generator=g).cuda()
    gout = torch.randn(b, c, n, k, generator=g).cuda()

    def weightnet_grads():
        mlp.zero_grad()
        fused.weightnet(xyz, xyz, idx, k, mlp).backward(gout)
        return [p.grad.clone() for p in mlp.parameters()]
    first, second = _twice(weightnet_grads)
    for x, y in zip(first, second):
        assert torch.equal(x, y)

    # single-level cost-volume gather adjoint, incl. M < k (the search pads with index 0: duplicates in a row)
    for (mm, kk) in ((512, 16), (5, 16)):
        cost = torch.randn(b, 256, mm, generator=g).cuda().requires_grad_(True)
        x1 = torch.randn(b, 3, 256, generator=g).cuda()
        x2 = torch.randn(b, 3, mm, generator=g).cuda()
        from camliflow_amd import csrc
        cross = csrc.k_nearest_neighbor(x2.transpose(1, 2).contiguous(), x1.transpose(1, 2).contiguous(), kk)
        go = torch.randn(b, 4, 256, kk, generator=g).cuda()

        def gather_grad():
            return torch.autograd.grad(fused.corr3d_lookup_input(cost, x1, x2, cross), cost, go)[0]
        first, second = _twice(gather_grad)
        assert torch.equal(first, second)
        want = torch.zeros(b * 256, mm, device='cuda').index_put_(
            (torch.arange(b * 256, device='cuda')[:, None].expand(-1, kk), cross.view(b * 256, kk)),
            go[:, 3].reshape(b * 256, kk), accumulate=True).view(b, 256, mm)
        torch.testing.assert_close(first, want, rtol=1e-5, atol=1e-5)

    # PointConvDW adjoint (round 4: ordered row kernel, no float atomics): twice the same bits, both gradients, on neighbour
    # tables with heavy collisions (few distinct targets) as well as the usual ones; and equal to the round-3 atomic
    # kernel up to fp32 summation order
    for targets in (m, 37):
        featd = torch.randn(b, c, m, generator=g).cuda()
        goutd = torch.randn(b, c, n, generator=g).cuda()
        wsel = torch.randn(b, c, n, generator=g).cuda()
        msel = torch.randint(0, targets, (b, c, n), generator=g, dtype=torch.int32).cuda()
        lib = fused._lib.load()

        def dw_bwd(entry=lib.camli_pointconv_dw_bwd_ordered):
            gfeat, gwsel = torch.empty_like(featd), torch.empty_like(goutd)
            fused._lib.check(entry(goutd.data_ptr(), featd.data_ptr(), wsel.data_ptr(), msel.data_ptr(), gfeat.data_ptr(),
                                   gwsel.data_ptr(), b, c, m, n, torch.cuda.current_stream().cuda_stream), 'camli_pointconv_dw_bwd')
            return gfeat, gwsel
        first, second = _twice(dw_bwd)
        assert torch.equal(first[0], second[0]) and torch.equal(first[1], second[1])
        atomic = dw_bwd(lib.camli_pointconv_dw_bwd)
        assert torch.equal(first[1], atomic[1])
        torch.testing.assert_close(first[0], atomic[0], rtol=1e-4, atol=1e-4)
        want = torch.zeros(b * c, m, device='cuda').index_put_(
            (torch.arange(b * c, device='cuda')[:, None].expand(-1, n), msel.view(b * c, n).long()),
            (goutd * wsel).view(b * c, n), accumulate=True).view(b, c, m)
        torch.testing.assert_close(first[0], want, rtol=1e-4, atol=1e-4)

    # PointConv mixing: k = 16 stays on HIP (sorted adjoint) and is reproducible; k = 8 leaves HIP in deterministic mode
    feat = torch.randn(b, 32, m, generator=g).cuda().requires_grad_(True)
    for kk, stays in ((16, True), (8, False)):
        conv = PointConv(32, 64, k=kk).cuda()

        def pointconv_grads():
            conv.zero_grad()
            out = conv(xyz, feat)
            return [torch.autograd.grad(out.square().sum(), feat, retain_graph=True)[0]]
        with runtime.use_backend('hip'):
            runtime.set_census(True)
            runtime.reset_census()
            torch.use_deterministic_algorithms(True, warn_only=True)
            try:
                first, second = _twice(pointconv_grads)
            finally:
                torch.use_deterministic_algorithms(False)
                census = runtime.census()
                runtime.set_census(False)
        assert (census['fused'].get('camli_pointconv_mix_bwd', 0) > 0) == stays, census
        if stays:
            assert torch.equal(first[0], second[0])


def test_nan_to_num_folded_into_blend_and_epilogue():
    """Round 3: the GRU's closing torch.nan_to_num (raft_core.py:138) rides on camli_gru_blend, the motion encoder's
    (raft_core.py:163-164) on the bias/ReLU epilogue (code 5).  Values AND gradients must follow torch's composition on
    inputs that contain NaN and +-inf (nan_to_num passes no gradient where its input was not finite)."""
    from camliflow_amd.csrc import fused
    g = torch.Generator().manual_seed(11)
    b, c, h, w = 2, 8, 6, 10
    bad = torch.tensor([float('nan'), float('inf'), float('-inf')])

    def poisoned(*shape):
        t = torch.randn(*shape, generator=g)
        flat = t.view(-1)
        pos = torch.randperm(flat.numel(), generator=g)[:9]
        flat[pos] = bad.repeat(3)
        return t.cuda()
    # ---- blend
    pre, ctx, hh = poisoned(b, c, h, w), torch.randn(b, c, h, w, generator=g).cuda(), poisoned(b, c, h, w)
    z = torch.rand(b, c, h, w, generator=g).cuda()
    go = torch.randn(b, c, h, w, generator=g).cuda()
    leaves = [t.clone().requires_grad_(True) for t in (pre, ctx, z, hh)]
    out = fused.gru_blend(*leaves, nan_to_num=True)
    grads = torch.autograd.grad(out, leaves, go)
    ref_leaves = [t.clone().requires_grad_(True) for t in (pre, ctx, z, hh)]
    p_, c_, z_, h_ = ref_leaves
    ref = torch.nan_to_num((1 - z_) * h_ + z_ * torch.tanh(p_ + c_))
    ref_grads = torch.autograd.grad(ref, ref_leaves, go)
    assert torch.isfinite(out).all() and torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    finite_out = torch.isfinite((1 - z) * hh + z * torch.tanh(pre + ctx))
    for got, want in zip(grads, ref_grads):
        # where the un-sanitised output is finite both are ordinary numbers; elsewhere both must be exactly zero or
        # agree as NaN-free values (torch multiplies the zeroed gradient by finite local derivatives)
        assert torch.allclose(got[finite_out], want[finite_out], rtol=1e-4, atol=1e-6)
        assert (torch.nan_to_num(got[~finite_out]) == 0).all()
    # ---- epilogue: relu + nan_to_num
    x0, bias = poisoned(b, c, h, w), torch.randn(c, generator=g).cuda()
    xa, ba = x0.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ya = fused.bias_act(xa * 1.0, ba, 'relu_nan_to_num')
    gxa, gba = torch.autograd.grad(ya, [xa, ba], go)
    xb, bb = x0.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yb = torch.nan_to_num(torch.relu(xb + bb.view(1, c, 1, 1)))
    gxb, gbb = torch.autograd.grad(yb, [xb, bb], go)
    assert torch.isfinite(ya).all() and torch.equal(ya, yb)
    assert torch.equal(gxa, gxb)
    assert torch.allclose(gba, gbb, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('shape', [(2, 16, 12, 20), (1, 5, 3, 7), (3, 64, 34, 60)], ids=str)
def test_residual_epilogue_vs_torch(shape):
    """camli_bias_act_res_fwd (round 3): relu(conv_out + bias + shortcut) in one pass -- values equal to the three torch ops,
    the gradient reaches the convolution output AND the shortcut, the bias gradient is the masked sum."""
    from camliflow_amd.csrc import fused
    g = torch.Generator().manual_seed(sum(shape))
    x0, r0 = torch.randn(*shape, generator=g).cuda(), torch.randn(*shape, generator=g).cuda()
    b0, go = torch.randn(shape[1], generator=g).cuda(), torch.randn(*shape, generator=g).cuda()
    for act in ('relu', None):
        xa, ra, ba = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        ya = fused.bias_act_res(xa * 1.0, ba, ra, act)
        ga = torch.autograd.grad(ya, [xa, ra, ba], go)
        xb, rb, bb = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        yb = xb + bb.view(1, -1, 1, 1) + rb
        yb = torch.relu(yb) if act else yb
        gb = torch.autograd.grad(yb, [xb, rb, bb], go)
        assert torch.allclose(ya, yb, rtol=1e-6, atol=1e-6)
        assert torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1])
        assert torch.allclose(ga[2], gb[2], rtol=1e-4, atol=1e-4)


def test_resnet_trunk_fused_residual_vs_composed():
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.raft2d import Encoder2D
    from modelutils import hashed_fill_
    torch.manual_seed(0)
    enc = hashed_fill_(Encoder2D(50), scale=0.5).cuda().train()
    x = torch.randn(2, 3, 64, 96, device='cuda')
    res = {}
    for backend in ('composed', 'hip'):
        enc.zero_grad()
        with runtime.use_backend(backend):
            runtime.set_census(True)
            runtime.reset_census()
            out = enc(x)
            out.square().mean().backward()
            census = runtime.census()
            runtime.set_census(False)
        res[backend] = (out.detach(), {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None})
    assert census['fused'].get('camli_bias_act_fwd', 0) > 7
    assert torch.allclose(res['hip'][0], res['composed'][0], rtol=1e-4, atol=1e-4)
    num = sum(((res['hip'][1][n] - res['composed'][1][n]).double() ** 2).sum().item() for n in res['hip'][1]) ** 0.5
    den = sum((res['composed'][1][n].double() ** 2).sum().item() for n in res['hip'][1]) ** 0.5
    assert res['hip'][1].keys() == res['composed'][1].keys() and num / den < 1e-3, num / den


@pytest.mark.parametrize('shape', [(2, 8, 64, 96), (1, 3, 17, 33), (1, 2, 1, 1), (2, 4, 2, 300), (1, 1, 272, 480)], ids=str)
def test_maxpool3x3s2_vs_torch(shape):
    """camli_maxpool3x3s2 (the ResNet stem's pooling): values and gradients equal torch's, including ties (post-ReLU zeros:
    the FIRST maximum in row-major window order takes the gradient), odd sizes and NaN propagation."""
    from camliflow_amd.csrc import fused
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sum(shape))
    x0 = torch.relu(torch.randn(*shape, generator=g)).cuda()          # many exact ties at 0
    if x0.numel() > 50:
        x0.view(-1)[7] = float('nan')
    xa = x0.clone().requires_grad_(True)
    ya = fused.maxpool3x3s2(xa)
    xb = x0.clone().requires_grad_(True)
    yb = F.max_pool2d(xb, 3, 2, 1)
    go = torch.randn(yb.shape, generator=g).cuda()
    assert ya.shape == yb.shape and torch.equal(torch.nan_to_num(ya, nan=-7.0), torch.nan_to_num(yb, nan=-7.0))
    ga, gb = torch.autograd.grad(ya, xa, go)[0], torch.autograd.grad(yb, xb, go)[0]
    assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-6), (ga - gb).abs().max().item()


@pytest.mark.parametrize('shape', [(2, 64, 12, 20), (1, 4, 3, 7), (3, 256, 17, 30), (1, 1024, 2, 3)], ids=str)
def test_channels_last_epilogue_vs_torch(shape):
    """The channels-last bias / activation / residual epilogue of the ResNet trunk (round 3) against the torch ops on the
    same channels-last tensors: values, the gradient of the convolution output and of the shortcut, the bias gradient; the
    results stay channels-last."""
    from camliflow_amd.csrc import fused
    g = torch.Generator().manual_seed(sum(shape))
    cl = torch.channels_last
    x0 = torch.randn(*shape, generator=g).cuda().contiguous(memory_format=cl)
    r0 = torch.randn(*shape, generator=g).cuda().contiguous(memory_format=cl)
    b0, go = torch.randn(shape[1], generator=g).cuda(), torch.randn(*shape, generator=g).cuda().contiguous(memory_format=cl)
    for act in ('relu', None):
        for with_res in (True, False):
            xa, ra, ba = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            ya = fused.bias_act_res(xa * 1.0, ba, ra, act) if with_res else fused.bias_act(xa * 1.0, ba, act)
            assert ya.is_contiguous(memory_format=cl)
            ins_a = [xa, ra, ba] if with_res else [xa, ba]
            ga = torch.autograd.grad(ya, ins_a, go)
            xb, rb, bb = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            yb = xb + bb.view(1, -1, 1, 1) + (rb if with_res else 0.0)
            yb = torch.relu(yb) if act else yb
            gb = torch.autograd.grad(yb, [xb, rb, bb] if with_res else [xb, bb], go)
            assert torch.allclose(ya, yb, rtol=1e-6, atol=1e-6)
            for u, v in zip(ga[:-1], gb[:-1]):
                assert torch.equal(u, v)
            assert torch.allclose(ga[-1], gb[-1], rtol=1e-4, atol=1e-4)

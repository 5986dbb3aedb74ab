"""Fused PointConvDW core (camli_pointconv_dw_fwd/bwd) against the oracle, and the PointConvDW
module under the 'hip' backend against its torch-composed formulation (fp32, tolerances stated)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(2, 128, 2048, 2048, 16, 32), (1, 125, 2048, 2048, 16, 32), (3, 32, 500, 300, 32, 32),
                                  (2, 128, 2048, 2048, 4, 32), (1, 7, 100, 77, 5, 9), (1, 16, 64, 64, 16, 16),
                                  # shared-row kernel: partial last workgroup / tile, M > 2048 and > 4096 row loads, odd M falls back
                                  (2, 20, 4100, 1100, 8, 8), (1, 9, 8192, 700, 16, 16), (1, 12, 2047, 513, 32, 32), (2, 5, 4096, 65, 4, 4)],
                         ids=lambda c: 'B%d_C%d_M%d_N%d_k%d_kk%d' % c)
def test_core_fwd_bwd_vs_oracle(case, oracle_lib):
    from camliflow_amd.csrc import fused
    b, c, m, n, k, kk = case
    rng = np.random.default_rng(sum(case))
    feat = rng.standard_normal((b, c, m)).astype(np.float32)
    weight = np.maximum(rng.standard_normal((b, c, n, k)), 0).astype(np.float32)   # post-ReLU like weight_net
    idx = rng.integers(0, m, size=(b, n, kk)).astype(np.int64)
    gout = rng.standard_normal((b, c, n)).astype(np.float32)

    tf = torch.from_numpy(feat).cuda().requires_grad_(True)
    tw = torch.from_numpy(weight).cuda().requires_grad_(True)
    shared = fused.SharedSetConvWeights(tw)
    out = fused.pointconv_dw(tf, shared, torch.from_numpy(idx).cuda(), k)
    out.backward(torch.from_numpy(gout).cuda())

    want, arg = oracle_lib.pointconv_dw_fwd(feat, weight, idx, k)
    assert np.array_equal(out.detach().cpu().numpy(), want)     # same products, same max -> bit exact
    gfeat, gweight = oracle_lib.pointconv_dw_bwd(gout, feat, weight, idx, arg, k)
    assert np.allclose(tf.grad.cpu().numpy(), gfeat, rtol=1e-5, atol=1e-5)   # atomics reorder the sums
    assert np.array_equal(tw.grad.cpu().numpy(), gweight)


@pytest.mark.parametrize('cfg', [(128, 128, 16), (3, 32, 32), (384, 128, 4), (144, 125, 16)])
def test_module_hip_vs_composed(cfg):
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.setconv import PointConvDW, pass_cache
    from camliflow_amd.csrc import k_nearest_neighbor
    cin, cout, k = cfg
    torch.manual_seed(cin)
    mod = PointConvDW(cin, cout, act=None if k == 4 else 'leaky_relu', k=k).cuda()
    xyz = torch.rand(2, 3, 1024, device='cuda') * 4
    feats = [torch.randn(2, cin, 1024, device='cuda', requires_grad=True) for _ in range(3)]
    knn = k_nearest_neighbor(xyz, xyz, 32)
    gouts = [torch.randn(2, cout, 1024, device='cuda') for _ in range(3)]
    res = {}
    for backend in ('hip', 'composed'):
        mod.zero_grad()
        for f in feats:
            f.grad = None
        with runtime.use_backend(backend), pass_cache():
            outs = [mod(xyz, f, knn_indices=knn) for f in feats]        # three "iterations" share the weights
            sum((o * g).sum() for o, g in zip(outs, gouts)).backward()
        res[backend] = ([o.detach() for o in outs], [f.grad.clone() for f in feats],
                        {n: p.grad.clone() for n, p in mod.named_parameters()})
    for a, b_ in zip(res['hip'][0], res['composed'][0]):
        assert torch.allclose(a, b_, rtol=1e-5, atol=1e-5)
    for a, b_ in zip(res['hip'][1], res['composed'][1]):
        assert torch.allclose(a, b_, rtol=1e-4, atol=1e-4)
    for name in res['hip'][2]:
        a, b_ = res['hip'][2][name], res['composed'][2][name]
        assert (a - b_).norm() <= 1e-4 * b_.norm() + 1e-5, name


@pytest.mark.parametrize('case', [(2, 8192, 4096, 99, 16, 16), (2, 4096, 2048, 131, 16, 16), (1, 300, 200, 19, 16, 32),
                                  (1, 256, 256, 198, 16, 16), (1, 64, 40, 7, 5, 9)], ids=str)
def test_pointconv_mix_vs_oracle_and_composed(case, oracle_lib):
    """fused gather+matmul (camli_pointconv_mix_fwd/bwd): forward vs the C oracle, gradients vs the
    torch composition the reference uses (gather, matmul)."""
    from camliflow_amd.csrc import fused
    b, m, n, ch, k, kk = case
    rng = np.random.default_rng(sum(case))
    feat = rng.standard_normal((b, m, ch)).astype(np.float32)
    wgt = rng.standard_normal((b, 16, n, k)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, kk)).astype(np.int64)
    tf = torch.from_numpy(feat).cuda().requires_grad_(True)
    tw = torch.from_numpy(wgt).cuda().requires_grad_(True)
    ti = torch.from_numpy(idx).cuda()
    out = fused.pointconv_mix(tf, tw, ti, k)
    want = oracle_lib.pointconv_mix_fwd(feat, wgt, idx, k)
    assert np.allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    gout = torch.randn_like(out)
    out.backward(gout)
    gf, gw = tf.grad.clone(), tw.grad.clone()
    tf.grad = tw.grad = None
    rows = torch.arange(b, device='cuda').view(b, 1, 1).expand(b, n, k)
    ref = torch.matmul(tw.transpose(1, 2), tf[rows, ti[:, :, :k], :])
    ref.backward(gout)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(gf, tf.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(gw, tw.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('case', [(2, 128, 2048, 2048, 16, 32), (1, 125, 2048, 2048, 32, 32), (3, 32, 500, 300, 8, 32),
                                  (2, 20, 4100, 1100, 4, 8), (1, 9, 8192, 700, 16, 16), (1, 5, 64, 1, 4, 4)],
                         ids=lambda c: 'B%d_C%d_M%d_N%d_k%d_kk%d' % c)
def test_k_major_layout_is_the_same_operator(case, oracle_lib):
    """Round 3: the product path keeps the neighbour weights k-major, [B,C,k,N].  Forward (values AND arg-max: same
    products, same first-maximum rule), feature gradient and the expanded dense weight gradient must equal the
    [B,C,N,k] kernels' results after the permutation, and the oracle's."""
    from camliflow_amd.csrc import fused
    b, c, m, n, k, kk = case
    rng = np.random.default_rng(sum(case) + 1)
    feat = rng.standard_normal((b, c, m)).astype(np.float32)
    weight = np.maximum(rng.standard_normal((b, c, n, k)), 0).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, kk)).astype(np.int64)
    gout = rng.standard_normal((b, c, n)).astype(np.float32)
    res = {}
    for k_major in (False, True):
        tf = torch.from_numpy(feat).cuda().requires_grad_(True)
        w_np = np.ascontiguousarray(weight.transpose(0, 1, 3, 2)) if k_major else weight
        tw = torch.from_numpy(w_np).cuda().requires_grad_(True)
        shared = fused.SharedSetConvWeights(tw, k_major=k_major)
        out = fused.pointconv_dw(tf, shared, torch.from_numpy(idx).cuda(), k)
        out.backward(torch.from_numpy(gout).cuda())
        gw = tw.grad.permute(0, 1, 3, 2) if k_major else tw.grad
        res[k_major] = (out.detach().cpu().numpy(), tf.grad.cpu().numpy(), gw.cpu().numpy())
    want, arg = oracle_lib.pointconv_dw_fwd(feat, weight, idx, k)
    gfeat, gweight = oracle_lib.pointconv_dw_bwd(gout, feat, weight, idx, arg, k)
    for k_major in (False, True):
        assert np.array_equal(res[k_major][0], want)
        assert np.allclose(res[k_major][1], gfeat, rtol=1e-5, atol=1e-5)
        assert np.array_equal(res[k_major][2], gweight)


@pytest.mark.parametrize('case', [(2, 128, 2048, 2048, 16), (3, 32, 500, 300, 32), (1, 125, 2048, 2048, 4)],
                         ids=lambda c: 'B%d_C%d_M%d_N%d_k%d' % c)
def test_core_adjoint_reads_a_sliced_output_gradient_in_place(case, oracle_lib):
    """The set-conv outputs are concatenated downstream (GRU3D, MotionEncoder3D): the adjoint of that cat hands
    camli_pointconv_dw_bwd_strided a channel slice of a wider gradient.  Same gradients as the oracle (C = 125: the slice is not
    16-byte aligned for every batch entry -> the wrapper copies it, the dense kernel runs)."""
    from camliflow_amd.csrc import fused
    b, c, m, n, k = case
    rng = np.random.default_rng(sum(case) + 7)
    feat = rng.standard_normal((b, c, m)).astype(np.float32)
    weight = np.maximum(rng.standard_normal((b, c, n, k)), 0).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, k)).astype(np.int64)
    gwide = rng.standard_normal((b, c + 8, n)).astype(np.float32)
    tf = torch.from_numpy(feat).cuda().requires_grad_(True)
    tw = torch.from_numpy(weight).cuda().requires_grad_(True)
    shared = fused.SharedSetConvWeights(tw)
    out = fused.pointconv_dw(tf, shared, torch.from_numpy(idx).cuda(), k)
    side = torch.zeros(b, 4, n, device='cuda', requires_grad=True)
    (torch.cat([side, out, side], dim=1) * torch.from_numpy(gwide).cuda()).sum().backward()
    _, arg = oracle_lib.pointconv_dw_fwd(feat, weight, idx, k)
    gfeat, gweight = oracle_lib.pointconv_dw_bwd(np.ascontiguousarray(gwide[:, 4:4 + c]), feat, weight, idx, arg, k)
    assert np.allclose(tf.grad.cpu().numpy(), gfeat, rtol=1e-5, atol=1e-5)
    assert np.array_equal(tw.grad.cpu().numpy(), gweight)

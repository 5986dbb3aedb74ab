"""oracle/dense.py (numpy restatement, forward + hand-written adjoints, of the dense pieces the product path runs on its
round-3 kernels) against tensors recorded from the REFERENCE's own modules with autograd
(tests/golden/dense_*.npz, tests/golden/make_dense_golden.py).  CPU only."""
import numpy as np

from oracle import dense


def _close(a, b, rtol=2e-5, atol=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.allclose(a, b, rtol=rtol, atol=atol * max(1.0, np.abs(b).max())), np.abs(a - b).max()


def test_cost_mlp_forward_and_adjoint_vs_reference_cost_mlp(golden):
    g = golden('dense_cost_mlp')
    levels = int(g['levels'])
    params = (g['w1'], g['b1'], g['w2'], g['b2'])
    _close(dense.cost_mlp_fwd(g['lookup'], *params, levels), g['out'])
    gx, gw1, gb1, gw2, gb2 = dense.cost_mlp_bwd(g['gout'], g['lookup'], *params, levels)
    for got, key in ((gx, 'glookup'), (gw1, 'gw1'), (gb1, 'gb1'), (gw2, 'gw2'), (gb2, 'gb2')):
        _close(got, g[key], rtol=1e-4, atol=1e-5)


def test_conv3x3_forward_and_adjoint_vs_reference_flow_head(golden):
    g = golden('dense_flow_head')
    y = dense.conv3x3_fwd(g['x'], g['w'], g['b'])
    _close(y, g['y'])
    _close(np.nan_to_num(y), g['head_out'])              # raft_core.py:180-181: .float() + nan_to_num
    gx, gw, gb = dense.conv3x3_bwd(g['gy'], g['x'], g['w'])
    _close(gx, g['gx'], rtol=1e-4, atol=1e-5)
    _close(gw, g['gw'], rtol=1e-4, atol=1e-5)
    _close(gb, g['gb'], rtol=1e-4, atol=1e-5)


def test_allpairs_pyramid_forward_and_adjoint_vs_reference_build(golden):
    for tag in ('even', 'odd'):
        g = golden('dense_allpairs_' + tag)
        pyr = dense.allpairs_pyramid_fwd(g['f1'], g['f2'], 4)
        for lvl, p in enumerate(pyr):
            _close(p, g['pyr%d' % lvl], rtol=1e-4, atol=1e-5)
        gf1, gf2 = dense.allpairs_pyramid_bwd([g['gpyr%d' % lvl] for lvl in range(4)], g['f1'], g['f2'])
        _close(gf1, g['gf1'], rtol=1e-4, atol=1e-5)
        _close(gf2, g['gf2'], rtol=1e-4, atol=1e-5)


def test_resnet_glue_vs_torch_modules_of_the_call_site(golden):
    g = golden('dense_resnet_glue')
    y, arg = dense.maxpool3x3s2_fwd(g['pool_x'])
    assert np.array_equal(y, g['pool_y'])
    gx = dense.maxpool3x3s2_bwd(g['pool_gy'], arg, g['pool_x'].shape[-2:])
    # a window whose maximum is a tie of zeros (post-ReLU input) may route its gradient to another zero than torch does;
    # those positions feed ReLU's flat side in the trunk -- compare where the input is positive, and the total mass
    pos = g['pool_x'] > 0
    _close(gx[pos], g['pool_gx'][pos])
    assert abs(gx.sum() - g['pool_gx'].sum()) <= 1e-4 * np.abs(g['pool_gx']).sum()
    out = dense.bias_act_res_fwd(g['conv_out'], g['bias'], g['identity'])
    _close(out, g['out'])
    gconv, gbias = dense.bias_act_res_bwd(g['gout'], out)
    _close(gconv, g['gconv'])
    _close(gbias, g['gbias'], rtol=1e-4, atol=1e-5)


def test_conv5_and_gate_arithmetic_vs_reference_gru2d(golden):
    """oracle/dense.conv5_fwd against every 1x5 / 5x1 convolution the reference's GRU2D ran, and the whole update (conv5 +
    the gate arithmetic of oracle/glue.py) against its new hidden state."""
    from oracle import glue
    g = golden('dense_gru2d')
    for name in ('convz1', 'convr1', 'convq1', 'convz2', 'convr2', 'convq2'):
        _close(dense.conv5_fwd(g[name + '_in'], g[name + '_w'], g[name + '_b']), g[name + '_out'], rtol=1e-5, atol=1e-6)
    h, x = g['h0'], g['x']
    for suffix in ('1', '2'):
        hx = np.concatenate([h, x], axis=1)
        pre_zr = np.concatenate([dense.conv5_fwd(hx, g['convz' + suffix + '_w'], g['convz' + suffix + '_b']),
                                 dense.conv5_fwd(hx, g['convr' + suffix + '_w'], g['convr' + suffix + '_b'])], axis=1)
        z, rh, _ = glue.gru_gates_fwd(pre_zr, np.zeros_like(pre_zr), h)
        pre_q = dense.conv5_fwd(np.concatenate([rh, x], axis=1), g['convq' + suffix + '_w'], g['convq' + suffix + '_b'])
        h, _ = glue.gru_blend_fwd(pre_q, np.zeros_like(pre_q), z, h, nan_to_num=(suffix == '2'))
    _close(h, g['out'], rtol=1e-5, atol=1e-6)


def test_point_volume_pyramid_and_channel_last_gather_vs_reference(golden, oracle_lib):
    """camliraft_l_core.py:51-60 (build + its autograd adjoint) and utils.py:85-104 (channel-last batch_indexing)."""
    g = golden('dense_point_volume')
    parents = [g['parents%d' % lvl] for lvl in range(3)]
    # the reference's k_nearest_neighbor on the recorded level clouds gives the recorded tables (KNN oracle, fp32 ties aside)
    for lvl in range(3):
        fine = np.ascontiguousarray(g['xyz%d' % lvl].transpose(0, 2, 1))
        coarse = np.ascontiguousarray(g['xyz%d' % (lvl + 1)].transpose(0, 2, 1))
        assert np.array_equal(oracle_lib.knn(fine, coarse, 3), parents[lvl])
    pyr = dense.point_volume_pyramid_fwd(g['f1'], g['f2'], parents)
    for lvl, p in enumerate(pyr):
        _close(p, g['pyr%d' % lvl], rtol=1e-4, atol=1e-5)
    gf1, gf2 = dense.point_volume_pyramid_bwd([g['gpyr%d' % lvl] for lvl in range(4)], g['f1'], g['f2'], parents)
    _close(gf1, g['gf1'], rtol=1e-4, atol=1e-5)
    _close(gf2, g['gf2'], rtol=1e-4, atol=1e-5)
    idx = g['cl_idx'].reshape(g['cl_idx'].shape[0], -1)
    assert np.array_equal(oracle_lib.gather_cl(g['cl_data'], idx).reshape(g['cl_out'].shape), g['cl_out'])
    _close(oracle_lib.scatter_add_cl(g['cl_gout'].reshape(idx.shape + (-1,)), idx, g['cl_data'].shape[1]), g['cl_gdata'])
    assert np.array_equal(oracle_lib.gather_cl(g['cl2_data'], g['cl2_idx']), g['cl2_out'])
    _close(oracle_lib.scatter_add_cl(g['cl2_gout'], g['cl2_idx'], g['cl2_data'].shape[1]), g['cl2_gdata'])


def test_conv_taps_forward_and_adjoints_vs_reference_gru2d_and_torch_conv2d(golden):
    """oracle/dense.conv_taps_fwd / _bwd (the oracle of camli_convcl_fwd / _wrw): the forward against every 1x5 / 5x1 convolution
    the reference's GRU2D ran (tests/golden/dense_gru2d.npz, models/raft_core.py:110-140), forward and both adjoints against
    torch's own conv2d + autograd in fp64 -- what nn.Conv2d of the reference's modules computes -- for the kernel shapes of the
    update block (1x5, 5x1, 3x3, 7x7, 1x1)."""
    import torch
    g = golden('dense_gru2d')
    for name in ('convz1', 'convr1', 'convq1', 'convz2', 'convr2', 'convq2'):
        w = g[name + '_w']
        pad = (w.shape[2] // 2, w.shape[3] // 2)
        _close(dense.conv_taps_fwd(g[name + '_in'], w, pad) + g[name + '_b'][None, :, None, None], g[name + '_out'], rtol=1e-5, atol=1e-6)
    rng = np.random.default_rng(5)
    for kh, kw in ((1, 5), (5, 1), (3, 3), (7, 7), (1, 1)):
        x = rng.standard_normal((2, 6, 9, 11)).astype(np.float32)
        w = rng.standard_normal((4, 6, kh, kw)).astype(np.float32)
        gy = rng.standard_normal((2, 4, 9, 11)).astype(np.float32)
        pad = (kh // 2, kw // 2)
        xt = torch.from_numpy(x).double().requires_grad_()
        wt = torch.from_numpy(w).double().requires_grad_()
        yt = torch.nn.functional.conv2d(xt, wt, padding=pad)
        yt.backward(torch.from_numpy(gy).double())
        _close(dense.conv_taps_fwd(x, w, pad), yt.detach().numpy(), rtol=1e-5, atol=1e-6)
        gx, gw = dense.conv_taps_bwd(gy, x, w, pad)
        _close(gx, xt.grad.numpy(), rtol=1e-5, atol=1e-6)
        _close(gw, wt.grad.numpy(), rtol=1e-5, atol=1e-6)

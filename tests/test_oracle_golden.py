"""The oracle (oracle/camli_oracle.c) against the golden vectors generated from the reference's own
Python path (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def _names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz')))


@pytest.mark.parametrize('name', _names('knn_'))
def test_knn_matches_reference_fallback_on_safe_queries(name, golden, oracle_lib):
    if name.startswith('knn_interpolation'):
        pytest.skip('not a knn index fixture')
    g = golden(name)
    idx = oracle_lib.knn(g['input'], g['query'], int(g['k']))
    safe = g['safe']
    assert safe.mean() > 0.9  # the mask must not hollow the test out
    assert np.array_equal(idx[safe], g['indices'][safe])
    # unsafe queries: same neighbour SET up to distance ties -> compare sorted fp64 distances
    d = ((g['query'][:, :, None, :].astype(np.float64) - g['input'][:, None, :, :]) ** 2).sum(-1)
    mine = np.take_along_axis(d, idx, axis=2)
    ref = np.take_along_axis(d, g['indices'], axis=2)
    assert np.allclose(mine, ref, rtol=1e-3, atol=1e-9)


@pytest.mark.parametrize('name', ['fps_a', 'fps_b', 'fps_dup'])
def test_fps_matches_reference_fallback(name, golden, oracle_lib):
    g = golden(name)
    idx = oracle_lib.fps(g['xyz'], int(g['n_samples']))
    assert np.array_equal(idx, g['indices'])
    # prefix property the point-cloud pyramid relies on (models/utils.py:121-125)
    half = oracle_lib.fps(g['xyz'], int(g['n_samples']) // 2)
    assert np.array_equal(half, idx[:, :half.shape[1]])


@pytest.mark.parametrize('name', ['corr2d_a', 'corr2d_b', 'corr2d_c'])
def test_correlation_matches_reference_fallback(name, golden, oracle_lib):
    g = golden(name)
    md = int(g['md'])
    in1 = np.ascontiguousarray(g['input1'].transpose(0, 2, 3, 1))
    in2 = np.ascontiguousarray(g['input2'].transpose(0, 2, 3, 1))
    out = oracle_lib.corr2d_fwd(in1, in2, md)
    # reference's own criterion: mean-abs < 1e-6 (correlation_test.cpp:82-89)
    assert np.abs(out - g['output']).mean() < 1e-6
    assert np.abs(out - g['output']).max() < 1e-5
    g1, g2 = oracle_lib.corr2d_bwd(g['grad_output'], in1, in2, md)
    assert np.abs(g1.transpose(0, 3, 1, 2) - g['grad1']).mean() < 1e-6
    assert np.abs(g2.transpose(0, 3, 1, 2) - g['grad2']).mean() < 1e-6


def test_gather_and_scatter(golden, oracle_lib):
    g = golden('batch_indexing')
    data, idx = g['data'], g['indices']
    flat = idx.reshape(idx.shape[0], -1)
    out = oracle_lib.gather_cf(data, flat).reshape(g['out_cf'].shape)
    assert np.array_equal(out, g['out_cf'])
    assert np.array_equal(out.transpose(0, 2, 3, 1), g['out_cl'])
    # adjoint: <gather(x), y> == <x, scatter_add(y)>
    rng = np.random.default_rng(0)
    y = rng.standard_normal(out.reshape(data.shape[0], data.shape[1], -1).shape).astype(np.float32)
    lhs = (out.reshape(y.shape) * y).sum()
    rhs = (data * oracle_lib.scatter_add_cf(y, flat, data.shape[2])).sum()
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_knn_interpolation(golden, oracle_lib):
    g = golden('knn_interpolation')
    out = oracle_lib.knn_interp_fwd(g['in_xyz'], g['feat'], g['q_xyz'], g['knn'])
    assert np.allclose(out, g['out'], rtol=1e-5, atol=1e-5)


def _avg_pool2(v):
    n, h, w = v.shape
    h2, w2 = h // 2, w // 2
    v = v[:, :h2 * 2, :w2 * 2].reshape(n, h2, 2, w2, 2)
    return v.mean(axis=(2, 4), dtype=np.float32).astype(np.float32)


def build_levels_numpy(g):
    """all-pairs pyramid from the fixture's fmaps, restating models/raft_core.py:52-68 in numpy"""
    f1, f2 = g['fmap1'], g['fmap2']
    b, _, h, w = f1.shape
    wgt = g['aligner_weight'].reshape(256, 128)
    a1 = np.einsum('oc,bcp->bop', wgt, f1.reshape(b, 128, -1)) + g['aligner_bias'][None, :, None]
    a2 = np.einsum('oc,bcp->bop', wgt, f2.reshape(b, 128, -1)) + g['aligner_bias'][None, :, None]
    vol = np.einsum('bcp,bcq->bpq', a1, a2).astype(np.float32) / np.float32(16.0)
    levels = [vol.reshape(b * h * w, h, w)]
    for _ in range(3):
        levels.append(_avg_pool2(levels[-1]))
    return levels


@pytest.mark.parametrize('name', ['allpairs_even', 'allpairs_odd'])
def test_allpairs_lookup(name, golden, oracle_lib):
    g = golden(name)
    if 'level0' in g.files:
        levels = [g['level%d' % l].reshape(-1, *g['level%d' % l].shape[-2:]) for l in range(4)]
        mine = build_levels_numpy(g)
        for a, b in zip(levels, mine):
            assert np.allclose(a, b, rtol=1e-4, atol=1e-4)
    else:
        levels = build_levels_numpy(g)
    out = oracle_lib.allpairs_lookup_fwd(levels, g['coords'], 4)
    assert out.shape == g['out'].shape
    assert np.allclose(out, g['out'], rtol=1e-4, atol=2e-4)
    gl = oracle_lib.allpairs_lookup_bwd([l.shape for l in levels], g['coords'], g['grad_out'], 4)
    if 'glevel0' in g.files:
        # autograd reports TOTAL derivatives: level l also feeds level l+1 through avg_pool2d
        total = [None] * 4
        total[3] = gl[3]
        for l in (2, 1, 0):
            up = np.zeros_like(gl[l])
            h2, w2 = total[l + 1].shape[-2:]
            up[:, :h2 * 2, :w2 * 2] = np.repeat(np.repeat(total[l + 1], 2, axis=1), 2, axis=2) * np.float32(0.25)
            total[l] = gl[l] + up
        for l in range(4):
            assert np.allclose(total[l], g['glevel%d' % l].reshape(gl[l].shape), rtol=1e-4, atol=1e-4)


def _weightnet_params(g):
    return [g['p_weight_net__convs__%d__conv_fn__%s' % (i, part)].reshape(
        g['p_weight_net__convs__%d__conv_fn__weight' % i].shape[0], -1) for i in range(3) for part in ('weight', 'bias')]


@pytest.mark.parametrize('name', ['pointconv_dw_a', 'pointconv_dw_b'])
def test_weightnet_and_setconv_match_reference_module(name, golden, oracle_lib):
    """oracle_weightnet_fwd (+ oracle_pointconv_dw_fwd) against the reference's PointConvDW: the
    weight_net output captured by a forward hook, and the module output (fp32, different summation
    order than the 1x1 convolutions: 1e-5)."""
    g = golden(name)
    k = int(g['k'])
    params = _weightnet_params(g)
    weight, h2 = oracle_lib.weightnet_fwd(g['xyz'], g['xyz'], g['knn'], k, params, want_hidden=True)
    assert weight.shape == g['weight'].shape and h2.shape[1] == 32
    assert np.allclose(weight, g['weight'], rtol=1e-5, atol=1e-6)
    mlp_w = g['p_mlp__convs__0__conv_fn__weight'][:, :, 0]
    pre = np.einsum('oi,bim->bom', mlp_w, g['feat']) + g['p_mlp__convs__0__conv_fn__bias'][None, :, None]
    feat = np.where(pre > 0, pre, np.float32(0.1) * pre).astype(np.float32)            # leaky_relu(0.1)
    out, _ = oracle_lib.pointconv_dw_fwd(feat, weight, g['knn'], k)
    assert np.allclose(out, g['out'], rtol=1e-4, atol=1e-5)
    # backward: against float64 autograd on the same formulas (composed from the golden offsets)
    import torch
    gout = np.random.default_rng(0).standard_normal(weight.shape).astype(np.float32)
    grads = oracle_lib.weightnet_bwd(g['xyz'], g['xyz'], g['knn'], k, params, gout)
    leaves = [torch.tensor(p if i % 2 == 0 else p[:, 0], dtype=torch.float64, requires_grad=True)
              for i, p in enumerate(params)]
    x = torch.tensor(g['knn_offset'], dtype=torch.float64)
    for w, b in zip(leaves[0::2], leaves[1::2]):
        x = torch.relu(torch.einsum('oi,binj->bonj', w, x) + b.view(1, -1, 1, 1))
    x.backward(torch.tensor(gout, dtype=torch.float64))
    for got, leaf in zip(grads, leaves):
        ref = leaf.grad.numpy()
        assert np.linalg.norm(got - ref) <= 1e-5 * np.linalg.norm(ref) + 1e-9


def test_bilinear_sample_matches_reference_grid_sample_wrapper(golden, oracle_lib):
    g = golden('grid_sample')
    out = oracle_lib.bilinear_sample_fwd(g['feat'], g['uv'])
    assert out.shape == g['out'].shape
    assert np.allclose(out, g['out'], rtol=1e-5, atol=1e-6)


def _ids_consts(g):
    (ph, pw), (qh, qw) = g['persp_hw'], g['paral_hw']
    rw, rh = (qw - 1) / (pw - 1), (qh - 1) / (ph - 1)
    return float(rw), float(rh), float(min(rw, rh)), float((qw - 1) / 2), float((qh - 1) / 2)


def test_ids_flow_matches_reference_paral2persp(golden, oracle_lib):
    g = golden('ids_flow')
    intr = g['intrinsics']
    out = oracle_lib.ids_flow_fwd(g['pc1'], g['flow'], g['origin'], intr[:, 0], intr[:, 1], intr[:, 2], *_ids_consts(g))
    assert np.allclose(out, g['out'], rtol=1e-5, atol=1e-5)

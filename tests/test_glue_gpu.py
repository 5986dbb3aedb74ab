"""The fused "glue" kernels (camli_gru_gates/blend, camli_sk_*, camli_gather_scale, camli_masked_l2) against
oracle/glue.py -- the numpy restatement pinned on the reference's own GRU2D / SKFusion / batch_indexing / sequence-loss
code in tests/test_glue_oracle.py.  Values and every gradient, fp32; tolerances in the asserts."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import glue

pytestmark = pytest.mark.gpu


def dev(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda().requires_grad_(grad)


def host(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol=1e-5, atol=2e-6):
    return np.allclose(host(a) if torch.is_tensor(a) else a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('shape', [(2, 128, 68, 120), (3, 8, 5, 6), (1, 16, 2048)], ids=str)
def test_gru_gates_and_blend_vs_oracle(shape):
    """raft_core.py:124-138: z | r = sigmoid(pre + ctx), r * h; h' = (1 - z) h + z tanh(pre + ctx) [+ nan_to_num]."""
    from camliflow_amd.csrc import fused
    b, c = shape[:2]
    rng = np.random.default_rng(sum(shape))
    zr_shape = (b, 2 * c) + shape[2:]
    pre_zr, ctx_zr = rng.standard_normal(zr_shape).astype(np.float32), rng.standard_normal(zr_shape).astype(np.float32)
    h = rng.standard_normal(shape).astype(np.float32)
    gz, grh = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
    t_pre, t_ctx, t_h = dev(pre_zr, True), dev(ctx_zr, True), dev(h, True)
    z, rh = fused.gru_gates(t_pre, t_ctx, t_h)
    want_z, want_rh, want_r = glue.gru_gates_fwd(pre_zr, ctx_zr, h)
    assert close(z, want_z) and close(rh, want_rh)
    ((z * dev(gz)).sum() + (rh * dev(grh)).sum()).backward()
    gpre, gh = glue.gru_gates_bwd(gz, grh, want_z, want_r, h)
    assert close(t_pre.grad, gpre) and close(t_ctx.grad, gpre) and close(t_h.grad, gh)

    pre_q, ctx_q = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
    g = rng.standard_normal(shape).astype(np.float32)
    for sanitize in (False, True):
        t_pq, t_cq, t_z, t_h = dev(pre_q, True), dev(ctx_q, True), dev(want_z, True), dev(h, True)
        out = fused.gru_blend(t_pq, t_cq, t_z, t_h, nan_to_num=sanitize)
        want, q = glue.gru_blend_fwd(pre_q, ctx_q, want_z, h, nan_to_num=sanitize)
        assert close(out, want)
        out.backward(dev(g))
        gpq, gzz, ghh = glue.gru_blend_bwd(g, want_z, h, q)
        assert close(t_pq.grad, gpq) and close(t_cq.grad, gpq) and close(t_z.grad, gzz) and close(t_h.grad, ghh)


@pytest.mark.parametrize('shape', [(8, 128, 68, 120), (2, 64, 2048), (3, 5, 7, 9), (1, 16, 1)], ids=str)
def test_sk_pool_mix_vs_oracle(shape):
    """clfm.py:199, 203-213 and their adjoints (the pool's incoming gradient gs enters the same backward kernel)."""
    from camliflow_amd.csrc import fused
    rng = np.random.default_rng(sum(shape))
    a, b = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
    w = rng.random((shape[0], shape[1], 2)).astype(np.float32)
    g = rng.standard_normal(shape).astype(np.float32)
    gs = rng.standard_normal(shape[:2]).astype(np.float32)
    ta, tb, tw = dev(a, True), dev(b, True), dev(w, True)
    state = fused.SkState()
    s = fused.sk_pool(ta, tb, state)
    out = fused.sk_mix(ta, tb, tw, state)
    assert close(s, glue.sk_pool_fwd(a, b), atol=1e-6)
    assert close(out, glue.sk_mix_fwd(a, b, w))
    ((out * dev(g)).sum() + (s * dev(gs)).sum()).backward()
    ga, gb, gw = glue.sk_fuse_bwd(g, a, b, w, gs)
    assert close(ta.grad, ga) and close(tb.grad, gb)
    assert np.linalg.norm(host(tw.grad) - gw) <= 1e-5 * np.linalg.norm(gw) + 1e-6       # sums over up to 8160 positions


@pytest.mark.parametrize('dims', [(8, 128, 64), (3, 20, 7), (1, 627, 313), (2, 1024, 512)], ids=lambda d: 'B%d_C%d_R%d' % d)
def test_sk_gate_vs_oracle(dims):
    """clfm.py:184-191, 200-202: softmax(sigmoid(relu(s Wmid^T) Wout^T)) and its adjoint."""
    from camliflow_amd.csrc import fused
    b, c, r = dims
    rng = np.random.default_rng(sum(dims))
    s = rng.standard_normal((b, c)).astype(np.float32)
    wmid = (rng.standard_normal((r, c)) * c ** -0.5).astype(np.float32)
    wout = (rng.standard_normal((2 * c, r)) * r ** -0.5).astype(np.float32)
    gw = rng.standard_normal((b, c, 2)).astype(np.float32)
    ts, tm, to = dev(s, True), dev(wmid, True), dev(wout, True)
    w = fused.sk_gate(ts, tm, to)
    assert close(w, glue.sk_gate_fwd(s, wmid, wout), rtol=1e-4, atol=1e-6)
    w.backward(dev(gw))
    gs, gwmid, gwout = glue.sk_gate_bwd(gw, s, wmid, wout)
    for got, want in ((ts.grad, gs), (tm.grad, gwmid), (to.grad, gwout)):
        assert close(got, want, rtol=1e-4, atol=1e-6), np.abs(host(got) - want).max()


@pytest.mark.parametrize('case', [(8, 128, 2048, 8160), (2, 7, 50, 301)], ids=str)
def test_gather_scale_vs_oracle(case):
    """clfm.py:62-76 with k = 1: one exact product per element, forward and score adjoint."""
    from camliflow_amd.csrc import fused
    b, c, m, p = case
    rng = np.random.default_rng(sum(case))
    data = rng.standard_normal((b, c, m)).astype(np.float32)
    score = rng.random((b, c, p)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, p))
    gout = rng.standard_normal((b, c, p)).astype(np.float32)
    ts = dev(score, True)
    out = fused.gather_scale(dev(data), ts, torch.from_numpy(idx).cuda())
    out.backward(dev(gout))
    want, gathered = glue.gather_scale_fwd(data, score, idx)
    assert np.array_equal(host(out), want)
    assert np.array_equal(host(ts.grad), glue.gather_scale_bwd_score(gout, gathered))


@pytest.mark.parametrize('case', [(2, 2, (64, 81), True), (4, 2, (135, 240), True), (3, 3, (4097,), False), (2, 3, (2048,), True)],
                         ids=lambda c: 'B%d_C%d_%s_%s' % (c[0], c[1], 'x'.join(map(str, c[2])), 'mask' if c[3] else 'nomask'))
def test_sequence_loss_l2_vs_oracle(case):
    """objectives._sequence_loss (camli_masked_l2_fwd/bwd) vs models/losses.py:64-119 as restated by the oracle."""
    from camliflow_amd.cores import objectives, runtime
    b, c, sp, masked = case
    rng = np.random.default_rng(sum(sp) + b)
    target = rng.standard_normal((b, c + int(masked)) + sp).astype(np.float32)
    if masked:
        target[:, c] = (rng.random((b,) + sp) > 0.3).astype(np.float32)
    preds = [rng.standard_normal((b, c) + sp).astype(np.float32) for _ in range(4)]
    preds[1][0, :, ..., :3] = target[0, :c, ..., :3]          # exact hits: zero error, zero gradient
    cfgs = SimpleNamespace(gamma=0.8, order='l2-norm')
    tp = [dev(q, True) for q in preds]
    with runtime.use_backend('hip'):
        loss = objectives._sequence_loss(tp, dev(target), cfgs, c)
    loss.backward()
    want = glue.sequence_loss_l2_fwd(preds, target, c, cfgs.gamma)
    assert abs(loss.item() - want) <= 1e-5 * abs(want)
    for got, g in zip(tp, glue.sequence_loss_l2_bwd(preds, target, c, cfgs.gamma)):
        assert np.allclose(host(got.grad), g, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('shape', [(2, 128, 68, 120), (3, 8, 5, 8), (1, 16, 2048)], ids=str)
def test_gates_and_bias_adjoints_read_channel_slices_in_place(shape):
    """The adjoint of cat([r*h, x]) hands the gates kernel a channel SLICE of a wider gradient, the adjoint of
    cat([c, f]) hands one to the bias / activation kernel: camli_gru_gates_bwd_strided / camli_bias_act_bwd_strided read it
    where it lies.  Same values as the oracle, and as the dense call (bit for bit)."""
    from camliflow_amd.csrc import fused
    b, c = shape[:2]
    rng = np.random.default_rng(sum(shape) + 1)
    zr_shape = (b, 2 * c) + shape[2:]
    wide = (b, 2 * c + 4) + shape[2:]
    pre_zr, ctx_zr = rng.standard_normal(zr_shape).astype(np.float32), rng.standard_normal(zr_shape).astype(np.float32)
    h = rng.standard_normal(shape).astype(np.float32)
    gz = rng.standard_normal(shape).astype(np.float32)
    gwide = rng.standard_normal(wide).astype(np.float32)            # gradient of cat([rh, extra]) -> grh = gwide[:, :c]
    extra = dev(rng.standard_normal((b, c + 4) + shape[2:]).astype(np.float32), True)
    t_pre, t_ctx, t_h = dev(pre_zr, True), dev(ctx_zr, True), dev(h, True)
    z, rh = fused.gru_gates(t_pre, t_ctx, t_h)
    ((z * dev(gz)).sum() + (torch.cat([rh, extra], dim=1) * dev(gwide)).sum()).backward()
    want_z, _, want_r = glue.gru_gates_fwd(pre_zr, ctx_zr, h)
    gpre, gh = glue.gru_gates_bwd(gz, gwide[:, :c], want_z, want_r, h)
    assert close(t_pre.grad, gpre) and close(t_h.grad, gh)
    d_pre, d_h = dev(pre_zr, True), dev(h, True)
    z2, rh2 = fused.gru_gates(d_pre, dev(ctx_zr), d_h)
    ((z2 * dev(gz)).sum() + (rh2 * dev(np.ascontiguousarray(gwide[:, :c]))).sum()).backward()
    assert torch.equal(d_pre.grad, t_pre.grad) and torch.equal(d_h.grad, t_h.grad)

    # bias + activation: y = act(x + bias) sits in the middle of a cat
    for act, fn in (('relu', lambda v: np.maximum(v, 0)), ('leaky_relu', lambda v: np.where(v > 0, v, 0.1 * v)), (None, lambda v: v)):
        x = rng.standard_normal(shape).astype(np.float32)
        bias = rng.standard_normal(c).astype(np.float32)
        left = dev(rng.standard_normal((b, 4) + shape[2:]).astype(np.float32), True)
        t_x, t_b = dev(x, True), dev(bias, True)
        y = fused.bias_act(t_x * 1.0, t_b, act)
        gw = rng.standard_normal((b, c + 8) + shape[2:]).astype(np.float32)
        (torch.cat([left, y, left], dim=1) * dev(gw)).sum().backward()
        pre = x.astype(np.float64) + bias.reshape((1, c) + (1,) * (len(shape) - 2))
        slope = np.ones_like(pre) if act is None else np.where(pre > 0, 1.0, 0.0 if act == 'relu' else 0.1)
        want_gx = gw[:, 4:4 + c] * slope
        assert close(y, fn(pre)) and close(t_x.grad, want_gx)
        assert close(t_b.grad, want_gx.sum(axis=(0,) + tuple(range(2, len(shape)))), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('shape', [(2, 68, 120), (3, 5, 8), (1, 2048)], ids=str)
def test_bias_act_cat_writes_the_concatenation_directly(shape):
    """fused.bias_act_cat == cat([act_i(x_i + b_i)..., tail]) (camli_bias_act_into_fwd + camli_bias_act_bwd_strided): values,
    every gradient, the inputs left untouched; relu_nan_to_num on a part that holds NaN / inf."""
    from camliflow_amd.csrc import fused
    b, spatial = shape[0], shape[1:]
    rng = np.random.default_rng(sum(shape))
    specs = [(7, 'relu'), (12, 'leaky_relu'), (3, None), (5, 'relu_nan_to_num')]
    xs = [rng.standard_normal((b, c) + spatial).astype(np.float32) for c, _ in specs]
    xs[3].reshape(-1)[::17] = np.inf
    xs[3].reshape(-1)[5::23] = np.nan
    xs[3].reshape(-1)[7::29] = -np.inf
    bs = [rng.standard_normal(c).astype(np.float32) for c, _ in specs]
    tail = rng.standard_normal((b, 2) + spatial).astype(np.float32)
    txs, tbs, ttail = [dev(x, True) for x in xs], [dev(v, True) for v in bs], dev(tail, True)
    keep = [t.detach().clone() for t in txs]
    out = fused.bias_act_cat([(t * 1.0, bb, act) for t, bb, (_, act) in zip(txs, tbs, specs)], tail=ttail)
    total = sum(c for c, _ in specs) + 2
    assert out.shape == (b, total) + spatial and out.is_contiguous()
    gout = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(gout))
    c0 = 0
    for x, bias, (c, act), tx, tb, k0 in zip(xs, bs, specs, txs, tbs, keep):
        assert torch.equal(tx.detach().view(torch.int32), k0.view(torch.int32))      # bit patterns: the NaNs compare too
        pre = x.astype(np.float64) + bias.reshape((1, c) + (1,) * len(spatial))
        g = gout[:, c0:c0 + c].astype(np.float64)
        with np.errstate(invalid='ignore'):
            if act == 'relu':
                want, slope = np.maximum(pre, 0), (pre > 0).astype(np.float64)
            elif act == 'leaky_relu':
                want, slope = np.where(pre > 0, pre, 0.1 * pre), np.where(pre > 0, 1.0, 0.1)
            elif act is None:
                want, slope = pre, np.ones_like(pre)
            else:
                want = np.nan_to_num(np.maximum(pre, 0).astype(np.float32), nan=0.0).astype(np.float64)
                slope = ((pre > 0) & np.isfinite(pre)).astype(np.float64)
        assert close(out[:, c0:c0 + c], want.astype(np.float32), rtol=1e-6, atol=1e-6), act
        assert close(tx.grad, (g * slope).astype(np.float32)), act
        assert close(tb.grad, (g * slope).sum(axis=(0,) + tuple(range(2, 2 + len(spatial)))).astype(np.float32), rtol=1e-4, atol=1e-3), act
        c0 += c
    assert torch.equal(out[:, c0:].detach(), dev(tail)) and close(ttail.grad, gout[:, c0:])

"""CPU restatement (TEST INFRASTRUCTURE, numpy) of the element-wise / reduction pieces of the reference that the
product path runs as fused "glue" kernels: the GRU2D gate arithmetic, the selective-kernel fusion, the score product of
the fusion-aware interpolation and the l2-norm sequence loss.  Forward AND adjoint of each, written out by hand (the
reference gets its adjoints from autograd).  Arithmetic in float64, results cast to float32: the kernels are compared
within the fp32 tolerances stated in the tests.

Pinned by tests/golden/glue_*.npz (tests/golden/make_glue_golden.py runs the reference's own GRU2D / SKFusion /
FusionAwareInterp / calc_sequence_loss_{2d,3d} with autograd and records inputs, intermediates, outputs, gradients);
tests/test_glue_oracle.py checks every function below against them on the CPU.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np


def _f64(*arrays):
    return [np.asarray(a, dtype=np.float64) for a in arrays]


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ---- GRU2D half-step, models/raft_core.py:124-130 (horizontal) and 132-136 (vertical) ---------------------------------
# hx = cat([h, x]); z = sigmoid(convz(hx)); r = sigmoid(convr(hx)); q = tanh(convq(cat([r*h, x]))); h' = (1-z)*h + z*q
# The product path splits each convolution into its h part (`pre`) and its x part (`ctx`, computed once per pass:
# x's context half does not change over the iterations), so the element-wise pieces take pre + ctx.

def gru_gates_fwd(pre_zr, ctx_zr, h):
    """pre_zr, ctx_zr [B,2C,...] (z rows first, then r: convz | convr stacked), h [B,C,...] -> z, r*h, r."""
    pre, ctx, h = _f64(pre_zr, ctx_zr, h)
    c = h.shape[1]
    s = _sigmoid(pre + ctx)
    z, r = s[:, :c], s[:, c:]
    return z.astype(np.float32), (r * h).astype(np.float32), r.astype(np.float32)


def gru_gates_bwd(gz, grh, z, r, h):
    """adjoint of gru_gates_fwd: (d/d pre_zr = d/d ctx_zr [B,2C,...], d/d h)."""
    gz, grh, z, r, h = _f64(gz, grh, z, r, h)
    gr = grh * h
    gpre = np.concatenate([gz * z * (1.0 - z), gr * r * (1.0 - r)], axis=1)
    return gpre.astype(np.float32), (grh * r).astype(np.float32)


def gru_blend_fwd(pre_q, ctx_q, z, h, nan_to_num=False):
    """h' = (1 - z) * h + z * tanh(pre_q + ctx_q); nan_to_num: followed by torch.nan_to_num (raft_core.py:138)."""
    pre, ctx, z, h = _f64(pre_q, ctx_q, z, h)
    q = np.tanh(pre + ctx)
    out = ((1.0 - z) * h + z * q).astype(np.float32)
    if nan_to_num:
        out = np.nan_to_num(out, nan=0.0, posinf=np.finfo(np.float32).max, neginf=np.finfo(np.float32).min)
    return out, q.astype(np.float32)


def gru_blend_bwd(g, z, h, q):
    """adjoint of gru_blend_fwd (finite values): d/d pre_q (= d/d ctx_q), d/d z, d/d h."""
    g, z, h, q = _f64(g, z, h, q)
    return (g * z * (1.0 - q * q)).astype(np.float32), (g * (q - h)).astype(np.float32), (g * (1.0 - z)).astype(np.float32)


# ---- SKFusion, models/clfm.py:193-213 --------------------------------------------------------------------------------
# weight = avg_pool(feat_2d + feat_3d); weight = fc_out(fc_mid(weight)).reshape(bs, C, 2); softmax(-1);
# return feat_2d * w1 + feat_3d * w2          (fc_mid = Linear(no bias) + ReLU, fc_out = Linear(no bias) + Sigmoid: 184-191)

def sk_pool_fwd(a, b):
    """[B,C,...] x 2 -> [B,C]: mean over the positions of a + b (clfm.py:199)."""
    a, b = _f64(a, b)
    return (a + b).reshape(a.shape[0], a.shape[1], -1).mean(-1).astype(np.float32)


def sk_gate_fwd(s, wmid, wout):
    """s [B,C], wmid [R,C], wout [2C,R] -> softmax(sigmoid(relu(s wmid^T) wout^T).reshape(B,C,2), -1)  (clfm.py:200-202)."""
    s, wmid, wout = _f64(s, wmid, wout)
    mid = np.maximum(s @ wmid.T, 0.0)
    sg = _sigmoid(mid @ wout.T).reshape(s.shape[0], -1, 2)
    e = np.exp(sg - sg.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def sk_gate_bwd(gw, s, wmid, wout):
    """adjoint of sk_gate_fwd: d/d s, d/d wmid, d/d wout."""
    gw, s, wmid, wout = _f64(gw, s, wmid, wout)
    b = s.shape[0]
    pre_mid = s @ wmid.T
    mid = np.maximum(pre_mid, 0.0)
    sg = _sigmoid(mid @ wout.T).reshape(b, -1, 2)
    e = np.exp(sg - sg.max(-1, keepdims=True))
    w = e / e.sum(-1, keepdims=True)
    gsg = w * (gw - (gw * w).sum(-1, keepdims=True))            # softmax adjoint
    gout = (gsg * sg * (1.0 - sg)).reshape(b, -1)                # sigmoid adjoint, [B,2C]
    gwout = gout.T @ mid
    gmid = (gout @ wout) * (pre_mid > 0)
    return (gmid @ wmid).astype(np.float32), (gmid.T @ s).astype(np.float32), gwout.astype(np.float32)


def sk_mix_fwd(a, b, w):
    """a, b [B,C,...], w [B,C,2] -> a * w[...,0] + b * w[...,1]  (clfm.py:203-213)."""
    a, b, w = _f64(a, b, w)
    bshape = (a.shape[0], a.shape[1]) + (1,) * (a.ndim - 2)
    return (a * w[..., 0].reshape(bshape) + b * w[..., 1].reshape(bshape)).astype(np.float32)


def sk_fuse_bwd(g, a, b, w, gs):
    """adjoint of out = sk_mix(a, b, w) together with s = sk_pool(a, b) whose incoming gradient is gs [B,C]:
    d/d a, d/d b, d/d w [B,C,2]."""
    g, a, b, w, gs = _f64(g, a, b, w, gs)
    bshape = (a.shape[0], a.shape[1]) + (1,) * (a.ndim - 2)
    p = a[0, 0].size
    pool = gs.reshape(bshape) / p
    ga = g * w[..., 0].reshape(bshape) + pool
    gb = g * w[..., 1].reshape(bshape) + pool
    red = tuple(range(2, a.ndim))
    gw = np.stack([(g * a).sum(red), (g * b).sum(red)], axis=-1)
    return ga.astype(np.float32), gb.astype(np.float32), gw.astype(np.float32)


# ---- score product of FusionAwareInterp, models/clfm.py:69-76 (k = 1: one neighbour per pixel) ------------------------
# knn_feat3d = batch_indexing(feat_3d, knn_indices); final = score * knn_feat3d; final.sum(dim=-1)

def gather_scale_fwd(data, score, idx):
    """data [B,C,M], score [B,C,P], idx [B,P] -> score * data[:, :, idx]; also returns the gathered rows."""
    data, score = np.asarray(data, np.float32), np.asarray(score, np.float32)
    gathered = np.take_along_axis(data, np.asarray(idx)[:, None, :].repeat(data.shape[1], 1), axis=2)
    return score * gathered, gathered


def gather_scale_bwd_score(gout, gathered):
    return np.asarray(gout, np.float32) * gathered


# ---- l2-norm sequence loss, models/losses.py:64-119 ------------------------------------------------------------------
# mask = target[:, C] > 0 (or all ones); loss_i = ||pred_i - target[:, :C]||_2 [mask].mean(); total = sum gamma^(n-i-1) loss_i

def sequence_loss_l2_fwd(preds, target, n_channels, gamma):
    target = np.asarray(target, np.float64)
    mask = target[:, n_channels] > 0 if target.shape[1] == n_channels + 1 else np.ones(target[:, 0].shape, bool)
    total = 0.0
    n = len(preds)
    for i, p in enumerate(preds):
        diff = np.asarray(p, np.float64) - target[:, :n_channels]
        total += gamma ** (n - i - 1) * np.sqrt((diff * diff).sum(1))[mask].mean()
    return np.float32(total)


def sequence_loss_l2_bwd(preds, target, n_channels, gamma):
    """d total / d pred_i for every iterate (zero where the error is exactly zero, as torch.linalg.norm's adjoint)."""
    target = np.asarray(target, np.float64)
    mask = target[:, n_channels] > 0 if target.shape[1] == n_channels + 1 else np.ones(target[:, 0].shape, bool)
    count = mask.sum()
    n = len(preds)
    grads = []
    for i, p in enumerate(preds):
        diff = np.asarray(p, np.float64) - target[:, :n_channels]
        norm = np.sqrt((diff * diff).sum(1, keepdims=True))
        g = np.where(norm > 0, diff / np.where(norm > 0, norm, 1.0), 0.0) * mask[:, None]
        grads.append((gamma ** (n - i - 1) / count * g).astype(np.float32))
    return grads

"""CPU restatement (TEST INFRASTRUCTURE, numpy) of the dense pieces of the reference that the product path runs on its
own round-3 kernels: the cost MLP + neighbour sum of the point cost-volume lookup, the two-channel 3x3 convolution heads,
the all-pairs volume build, the ResNet stem's max pooling and the bias / shortcut / ReLU epilogues.  Forward AND adjoint
of each, written out by hand (the reference gets its adjoints from autograd).  Arithmetic in float64, results cast to
float32: the kernels are compared within the fp32 tolerances stated in the tests.

Pinned by tests/golden/dense_*.npz (tests/golden/make_dense_golden.py runs the reference's own Correlation3D.cost_mlp,
FlowHead2D and Correlation2D.build_cost_volume_pyramid with autograd and records inputs, outputs, gradients; the
pooling / epilogue pieces belong to mmdet's ResNet, which is not under /root/reference -- SURVEY 8c -- and are recorded
from the torch modules the reference's call site instantiates it with); tests/test_dense_oracle.py checks every function
below against them on the CPU.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np


def _f64(*arrays):
    return [np.asarray(a, dtype=np.float64) for a in arrays]


# ---- cost MLP + neighbour sum, models/camliraft_l_core.py:68-101 ------------------------------------------------------
# calc_matching_cost: cost = MLP2d(4 -> H -> H, relu)(cat[knn_xyz2 - xyz1, knn_corr]) summed over the k neighbours (:96-98);
# forward: the four levels' costs concatenated along the channels (:93), level-major.  The product path hands all levels
# to one kernel as lookup [B,4,N,L*k] (column l*k + j = neighbour j of level l).

def cost_mlp_fwd(lookup, w1, b1, w2, b2, levels):
    """lookup [B,4,N,L*k], w1 [H,4], b1 [H], w2 [H,H], b2 [H] -> [B, L*H, N] (channel l*H + o)."""
    x, w1, b1, w2, b2 = _f64(lookup, w1, b1, w2, b2)
    b, _, n, lk = x.shape
    k = lk // levels
    h1 = np.maximum(np.einsum('oc,bcnj->bonj', w1, x) + b1[None, :, None, None], 0.0)
    h2 = np.maximum(np.einsum('oi,binj->bonj', w2, h1) + b2[None, :, None, None], 0.0)
    out = h2.reshape(b, -1, n, levels, k).sum(axis=-1)            # [B,H,N,L]
    return out.transpose(0, 3, 1, 2).reshape(b, -1, n).astype(np.float32)


def cost_mlp_bwd(gout, lookup, w1, b1, w2, b2, levels):
    """adjoint of cost_mlp_fwd: (d/d lookup [B,4,N,L*k], d/d w1, d/d b1, d/d w2, d/d b2).  The product path only
    propagates channel 3 of d/d lookup (the cost-volume entry); the reference detaches nothing here, autograd returns all
    four, so all four are restated."""
    g, x, w1, b1, w2, b2 = _f64(gout, lookup, w1, b1, w2, b2)
    b, _, n, lk = x.shape
    k = lk // levels
    hdim = w2.shape[0]
    pre1 = np.einsum('oc,bcnj->bonj', w1, x) + b1[None, :, None, None]
    h1 = np.maximum(pre1, 0.0)
    pre2 = np.einsum('oi,binj->bonj', w2, h1) + b2[None, :, None, None]
    g2 = np.repeat(g.reshape(b, levels, hdim, n).transpose(0, 2, 3, 1)[..., None], k, axis=-1).reshape(b, hdim, n, lk)
    g2 = g2 * (pre2 > 0.0)
    gw2 = np.einsum('bonj,binj->oi', g2, h1)
    gb2 = g2.sum(axis=(0, 2, 3))
    g1 = np.einsum('oi,bonj->binj', w2, g2) * (pre1 > 0.0)
    gw1 = np.einsum('bonj,bcnj->oc', g1, x)
    gb1 = g1.sum(axis=(0, 2, 3))
    gx = np.einsum('oc,bonj->bcnj', w1, g1)
    return tuple(a.astype(np.float32) for a in (gx, gw1, gb1, gw2, gb2))


# ---- 3x3 convolution, zero padding 1: FlowHead2D.conv2, models/raft_core.py:169-182 (and PWC's conv_last) -------------

def conv3x3_fwd(x, w, bias=None):
    """x [B,Ci,H,W], w [Co,Ci,3,3], bias [Co] or None -> [B,Co,H,W] (cross-correlation, as nn.Conv2d)."""
    x, w = _f64(x, w)
    b, ci, hh, ww = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((b, w.shape[0], hh, ww))
    for dy in range(3):
        for dx in range(3):
            out += np.einsum('oc,bchw->bohw', w[:, :, dy, dx], xp[:, :, dy:dy + hh, dx:dx + ww])
    if bias is not None:
        out += np.asarray(bias, dtype=np.float64)[None, :, None, None]
    return out.astype(np.float32)


def conv3x3_bwd(gy, x, w):
    """adjoint of conv3x3_fwd: (d/d x, d/d w, d/d bias)."""
    gy, x, w = _f64(gy, x, w)
    b, ci, hh, ww = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    for dy in range(3):
        for dx in range(3):
            gw[:, :, dy, dx] = np.einsum('bohw,bchw->oc', gy, xp[:, :, dy:dy + hh, dx:dx + ww])
            gxp[:, :, dy:dy + hh, dx:dx + ww] += np.einsum('oc,bohw->bchw', w[:, :, dy, dx], gy)
    return gxp[:, :, 1:-1, 1:-1].astype(np.float32), gw.astype(np.float32), gy.sum(axis=(0, 2, 3)).astype(np.float32)


# ---- 1x5 / 5x1 convolution, zero padding 2 along the kernel: GRU2D's gates, models/raft_core.py:110-122 -----------------

def conv5_fwd(x, w, bias=None):
    """x [B,Ci,H,W], w [Co,Ci,1,5] (horizontal) or [Co,Ci,5,1] (vertical), bias [Co] or None -> [B,Co,H,W] (as nn.Conv2d with
    padding (0,2) / (2,0): cross-correlation)."""
    x, w = _f64(x, w)
    vertical = w.shape[2] == 5
    taps = w.reshape(w.shape[0], w.shape[1], 5)
    b, ci, hh, ww = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (2, 2), (0, 0)) if vertical else ((0, 0), (0, 0), (0, 0), (2, 2)))
    out = np.zeros((b, w.shape[0], hh, ww))
    for t in range(5):
        win = xp[:, :, t:t + hh, :] if vertical else xp[:, :, :, t:t + ww]
        out += np.einsum('oc,bchw->bohw', taps[:, :, t], win)
    if bias is not None:
        out += np.asarray(bias, dtype=np.float64)[None, :, None, None]
    return out.astype(np.float32)


# ---- any stride-1 convolution with zero padding, as nn.Conv2d computes it (cross-correlation) -- models/raft_core.py:110-197 ----
# (GRU2D's 1x5 / 5x1, MotionEncoder2D's 1x1 / 3x3 / 7x7, the flow and mask heads' 3x3): forward and both adjoints, the
# oracle of camli_convcl_fwd / camli_convcl_wrw.  Pinned on torch's own conv2d in fp64 (tests/test_dense_oracle.py).

def conv_taps_fwd(x, w, padding):
    """x [B,Ci,H,W], w [Co,Ci,kh,kw], padding (ph, pw), stride 1 -> [B,Co,H+2ph-kh+1,W+2pw-kw+1]."""
    x, w = _f64(x, w)
    ph, pw = padding
    kh, kw = w.shape[2:]
    b, ci, hh, ww = x.shape
    ho, wo = hh + 2 * ph - kh + 1, ww + 2 * pw - kw + 1
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    out = np.zeros((b, w.shape[0], ho, wo))
    for ky in range(kh):
        for kx in range(kw):
            out += np.einsum('oc,bchw->bohw', w[:, :, ky, kx], xp[:, :, ky:ky + ho, kx:kx + wo])
    return out.astype(np.float32)


def conv_taps_bwd(gy, x, w, padding):
    """adjoint of conv_taps_fwd: (d/d x, d/d w)."""
    gy, x, w = _f64(gy, x, w)
    ph, pw = padding
    kh, kw = w.shape[2:]
    ho, wo = gy.shape[2:]
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    for ky in range(kh):
        for kx in range(kw):
            gw[:, :, ky, kx] = np.einsum('bohw,bchw->oc', gy, xp[:, :, ky:ky + ho, kx:kx + wo])
            gxp[:, :, ky:ky + ho, kx:kx + wo] += np.einsum('oc,bohw->bchw', w[:, :, ky, kx], gy)
    hh, ww = x.shape[2:]
    return gxp[:, :, ph:ph + hh, pw:pw + ww].astype(np.float32), gw.astype(np.float32)


# ---- all-pairs volume pyramid, models/raft_core.py:52-68 (after fnet_aligner) -----------------------------------------
# cost_volume = f1^T f2 / sqrt(C) as [B*P, 1, h, w]; then avg_pool2d(2, stride 2) over the TARGET dims, num_levels - 1
# times (floor: an odd trailing row / column is dropped).

def _avg_pool2(v):
    hh, ww = v.shape[-2] // 2, v.shape[-1] // 2
    v = v[..., :2 * hh, :2 * ww]
    return 0.25 * (v[..., 0::2, 0::2] + v[..., 0::2, 1::2] + v[..., 1::2, 0::2] + v[..., 1::2, 1::2])


def allpairs_pyramid_fwd(f1, f2, num_levels=4):
    """f1, f2 [B,C,h,w] -> list of num_levels arrays [B*h*w, h_l, w_l]."""
    f1, f2 = _f64(f1, f2)
    b, c, hh, ww = f1.shape
    vol = np.einsum('bcp,bcq->bpq', f1.reshape(b, c, -1), f2.reshape(b, c, -1)) / np.sqrt(np.float64(np.float32(c)))
    vol = vol.reshape(b * hh * ww, hh, ww)
    pyr = [vol]
    for _ in range(num_levels - 1):
        pyr.append(_avg_pool2(pyr[-1]))
    return [p.astype(np.float32) for p in pyr]


def allpairs_pyramid_bwd(gpyr, f1, f2):
    """adjoint: gpyr = list of gradients of the pyramid levels -> (d/d f1, d/d f2).  The pooling adjoint spreads a
    quarter of a pooled gradient over its four sources (nothing reaches a dropped odd row / column)."""
    f1, f2 = _f64(f1, f2)
    b, c, hh, ww = f1.shape
    g = None
    for lvl in reversed(range(len(gpyr))):
        cur = np.asarray(gpyr[lvl], dtype=np.float64).copy()
        if g is not None:
            up = np.zeros_like(cur)
            ph, pw = g.shape[-2], g.shape[-1]
            for dy in range(2):
                for dx in range(2):
                    up[..., dy:2 * ph:2, dx:2 * pw:2] += 0.25 * g
            cur += up
        g = cur
    gvol = g.reshape(b, hh * ww, hh * ww) / np.sqrt(np.float64(np.float32(c)))
    gf1 = np.einsum('bpq,bcq->bcp', gvol, f2.reshape(b, c, -1)).reshape(f1.shape)
    gf2 = np.einsum('bpq,bcp->bcq', gvol, f1.reshape(b, c, -1)).reshape(f2.shape)
    return gf1.astype(np.float32), gf2.astype(np.float32)


# ---- point cost-volume pyramid, models/camliraft_l_core.py:51-60 -------------------------------------------------------
# cost_volume = bmm(f1^T, f2) / C  [B,N,M0]; level i = mean over the k nearest level-(i-1) targets of every level-i target
# of the level-(i-1) volume's columns (batch_indexing(volume, knn) -> [B,N,M_i,k], mean over k).  The neighbour tables
# are inputs here (k_nearest_neighbor has its own oracle).

def point_volume_pyramid_fwd(f1, f2, parents):
    """f1 [B,C,N], f2 [B,C,M0], parents[i] int [B,M_{i+1},k] -> list of [B,N,M_i]."""
    f1, f2 = _f64(f1, f2)
    b, c, _ = f1.shape
    pyr = [np.einsum('bcn,bcm->bnm', f1, f2) / np.float64(c)]
    for idx in parents:
        idx = np.asarray(idx, dtype=np.int64)
        prev = pyr[-1]
        cols = np.stack([prev[bi][:, idx[bi]] for bi in range(b)])          # [B,N,M_i,k]
        pyr.append(cols.mean(axis=-1))
    return [p.astype(np.float32) for p in pyr]


def point_volume_pyramid_bwd(gpyr, f1, f2, parents):
    """adjoint of the above: gradients of the levels -> (d/d f1, d/d f2).  The mean's adjoint adds g / k into each of
    the k gathered columns (repeated parents add repeatedly)."""
    f1, f2 = _f64(f1, f2)
    b, c, _ = f1.shape
    g = None
    for lvl in reversed(range(len(gpyr))):
        cur = np.asarray(gpyr[lvl], dtype=np.float64).copy()
        if g is not None:
            idx = np.asarray(parents[lvl], dtype=np.int64)                   # [B,M_{lvl+1},k] into level lvl
            k = idx.shape[-1]
            for bi in range(b):
                for j in range(k):
                    np.add.at(cur[bi], (slice(None), idx[bi, :, j]), g[bi] / k)
        g = cur
    gvol = g / np.float64(c)
    gf1 = np.einsum('bnm,bcm->bcn', gvol, f2)
    gf2 = np.einsum('bnm,bcn->bcm', gvol, f1)
    return gf1.astype(np.float32), gf2.astype(np.float32)


# ---- ResNet stem max pooling (kernel 3, stride 2, padding 1) and the bottleneck epilogue -------------------------------
# mmdet 2.14 ResNet (README.md:78-79; call site models/raft_core.py:10-38): self.maxpool = nn.MaxPool2d(3, 2, 1);
# Bottleneck.forward ends with  out = bn3(conv3(.)) ; out += identity ; out = relu(out)  -- with the frozen / folded
# BatchNorm of the product path bn3 is a per-channel bias on the folded convolution.

def maxpool3x3s2_fwd(x):
    """x [B,C,H,W] -> (y [B,C,Ho,Wo], flat argmax index into H*W of the FIRST maximum in row-major window order)."""
    x = np.asarray(x, dtype=np.float32)
    b, c, hh, ww = x.shape
    ho, wo = (hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1
    y = np.full((b, c, ho, wo), -np.inf, dtype=np.float32)
    arg = np.zeros((b, c, ho, wo), dtype=np.int64)
    oy, ox = np.arange(ho)[:, None], np.arange(wo)[None, :]
    for dy in range(3):
        for dx in range(3):
            iy, ix = 2 * oy - 1 + dy, 2 * ox - 1 + dx
            ok = (iy >= 0) & (iy < hh) & (ix >= 0) & (ix < ww)
            v = np.where(ok[None, None], x[:, :, np.clip(iy, 0, hh - 1), np.clip(ix, 0, ww - 1)], -np.inf)
            take = v > y
            y = np.where(take, v, y)
            arg = np.where(take, (iy * ww + ix)[None, None], arg)
    return y, arg


def maxpool3x3s2_bwd(gy, arg, in_hw):
    """adjoint: every pooled gradient goes to the input position that attained the maximum."""
    gy = np.asarray(gy, dtype=np.float64)
    b, c = gy.shape[:2]
    gx = np.zeros((b, c, in_hw[0] * in_hw[1]))
    bi, ci = np.meshgrid(np.arange(b), np.arange(c), indexing='ij')
    np.add.at(gx, (bi[..., None, None], ci[..., None, None], arg), gy)
    return gx.reshape(b, c, *in_hw).astype(np.float32)


def bias_act_res_fwd(x, bias, res=None, relu=True):
    """y = [relu](x + bias[c] [+ res]) on NCHW arrays (the channels-last kernels compute the same numbers)."""
    x, bias = _f64(x, bias)
    y = x + bias[None, :, None, None]
    if res is not None:
        y = y + np.asarray(res, dtype=np.float64)
    if relu:
        y = np.maximum(y, 0.0)
    return y.astype(np.float32)


def bias_act_res_bwd(gy, y, relu=True):
    """adjoint: (d/d x = d/d res, d/d bias)."""
    gy, y = _f64(gy, y)
    gx = gy * (y > 0.0) if relu else gy
    return gx.astype(np.float32), gx.sum(axis=(0, 2, 3)).astype(np.float32)

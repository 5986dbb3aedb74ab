"""numpy <-> liboracle.so (ctypes).  All arrays are C-contiguous float32 / int64."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "camli_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(LIB_PATH):
        subprocess.run(["make", "-C", HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed: %d" % (name, rc))


def knn(inp, query, k):
    """inp [B,M,D], query [B,Nq,D] -> int64 [B,Nq,k]"""
    inp, query = _f32(inp), _f32(query)
    B, M, D = inp.shape
    Nq = query.shape[1]
    out = np.zeros((B, Nq, k), dtype=np.int64)
    _chk(_load().oracle_knn(_p(inp), _p(query), _p(out), B, M, Nq, D, k), "knn")
    return out


def fps(xyz, n_samples):
    """xyz [B,N,3] -> int64 [B,n_samples]"""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, n_samples), dtype=np.int64)
    _chk(_load().oracle_fps(_p(xyz), _p(out), B, N, n_samples), "fps")
    return out


def corr2d_fwd(in1_nhwc, in2_nhwc, md):
    in1, in2 = _f32(in1_nhwc), _f32(in2_nhwc)
    B, H, W, C = in1.shape
    d = 2 * md + 1
    out = np.zeros((B, d * d, H, W), dtype=np.float32)
    _chk(_load().oracle_corr2d_fwd(_p(in1), _p(in2), _p(out), B, C, H, W, md), "corr2d_fwd")
    return out


def corr2d_bwd(gout_nchw, in1_nhwc, in2_nhwc, md):
    g, in1, in2 = _f32(gout_nchw), _f32(in1_nhwc), _f32(in2_nhwc)
    B, H, W, C = in1.shape
    g1, g2 = np.zeros_like(in1), np.zeros_like(in2)
    _chk(_load().oracle_corr2d_bwd(_p(g), _p(in1), _p(in2), _p(g1), _p(g2), B, C, H, W, md), "corr2d_bwd")
    return g1, g2


def _level_args(vols):
    L = len(vols)
    ptrs = (ctypes.c_void_p * L)(*[v.ctypes.data for v in vols])
    hs = (ctypes.c_int * L)(*[v.shape[-2] for v in vols])
    ws = (ctypes.c_int * L)(*[v.shape[-1] for v in vols])
    return L, ptrs, hs, ws


def allpairs_lookup_fwd(vols, coords, radius):
    """vols: list of [B*P, h_l, w_l]; coords [B,2,h,w] -> [B, L*(2r+1)^2, h, w]"""
    vols = [_f32(v) for v in vols]
    coords = _f32(coords)
    B, _, h, w = coords.shape
    L, ptrs, hs, ws = _level_args(vols)
    d = 2 * radius + 1
    out = np.zeros((B, L * d * d, h, w), dtype=np.float32)
    _chk(_load().oracle_allpairs_lookup_fwd(ptrs, hs, ws, L, _p(coords), _p(out), B, h, w, radius), "lookup_fwd")
    return out


def allpairs_lookup_bwd(vol_shapes, coords, gout, radius):
    """returns list of grads shaped like vol_shapes"""
    coords, gout = _f32(coords), _f32(gout)
    B, _, h, w = coords.shape
    gvols = [np.zeros(s, dtype=np.float32) for s in vol_shapes]
    L, ptrs, hs, ws = _level_args(gvols)
    _chk(_load().oracle_allpairs_lookup_bwd(ptrs, hs, ws, L, _p(coords), _p(gout), B, h, w, radius), "lookup_bwd")
    return gvols


def gather_cf(data, idx):
    """data [B,C,N], idx int64 [B,I] -> [B,C,I]"""
    data, idx = _f32(data), _i64(idx)
    B, C, N = data.shape
    I = idx.shape[1]
    out = np.zeros((B, C, I), dtype=np.float32)
    _chk(_load().oracle_gather_cf(_p(data), _p(idx), _p(out), B, C, N, I), "gather_cf")
    return out


def scatter_add_cf(gout, idx, N):
    gout, idx = _f32(gout), _i64(idx)
    B, C, I = gout.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    _chk(_load().oracle_scatter_add_cf(_p(gout), _p(idx), _p(out), B, C, N, I), "scatter_add_cf")
    return out


def gather_cl(data, idx):
    """channel-last batch_indexing: data [B,N,C] (or [B,N]), idx int64 [B,I] -> [B,I,C] (or [B,I])"""
    data, idx = _f32(data), _i64(idx)
    flat = data.ndim == 2
    B, N = data.shape[:2]
    C = 1 if flat else data.shape[2]
    I = idx.shape[1]
    out = np.zeros((B, I) if flat else (B, I, C), dtype=np.float32)
    _chk(_load().oracle_gather_cl(_p(data), _p(idx), _p(out), B, C, N, I), "gather_cl")
    return out


def scatter_add_cl(gout, idx, N):
    gout, idx = _f32(gout), _i64(idx)
    flat = gout.ndim == 2
    B, I = gout.shape[:2]
    C = 1 if flat else gout.shape[2]
    out = np.zeros((B, N) if flat else (B, N, C), dtype=np.float32)
    _chk(_load().oracle_scatter_add_cl(_p(gout), _p(idx), _p(out), B, C, N, I), "scatter_add_cl")
    return out


def knn_interp_fwd(in_xyz, feat, q_xyz, knn_idx):
    """channel-first: in_xyz [B,3,M], feat [B,C,M], q_xyz [B,3,Nq], knn_idx [B,Nq,k] -> [B,C,Nq]"""
    in_xyz, feat, q_xyz, knn_idx = _f32(in_xyz), _f32(feat), _f32(q_xyz), _i64(knn_idx)
    B, C, M = feat.shape
    Nq, k = knn_idx.shape[1], knn_idx.shape[2]
    out = np.zeros((B, C, Nq), dtype=np.float32)
    _chk(_load().oracle_knn_interp_fwd(_p(in_xyz), _p(feat), _p(q_xyz), _p(knn_idx), _p(out), B, C, M, Nq, k),
         "knn_interp_fwd")
    return out


def pointconv_dw_fwd(feat, weight, idx, k):
    """feat [B,C,M], weight [B,C,N,k], idx int64 [B,N,kk>=k] -> out [B,C,N], arg uint8 [B,C,N]"""
    feat, weight, idx = _f32(feat), _f32(weight), _i64(idx)
    B, C, M = feat.shape
    N = weight.shape[2]
    out = np.zeros((B, C, N), dtype=np.float32)
    arg = np.zeros((B, C, N), dtype=np.uint8)
    _chk(_load().oracle_pointconv_dw_fwd(_p(feat), _p(weight), _p(idx), idx.shape[2], _p(out), _p(arg),
                                         B, C, M, N, k), "pointconv_dw_fwd")
    return out, arg


def pointconv_dw_bwd(gout, feat, weight, idx, arg, k):
    gout, feat, weight, idx = _f32(gout), _f32(feat), _f32(weight), _i64(idx)
    arg = np.ascontiguousarray(arg, dtype=np.uint8)
    B, C, M = feat.shape
    N = weight.shape[2]
    gfeat, gweight = np.zeros_like(feat), np.zeros_like(weight)
    _chk(_load().oracle_pointconv_dw_bwd(_p(gout), _p(feat), _p(weight), _p(idx), idx.shape[2], _p(arg),
                                         _p(gfeat), _p(gweight), B, C, M, N, k), "pointconv_dw_bwd")
    return gfeat, gweight


def knn_interp_bwd(in_xyz, gout, q_xyz, knn_idx, M):
    in_xyz, gout, q_xyz, knn_idx = _f32(in_xyz), _f32(gout), _f32(q_xyz), _i64(knn_idx)
    B, C, Nq = gout.shape
    k = knn_idx.shape[2]
    gfeat = np.zeros((B, C, M), dtype=np.float32)
    _chk(_load().oracle_knn_interp_bwd(_p(in_xyz), _p(gout), _p(q_xyz), _p(knn_idx), _p(gfeat), B, C, M, Nq, k),
         "knn_interp_bwd")
    return gfeat


def knn_interp_bwd_xyz(in_xyz, feat, gout, q_xyz, knn_idx):
    """-> (g_in_xyz [B,3,M], g_q_xyz [B,3,Nq])"""
    in_xyz, feat, gout, q_xyz, knn_idx = _f32(in_xyz), _f32(feat), _f32(gout), _f32(q_xyz), _i64(knn_idx)
    B, C, M = feat.shape
    Nq, k = knn_idx.shape[1], knn_idx.shape[2]
    g_in = np.zeros((B, 3, M), dtype=np.float32)
    g_q = np.zeros((B, 3, Nq), dtype=np.float32)
    _chk(_load().oracle_knn_interp_bwd_xyz(_p(in_xyz), _p(feat), _p(gout), _p(q_xyz), _p(knn_idx), _p(g_in), _p(g_q),
                                           B, C, M, Nq, k), "knn_interp_bwd_xyz")
    return g_in, g_q


def corr3d_gather_fwd(xyz1, xyz2, cost, knn_idx):
    xyz1, xyz2, cost, knn_idx = _f32(xyz1), _f32(xyz2), _f32(cost), _i64(knn_idx)
    B, N, M = cost.shape
    k = knn_idx.shape[2]
    out = np.zeros((B, 4, N, k), dtype=np.float32)
    _chk(_load().oracle_corr3d_gather_fwd(_p(xyz1), _p(xyz2), _p(cost), _p(knn_idx), _p(out), B, N, M, k),
         "corr3d_gather_fwd")
    return out


def pointconv_mix_fwd(feat_cl, wgt, idx, k):
    """feat_cl [B,M,CH], wgt [B,Wn,N,k], idx int64 [B,N,kk>=k] -> [B,N,Wn,CH]"""
    feat_cl, wgt, idx = _f32(feat_cl), _f32(wgt), _i64(idx)
    B, M, CH = feat_cl.shape
    Wn, N = wgt.shape[1], wgt.shape[2]
    out = np.zeros((B, N, Wn, CH), dtype=np.float32)
    _chk(_load().oracle_pointconv_mix_fwd(_p(feat_cl), _p(wgt), _p(idx), idx.shape[2], _p(out), B, M, N, CH, Wn, k),
         "pointconv_mix_fwd")
    return out


def pwc3d_pair_fwd(a, bm, e, idx, slope=0.1):
    a, bm, e, idx = _f32(a), _f32(bm), _f32(e), _i64(idx)
    B, C, N, k = e.shape
    out = np.zeros_like(e)
    _chk(_load().oracle_pwc3d_pair_fwd(_p(a), _p(bm), _p(e), _p(idx), _p(out), B, C, bm.shape[2], N, k,
                                       ctypes.c_float(slope)), "pwc3d_pair_fwd")
    return out


def ksum_fwd(w, h):
    w, h = _f32(w), _f32(h)
    B, C, N, k = w.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    _chk(_load().oracle_ksum_fwd(_p(w), _p(h), _p(out), B, C, N, k), "ksum_fwd")
    return out


def gather_wsum_fwd(w, feat, idx):
    w, feat, idx = _f32(w), _f32(feat), _i64(idx)
    B, C, N, k = w.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    _chk(_load().oracle_gather_wsum_fwd(_p(w), _p(feat), _p(idx), _p(out), B, C, feat.shape[2], N, k), "gather_wsum_fwd")
    return out


def convex_upsample_fwd(flow, mask, scale):
    """flow [B,2,h,w], mask [B,9*S*S,h,w] -> [B,2,h*S,w*S]"""
    flow, mask = _f32(flow), _f32(mask)
    B, _, h, w = flow.shape
    out = np.zeros((B, 2, h * scale, w * scale), dtype=np.float32)
    _chk(_load().oracle_convex_upsample_fwd(_p(flow), _p(mask), _p(out), B, h, w, scale), "convex_upsample_fwd")
    return out


def weightnet_fwd(xyz, centres, idx, k, params, want_hidden=False):
    """xyz [B,3,M], centres [B,3,N], idx int64 [B,N,kk>=k], params = (w1[8,3], b1, w2[32,8], b2, w3[C,32], b3)
    -> out [B,C,N,k] (and h2 [B,32,N,k])"""
    xyz, centres, idx = _f32(xyz), _f32(centres), _i64(idx)
    w1, b1, w2, b2, w3, b3 = [_f32(t) for t in params]
    B, _, M = xyz.shape
    N, C = centres.shape[2], w3.shape[0]
    out = np.zeros((B, C, N, k), dtype=np.float32)
    h2 = np.zeros((B, 32, N, k), dtype=np.float32) if want_hidden else None
    _chk(_load().oracle_weightnet_fwd(_p(xyz), _p(centres), _p(idx), idx.shape[2], _p(w1), _p(b1), _p(w2), _p(b2),
                                      _p(w3), _p(b3), _p(out), _p(h2) if want_hidden else None, B, C, M, N, k),
         "weightnet_fwd")
    return (out, h2) if want_hidden else out


def weightnet_bwd(xyz, centres, idx, k, params, gout):
    """-> float64 gradients [gw1 [8,3], gb1 [8], gw2 [32,8], gb2 [32], gw3 [C,32], gb3 [C]]"""
    xyz, centres, idx, gout = _f32(xyz), _f32(centres), _i64(idx), _f32(gout)
    w1, b1, w2, b2, w3, b3 = [_f32(t) for t in params]
    B, _, M = xyz.shape
    N, C = centres.shape[2], w3.shape[0]
    flat = np.zeros((24 + 8 + 256 + 32 + C * 33,), dtype=np.float64)
    _chk(_load().oracle_weightnet_bwd(_p(xyz), _p(centres), _p(idx), idx.shape[2], _p(w1), _p(b1), _p(w2), _p(b2),
                                      _p(w3), _p(b3), _p(gout), _p(flat), B, C, M, N, k), "weightnet_bwd")
    sizes = [(8, 3), (8,), (32, 8), (32,), (C, 32), (C,)]
    out, pos = [], 0
    for shape in sizes:
        n = int(np.prod(shape))
        out.append(flat[pos:pos + n].reshape(shape))
        pos += n
    return out


def bilinear_sample_fwd(feat, uv):
    """feat [B,C,H,W], uv [B,2,N] pixel coordinates -> [B,C,N]"""
    feat, uv = _f32(feat), _f32(uv)
    B, C, H, W = feat.shape
    N = uv.shape[2]
    out = np.zeros((B, C, N), dtype=np.float32)
    _chk(_load().oracle_bilinear_sample_fwd(_p(feat), _p(uv), _p(out), B, C, H, W, N), "bilinear_sample_fwd")
    return out


def ids_flow_fwd(pc1, flow, origin, f, cx, cy, rw, rh, rm, aw, ah):
    """paral2persp(pc1 + flow) - origin on [B,3,N] arrays; f, cx, cy [B]"""
    pc1, flow, origin, f, cx, cy = [_f32(t) for t in (pc1, flow, origin, f, cx, cy)]
    B, _, N = pc1.shape
    out = np.zeros_like(pc1)
    c = ctypes.c_float
    _chk(_load().oracle_ids_flow_fwd(_p(pc1), _p(flow), _p(origin), _p(f), _p(cx), _p(cy), _p(out), c(rw), c(rh), c(rm),
                                     c(aw), c(ah), B, N), "ids_flow_fwd")
    return out


def persp2paral(pcs, intr, persp_hw, paral_hw):
    """pcs [B,6,N], intr [B,3] -> (pc1, pc2) [B,3,N] in the parallel camera (models/ids.py:4-33)"""
    pcs, intr = _f32(pcs), _f32(intr)
    B, _, N = pcs.shape
    rw = (paral_hw[1] - 1) / (persp_hw[1] - 1)
    rh = (paral_hw[0] - 1) / (persp_hw[0] - 1)
    out1, out2 = np.zeros((B, 3, N), dtype=np.float32), np.zeros((B, 3, N), dtype=np.float32)
    cf = ctypes.c_float
    _chk(_load().oracle_persp2paral(_p(pcs), _p(intr), _p(out1), _p(out2), B, N, cf(rw), cf(rh), cf(min(rw, rh)),
                                    cf((paral_hw[1] - 1) / 2), cf((paral_hw[0] - 1) / 2)), "persp2paral")
    return out1, out2


def project_pc2image(pc, intr, perspective, cx, cy, sx, sy):
    """pc [B,3,N] (+ intr [B,3] for the perspective camera) -> uv [B,2,N] (models/utils.py:234-259 + grid rescale)"""
    pc = _f32(pc)
    B, _, N = pc.shape
    intr = _f32(intr) if intr is not None else np.zeros((B, 3), dtype=np.float32)
    uv = np.zeros((B, 2, N), dtype=np.float32)
    cf = ctypes.c_float
    _chk(_load().oracle_project_pc2image(_p(pc), _p(intr), _p(uv), B, N, int(bool(perspective)), cf(cx), cf(cy), cf(sx), cf(sy)),
         "project_pc2image")
    return uv


def pad_normalize(images, pad, mean, std):
    """images [B,6,H,W], pad = [left, right, 0, bottom] -> (image1, image2) [B,3,Hp,Wp]"""
    images = _f32(images)
    B, _, H, W = images.shape
    left, right, _top, bottom = pad
    Hp, Wp = H + bottom, W + left + right
    out1, out2 = np.zeros((B, 3, Hp, Wp), dtype=np.float32), np.zeros((B, 3, Hp, Wp), dtype=np.float32)
    m3, s3 = _f32(mean), _f32(std)
    _chk(_load().oracle_pad_normalize(_p(images), _p(out1), _p(out2), B, H, W, Hp, Wp, left, _p(m3), _p(s3)), "pad_normalize")
    return out1, out2

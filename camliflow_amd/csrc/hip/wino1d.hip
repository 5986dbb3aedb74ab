// C-ABI of the 1-D Winograd F(4, 5) form of GRU2D's 1x5 / 5x1 convolutions (wino1d.h; models/raft_core.py:110-140): the
// half-step convolutions with their gate / blend arithmetic and the data gradient, on NHWC tensors, as three launches each
// (input transform, 8 plane contractions on the k-contiguous core of convcl.h, output transform + epilogue): 8
// multiplications per 4 outputs and channel pair where the tap convolution of convcl.hip spends 20.
#include "camli_common.h"
#include "wino1d.h"
#include <stdlib.h>

namespace {

constexpr int NBUF = 3;

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NTW>
int launch_planes(const w1d::PlanesBatch& pb, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * (256 + 32 * NTW) * 16 * sizeof(float);
    auto kern = &w1d::planes_cl_kernel<NTW, NBUF>;
    static unsigned long long reserved = 0;
    if (!camli_reserve_lds(reinterpret_cast<const void*>(kern), lds, reserved)) {
        camli_set_error("camli_wino1d: cannot reserve %zu bytes of LDS", lds);
        return CAMLI_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(pb.base.tiles_p * pb.base.tiles_n, 8), dim3(256), lds, s, pb);
    return CAMLI_OK;
}

// x -> V -> Mo for one convolution; the caller finishes with its output transform
int transform_and_contract(const char* what, const float* x0, int ldx0, int C0, const float* x1, int ldx1, int C1, const float* U,
                           int Cout, float* workspace, int64_t workspace_bytes, int B, int H, int W, int axis, w1d::Lines& l, float*& Mo,
                           hipStream_t s, float* V_keep = nullptr) {
    const int C = C0 + C1;
    if (!x0 || !U || !workspace || (C1 > 0 && !x1)) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (B < 1 || H < 1 || W < 1 || (axis != 0 && axis != 1) || C0 < 16 || C0 % 16 || C1 < 0 || C1 % 16 || C / 16 < NBUF - 1 || Cout < 128 ||
        Cout % 128 || ldx0 < C0 || ldx0 % 4 || (C1 > 0 && (ldx1 < C1 || ldx1 % 4))) {
        camli_set_error("%s: unsupported shape B=%d %dx%d C0=%d C1=%d Cout=%d axis=%d (input channels in multiples of 16, output of 128)", what, B,
                        H, W, C0, C1, Cout, axis);
        return CAMLI_ENOTSUP;
    }
    l = w1d::make_lines(B, H, W, axis);
    const int64_t need = (int64_t)8 * l.tiles * ((int64_t)C + Cout) * 4;
    if (workspace_bytes < need) { camli_set_error("%s: workspace of %lld bytes, %lld needed", what, (long long)workspace_bytes, (long long)need); return CAMLI_EINVAL; }
    if ((int64_t)l.tiles * (C > Cout ? C : Cout) * 4 >= (int64_t)0x7FF00000 || (int64_t)B * H * W >= ((int64_t)1 << 29)) {
        camli_set_error("%s: a transform-domain plane beyond 2 GB", what);
        return CAMLI_ENOTSUP;
    }
    if (!aligned16(x0) || !aligned16(x1) || !aligned16(U) || !aligned16(workspace) || !aligned16(V_keep)) { camli_set_error("%s: pointers must be 16-byte aligned", what); return CAMLI_EINVAL; }
    // V_keep: the caller keeps the transformed input [8][tiles][C] (the weight gradient contracts it again: camli_wino1d_wrw's v_in)
    float* V = V_keep ? V_keep : workspace;
    Mo = V_keep ? workspace : workspace + (size_t)8 * l.tiles * C;
    hipLaunchKernelGGL(w1d::input_transform_1d_kernel, dim3(camli_divup(l.tiles, 4)), dim3(256), 0, s, x0, ldx0, C0, C1 > 0 ? x1 : x0, C1 > 0 ? ldx1 : ldx0,
                       C1, V, l.tiles, l);
    w1d::PlanesBatch pb;
    ccl::Problem& p = pb.base;
    p.x = p.x1 = V; p.w = U; p.y = p.y1 = Mo;
    p.B = 1; p.H = 1; p.W = l.tiles; p.Cin = p.C0 = C; p.Cout = p.N0 = Cout; p.T = 1;
    p.ldx = p.ldx1 = C; p.ldw = C; p.ldy = p.ldy1 = Cout;
    p.xk = p.wk = 16;
    p.xrec = p.x1rec = (uint32_t)((int64_t)l.tiles * C * 4); p.wrec = (uint32_t)((int64_t)Cout * C * 4);
    const int NT = Cout % 256 == 0 ? 256 : 128;
    p.tiles_p = camli_divup(l.tiles, 256); p.tiles_n = Cout / NT;
    p.add = p.h = p.z = V; p.y2 = Mo; p.ld_add = p.ld_h = p.ld_z = p.ldy2 = 4;
    p.acc0 = p.acc1 = p.sanitize = 0;
    for (int t = 0; t < ccl::MAX_TAPS; ++t) p.dy[t] = p.dx[t] = 0;
    pb.x_plane = (int64_t)l.tiles * C; pb.w_plane = (int64_t)Cout * C; pb.y_plane = (int64_t)l.tiles * Cout;
    return NT == 256 ? launch_planes<8>(pb, s) : launch_planes<4>(pb, s);
}

template <int EPI>
void launch_output(const float* Mo, const w1d::Epilogue& e, const w1d::Lines& l, hipStream_t s) {
    hipLaunchKernelGGL(w1d::output_transform_1d_kernel<EPI>, dim3(camli_divup(l.tiles, 4)), dim3(256), 0, s, Mo, e, l);
}

}  // namespace

extern "C" int64_t camli_wino1d_workspace_bytes(int B, int H, int W, int Cin, int Cout, int axis) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (axis != 0 && axis != 1)) return 0;
    const w1d::Lines l = w1d::make_lines(B, H, W, axis);
    return (int64_t)8 * l.tiles * ((int64_t)Cin + Cout) * 4;
}

// U [8][N][C] from the packed weights wp [N][5][C] of camli_convcl_* (flip: taps reversed -- with the transposed packing
// [Cin][5][Cout] that is the data gradient's weights)
extern "C" int camli_wino1d_weights(const float* wp, float* U, int N, int C, int flip, void* stream) {
    const char* what = "camli_wino1d_weights";
    if (!wp || !U || N < 1 || C < 1) { camli_set_error("%s: bad arguments", what); return CAMLI_EINVAL; }
    hipLaunchKernelGGL(w1d::weight_transform_1d_kernel, dim3(camli_divup(N * C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wp, U, N, C,
                       flip ? 1 : 0);
    return camli_check_launch(what);
}

// y0 | y1 (= | +=) the 5-tap convolution of cat[x0, x1] along `axis` (camli_convcl_fwd's PLAIN form with dy / dx = the taps of
// a 1 x 5 (axis 0) or 5 x 1 (axis 1) kernel)
extern "C" int camli_wino1d_conv(const float* x0, int ldx0, int C0, const float* x1, int ldx1, int C1, const float* U, float* y0, int ldy0,
                                 int N0, float* y1, int ldy1, float* workspace, int64_t workspace_bytes, int B, int H, int W, int Cout,
                                 int axis, int accumulate0, int accumulate1, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino1d_conv";
    if (!y0 || (N0 < Cout && !y1) || N0 < 4 || N0 > Cout || N0 % 4 || ldy0 < N0 || ldy0 % 4 || (N0 < Cout && (ldy1 < Cout - N0 || ldy1 % 4)) ||
        !aligned16(y0) || !aligned16(y1)) {
        camli_set_error("%s: bad outputs (N0=%d Cout=%d ld %d %d)", what, N0, Cout, ldy0, ldy1);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    w1d::Lines l;
    float* Mo = nullptr;
    const int rc = transform_and_contract(what, x0, ldx0, C0, x1, ldx1, C1, U, Cout, workspace, workspace_bytes, B, H, W, axis, l, Mo, s);
    if (rc != CAMLI_OK) return rc;
    w1d::Epilogue e = {};
    e.N = Cout; e.N0 = N0; e.y = y0; e.ldy = ldy0; e.y1 = N0 < Cout ? y1 : y0; e.ldy1 = N0 < Cout ? ldy1 : ldy0;
    e.acc0 = accumulate0 ? 1 : 0; e.acc1 = accumulate1 ? 1 : 0;
    launch_output<ccl::EPI_PLAIN>(Mo, e, l, s);
    return camli_check_launch(what);
}

// camli_convcl_gru_gates / _gru_blend on the Winograd form: same tensors, same arithmetic in the epilogue
extern "C" int camli_wino1d_gru_gates(const float* h, const float* x, int CX, const float* U_zr, const float* ctx_zr, float* z, float* rh,
                                      float* r, float* v_keep, float* workspace, int64_t workspace_bytes, int B, int H, int W, int axis, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino1d_gru_gates";
    if (!ctx_zr || !z || !rh || !r || !aligned16(ctx_zr) || !aligned16(z) || !aligned16(rh) || !aligned16(r)) { camli_set_error("%s: bad pointers", what); return CAMLI_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    w1d::Lines l;
    float* Mo = nullptr;
    const int rc = transform_and_contract(what, h, 128, 128, x, CX, CX, U_zr, 256, workspace, workspace_bytes, B, H, W, axis, l, Mo, s, v_keep);
    if (rc != CAMLI_OK) return rc;
    w1d::Epilogue e = {};
    e.N = 256; e.N0 = 256; e.y = z; e.ldy = 128; e.y1 = rh; e.ldy1 = 128; e.y2 = r; e.ldy2 = 128;
    e.add = ctx_zr; e.ld_add = 256; e.h = h; e.ld_h = 128;
    launch_output<ccl::EPI_GATES>(Mo, e, l, s);
    return camli_check_launch(what);
}

extern "C" int camli_wino1d_gru_blend(const float* rh, const float* x, int CX, const float* U_q, const float* ctx_q, const float* z,
                                      const float* h, float* h_new, float* q, int nan_to_num, float* v_keep, float* workspace, int64_t workspace_bytes,
                                      int B, int H, int W, int axis, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino1d_gru_blend";
    if (!ctx_q || !z || !h || !h_new || !q || !aligned16(ctx_q) || !aligned16(z) || !aligned16(h) || !aligned16(h_new) || !aligned16(q)) {
        camli_set_error("%s: bad pointers", what);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    w1d::Lines l;
    float* Mo = nullptr;
    const int rc = transform_and_contract(what, rh, 128, 128, x, CX, CX, U_q, 128, workspace, workspace_bytes, B, H, W, axis, l, Mo, s, v_keep);
    if (rc != CAMLI_OK) return rc;
    w1d::Epilogue e = {};
    e.N = 128; e.N0 = 128; e.y = h_new; e.ldy = 128; e.y1 = q; e.ldy1 = 128;
    e.add = ctx_q; e.ld_add = 128; e.h = h; e.ld_h = 128; e.z = z; e.ld_z = 128; e.sanitize = nan_to_num ? 1 : 0;
    launch_output<ccl::EPI_BLEND>(Mo, e, l, s);
    return camli_check_launch(what);
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------
namespace {

struct WrwPlan1d {
    int Sp;          // K splits per plane
    int rows;        // tiles per plane as allocated: a multiple of 16 Sp (the K split is rows / Sp pixels, a multiple of 16)
};

WrwPlan1d wrw_plan_1d(int tiles, int out_tiles, bool exact = false) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    WrwPlan1d pl;
    int sp = cus / (8 * out_tiles);
    if (sp < 1) sp = 1;
    while (sp > 1 && tiles / sp < 48) --sp;          // at least the pipeline's depth per split
    if (exact) {
        // the planes as the forward left them ([8][tiles][C], no padding rows): the K split must divide the plane into whole
        // 16-row steps; rows = 0 when no split within 3/4 of the wanted one does (the caller transforms again)
        const int want = sp;
        while (sp >= 1 && tiles % (16 * sp) != 0) --sp;
        pl.Sp = sp;
        pl.rows = (sp >= 1 && 4 * sp >= 3 * want) ? tiles : 0;
        return pl;
    }
    pl.Sp = sp;
    pl.rows = camli_divup(tiles, 16 * sp) * 16 * sp;
    return pl;
}

template <int TBN>
int launch_wrw_1d(const wrw::Problem& p, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * 16 * (256 + 32 * TBN) * sizeof(float);
    auto kern = &wrw::wrw_kernel<TBN, NBUF>;
    static unsigned long long reserved = 0;
    if (!camli_reserve_lds(reinterpret_cast<const void*>(kern), lds, reserved)) {
        camli_set_error("camli_wino1d_wrw: cannot reserve %zu bytes of LDS", lds);
        return CAMLI_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.S * p.T * p.tiles_m * p.tiles_n), dim3(256), lds, s, p);
    return CAMLI_OK;
}

}  // namespace

extern "C" int64_t camli_wino1d_wrw_workspace_bytes(int B, int H, int W, int Cin, int Cout, int axis) {
    if (B < 1 || H < 1 || W < 1 || Cin < 256 || Cin % 256 || Cout < 128 || Cout % 128 || (axis != 0 && axis != 1)) return 0;
    const w1d::Lines l = w1d::make_lines(B, H, W, axis);
    const WrwPlan1d pl = wrw_plan_1d(l.tiles, (Cin / 256) * (Cout / (Cout % 256 == 0 ? 256 : 128)));
    return ((int64_t)8 * pl.rows * ((int64_t)Cin + Cout) + (int64_t)8 * pl.Sp * Cin * Cout) * 4;
}

// 1 when camli_wino1d_wrw can contract a transformed input kept by the forward (v_keep of camli_wino1d_gru_gates / _blend) as it
// lies -- [8][tiles][Cin], no padding rows -- on a K split that fills the device; 0: it transforms the input again
extern "C" int camli_wino1d_wrw_reuse(int B, int H, int W, int Cin, int Cout, int axis) {
    if (camli_wino1d_wrw_workspace_bytes(B, H, W, Cin, Cout, axis) == 0) return 0;
    const w1d::Lines l = w1d::make_lines(B, H, W, axis);
    return wrw_plan_1d(l.tiles, (Cin / 256) * (Cout / (Cout % 256 == 0 ? 256 : 128)), true).rows > 0 ? 1 : 0;
}

// gw [Cout][C0 + C1][5] (= the [Cout, Cin, 1, 5] | [Cout, Cin, 5, 1] weight tensor) (= | +=) the weight gradient of the 5-tap
// convolution of cat[x0, x1] along `axis` for the output gradient gy [P][ldg]: camli_convcl_wrw's result, contracted in the
// transform domain (8 planes x (tiles x Cin x Cout) instead of 5 taps x (pixels x Cin x Cout): 2.5 x fewer multiplications).
// C0 + C1 a multiple of 256, Cout of 128.  Deterministic (fixed summation order).  v_in (may be null): the transformed input the
// forward kept (camli_wino1d_wrw_reuse must say 1); x0 / x1 are then not read.
extern "C" int camli_wino1d_wrw(const float* x0, int ldx0, int C0, const float* x1, int ldx1, int C1, const float* gy, int ldg, float* gw,
                                const float* v_in, float* workspace, int64_t workspace_bytes, int B, int H, int W, int Cout, int axis, int accumulate,
                                void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino1d_wrw";
    const int C = C0 + C1;
    if ((!v_in && (!x0 || (C1 > 0 && !x1))) || !gy || !gw || !workspace) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    const int64_t need = camli_wino1d_wrw_workspace_bytes(B, H, W, C, Cout, axis);
    if (need == 0 || C0 < 16 || C0 % 16 || C1 < 0 || C1 % 16 || ldx0 < C0 || ldx0 % 4 || (C1 > 0 && (ldx1 < C1 || ldx1 % 4)) || ldg < Cout || ldg % 4) {
        camli_set_error("%s: unsupported shape B=%d %dx%d C0=%d C1=%d Cout=%d (input channels a multiple of 256, output of 128)", what, B, H, W, C0, C1, Cout);
        return CAMLI_ENOTSUP;
    }
    if (workspace_bytes < need) { camli_set_error("%s: workspace of %lld bytes, %lld needed", what, (long long)workspace_bytes, (long long)need); return CAMLI_EINVAL; }
    if (!aligned16(x0) || !aligned16(x1) || !aligned16(gy) || !aligned16(gw) || !aligned16(workspace)) { camli_set_error("%s: pointers must be 16-byte aligned", what); return CAMLI_EINVAL; }
    const w1d::Lines l = w1d::make_lines(B, H, W, axis);
    const int NB = Cout % 256 == 0 ? 256 : 128;
    const WrwPlan1d pl = wrw_plan_1d(l.tiles, (C / 256) * (Cout / NB), v_in != nullptr);
    if (v_in && (pl.rows == 0 || !aligned16(v_in))) { camli_set_error("%s: the kept transform cannot be contracted as it lies (camli_wino1d_wrw_reuse)", what); return CAMLI_ENOTSUP; }
    if ((int64_t)8 * pl.rows * (C > Cout ? C : Cout) * 4 >= (int64_t)0x7FF00000) { camli_set_error("%s: transform domain beyond 2 GB", what); return CAMLI_ENOTSUP; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* V = v_in ? const_cast<float*>(v_in) : workspace;
    float* gM = v_in ? workspace : workspace + (size_t)8 * pl.rows * C;
    float* parts = gM + (size_t)8 * pl.rows * Cout;
    if (!v_in)
        hipLaunchKernelGGL(w1d::input_transform_1d_kernel, dim3(camli_divup(pl.rows, 4)), dim3(256), 0, s, x0, ldx0, C0, C1 > 0 ? x1 : x0, C1 > 0 ? ldx1 : ldx0,
                           C1, V, pl.rows, l);
    hipLaunchKernelGGL(w1d::grad_transform_1d_kernel, dim3(camli_divup(pl.rows, 4)), dim3(256), 0, s, gy, ldg, Cout, gM, pl.rows, l);
    // the 8 planes end to end are 8 * rows "pixels" of an image one pixel high; one tap, no shift; parts = 8 Sp K ranges
    wrw::Problem p;
    p.x = p.x1 = V; p.zero = camli_zero_page(); p.gy = gM; p.part = parts;
    if (!p.zero) { camli_set_error("%s: cannot resolve the zero page", what); return CAMLI_ELAUNCH; }
    p.B = 1; p.H = 1; p.W = 8 * pl.rows; p.Cin = C; p.Cout = Cout; p.T = 1;
    p.C0 = C; p.ldx = p.ldx1 = C; p.ldg = Cout;
    p.S = 8 * pl.Sp; p.ksplit = pl.rows / pl.Sp;
    p.tiles_m = C / 256; p.tiles_n = Cout / NB;
    for (int t = 0; t < ccl::MAX_TAPS; ++t) p.dy[t] = p.dx[t] = 0;
    const int rc = NB == 256 ? launch_wrw_1d<8>(p, s) : launch_wrw_1d<4>(p, s);
    if (rc != CAMLI_OK) return rc;
    hipLaunchKernelGGL(w1d::wrw_reduce_1d_kernel, dim3(C * (Cout / 128)), dim3(256), 0, s, parts, pl.Sp, gw, C, Cout, accumulate ? 1 : 0);
    return camli_check_launch(what);
}

// C-ABI of the channels-last tap convolution on the fp32 matrix cores (convcl.h, wrwcl.h): forward / data gradient
// (one kernel: the data gradient is the convolution with the negated taps on the transposed packing) and weight gradient.
// Replaces, for GRU2D's 1x5 / 5x1 convolutions (models/raft_core.py:110-140), the library's NHWC implicit-GEMM kernels and
// the layout transposes it wraps around them.
#include "camli_common.h"
#include "wrwcl.h"
#include <stdlib.h>

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}

// 1 KB of zeros per device: the source of padded rows in the weight-gradient kernel.  A zero-initialised __device__ array
// of the module (every device gets its copy when the code object is loaded there): nothing is allocated or filled at run time,
// so the first call may as well happen under stream capture, on any stream, from any thread (r5 advice: the lazily hipMalloc'd
// + hipMemset page touched the legacy stream inside the first launch).
__device__ __attribute__((aligned(1024))) float camli_zero_page[256];

const float* zero_page() {
    static const float* pages[64] = {nullptr};       // address per device (a benign race: every thread computes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(camli_zero_page)) != hipSuccess) return nullptr;
        pages[dev] = static_cast<const float*>(q);
    }
    return pages[dev];
}

}  // namespace
const float* camli_zero_page() { return zero_page(); }
namespace {

int taps_ok(const char* what, int T, const signed char* dy, const signed char* dx) {
    if (T < 1 || T > ccl::MAX_TAPS || !dy || !dx) { camli_set_error("%s: 1 <= T <= %d taps with offset arrays", what, ccl::MAX_TAPS); return 0; }
    return 1;
}

constexpr int NBUF = 3;

// a tap that reaches beyond the image only ever reads padding: legal, it contributes nothing
template <class Problem>
void set_taps(Problem& p, int T, const signed char* dy, const signed char* dx) {
    for (int t = 0; t < T; ++t) { p.dy[t] = dy[t]; p.dx[t] = dx[t]; }
    for (int t = T; t < ccl::MAX_TAPS; ++t) p.dy[t] = p.dx[t] = 0;
}

template <int NTW, int EPI = ccl::EPI_PLAIN>
int launch_conv(const ccl::Problem& p, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * (256 + 32 * NTW) * 16 * sizeof(float);
    static unsigned long long reserved = 0;
    if (!camli_reserve_lds(reinterpret_cast<const void*>(&ccl::convcl_kernel<NTW, NBUF, EPI>), lds, reserved)) {
        camli_set_error("camli_convcl_fwd: cannot reserve %zu bytes of LDS", lds);
        return CAMLI_ELAUNCH;
    }
    const int tiles = p.tiles_p * p.tiles_n;
    const int cus = cu_count();
    hipLaunchKernelGGL((ccl::convcl_kernel<NTW, NBUF, EPI>), dim3(tiles < cus ? tiles : cus), dim3(256), lds, s, p);
    return CAMLI_OK;
}

// r6: tile selection by problem size.  The wide channel tile (256 / 128 channels) is the efficient one when the pixel tiles
// fill the chip; when they do not -- batch 4 at 68 x 120 (configs[3]'s per-rank batch) is 128 pixel tiles on 256 CUs, KITTI
// batch 1 is 29 -- halving the channel tile doubles the workgroups, and a half-empty chip gains more from that than the
// narrower tile loses (DMA issues per MFMA x 1.5).  CAMLI_CONVCL_TILES=wide | narrow forces a choice (A/B, tests).
bool narrow_tiles(int64_t P, int wide_tiles_n) {
    const char* e = getenv("CAMLI_CONVCL_TILES");       // read per call: the tests force either choice on one shape
    if (e && e[0] == 'w') return false;
    if (e && e[0] == 'n') return true;
    return ((P + 255) / 256) * wide_tiles_n * 4 < (int64_t)cu_count() * 3;
}

template <int TBN>
int launch_wrw(const wrw::Problem& p, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * 16 * (256 + 32 * TBN) * sizeof(float);
    static unsigned long long reserved = 0;
    if (!camli_reserve_lds(reinterpret_cast<const void*>(&wrw::wrw_kernel<TBN, NBUF>), lds, reserved)) {
        camli_set_error("camli_convcl_wrw: cannot reserve %zu bytes of LDS", lds);
        return CAMLI_ELAUNCH;
    }
    hipLaunchKernelGGL((wrw::wrw_kernel<TBN, NBUF>), dim3(p.S * p.T * p.tiles_m * p.tiles_n), dim3(256), lds, s, p);
    return CAMLI_OK;
}

// parts of the pixel range: one workgroup per (part, tap, tile), about one per CU; a part is a multiple of 16 pixels
void wrw_split(int P, int T, int tiles, int& S, int& ksplit) {
    int s = cu_count() / (T * tiles);
    if (s < 1) s = 1;
    ksplit = ((P + s - 1) / s + 15) / 16 * 16;
    if (ksplit < 48) ksplit = 48;           // at least the pipeline's depth
    S = (P + ksplit - 1) / ksplit;
}

}  // namespace

extern "C" int camli_convcl_fwd(const float* x0, int ldx0, int C0, const float* x1, int ldx1, int C1, const float* wp, float* y0,
                                int ldy0, int N0, float* y1, int ldy1, int B, int H, int W, int Cout, int T, const signed char* dy,
                                const signed char* dx, int accumulate0, int accumulate1, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_convcl_fwd";
    if (!x0 || !wp || !y0 || (C1 > 0 && !x1) || (N0 < Cout && !y1)) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (!taps_ok(what, T, dy, dx)) return CAMLI_EINVAL;
    const int Cin = C0 + C1;
    const int64_t P = (int64_t)B * H * W;
    const int NT = Cout % 256 == 0 && !narrow_tiles((int64_t)B * H * W, Cout / 256) ? 256 : 128;
    if (B < 0 || H < 1 || W < 1 || C0 < 16 || C0 % 16 || C1 < 0 || C1 % 16 || Cout < 128 || Cout % 128 || N0 < 1 || N0 > Cout ||
        N0 % (NT / 2) || ldx0 < C0 || ldx0 % 4 || (C1 > 0 && (ldx1 < C1 || ldx1 % 4)) || ldy0 < N0 || ldy0 % 4 ||
        (N0 < Cout && (ldy1 < Cout - N0 || ldy1 % 4))) {
        camli_set_error("%s: unsupported shape B=%d %dx%d C0=%d C1=%d Cout=%d N0=%d ld %d %d %d %d (channels in multiples of 16 in, 128 out; "
                        "an output split on a multiple of %d)", what, B, H, W, C0, C1, Cout, N0, ldx0, ldx1, ldy0, ldy1, NT / 2);
        return CAMLI_ENOTSUP;
    }
    const int64_t lim = (int64_t)0x7FF00000;
    if (P * ldx0 * 4 >= lim || P * (int64_t)(C1 > 0 ? ldx1 : 0) * 4 >= lim || P * ldy0 * 4 >= lim || P * (int64_t)(N0 < Cout ? ldy1 : 0) * 4 >= lim ||
        (int64_t)Cout * T * Cin * 4 >= lim || (int64_t)Cin / 16 * T < NBUF - 1) {
        camli_set_error("%s: tensor beyond 2 GB (32-bit offsets behind buffer descriptors) or fewer than %d K steps", what, NBUF - 1);
        return CAMLI_ENOTSUP;
    }
    if (!aligned16(x0) || !aligned16(x1) || !aligned16(wp) || !aligned16(y0) || !aligned16(y1)) {
        camli_set_error("%s: pointers must be 16-byte aligned", what);
        return CAMLI_EINVAL;
    }
    ccl::Problem p;
    p.x = x0; p.x1 = C1 > 0 ? x1 : x0; p.w = wp; p.y = y0; p.y1 = N0 < Cout ? y1 : y0;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.T = T; p.C0 = C0; p.N0 = N0;
    p.ldx = ldx0; p.ldx1 = C1 > 0 ? ldx1 : ldx0; p.ldy = ldy0; p.ldy1 = N0 < Cout ? ldy1 : ldy0;
    p.tiles_p = (int)((P + 255) / 256); p.tiles_n = Cout / NT; p.ldw = T * Cin;
    p.xk = p.wk = 16; p.xrec = (uint32_t)(P * p.ldx * 4); p.x1rec = (uint32_t)(P * p.ldx1 * 4); p.wrec = (uint32_t)((int64_t)Cout * p.ldw * 4);
    p.add = p.h = p.z = x0; p.y2 = y0; p.ld_add = p.ld_h = p.ld_z = p.ldy2 = 4;
    p.acc0 = accumulate0 ? 1 : 0; p.acc1 = accumulate1 ? 1 : 0; p.sanitize = 0;
    set_taps(p, T, dy, dx);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rc = NT == 256 ? launch_conv<8>(p, s) : launch_conv<4>(p, s);
    if (rc != CAMLI_OK) return rc;
    return camli_check_launch(what);
}

namespace {
// The kernel wants 256 channels on its M side and 128 / 256 on its N side.  Input channels on M when they come in multiples of
// 256; otherwise (a 128-channel input, one tensor) the roles are exchanged: M = output channels (gy rows, shifted by the NEGATED
// tap -- sum_p gy[p] x[p + d] = sum_q gy[q - d] x[q], the same border test), N = input channels.  0: unsupported.
int wrw_mode(int C0, int C1, int Cout) {
    const int Cin = C0 + C1;
    if (Cin >= 256 && Cin % 256 == 0 && Cout >= 128 && Cout % 128 == 0) return 1;
    if (C1 == 0 && Cout >= 256 && Cout % 256 == 0 && Cin >= 128 && Cin % 128 == 0) return 2;
    return 0;
}
}  // namespace

extern "C" int64_t camli_convcl_wrw_workspace_bytes(int B, int H, int W, int Cin, int Cout, int T) {
    if (B < 1 || H < 1 || W < 1 || T < 1 || T > ccl::MAX_TAPS) return 0;
    const int mode = wrw_mode(Cin, 0, Cout);
    if (!mode) return 0;
    const int M = mode == 1 ? Cin : Cout, N = mode == 1 ? Cout : Cin;
    const int NB = N % 256 == 0 ? 256 : 128;
    int S, ksplit;
    wrw_split(B * H * W, T, (M / 256) * (N / NB), S, ksplit);
    return (int64_t)S * T * Cin * Cout * (int64_t)sizeof(float);
}

// gw [Cout][Cin][T] (the framework's [Cout, Cin, kh, kw] with the taps in row-major order) = or += the weight gradient
extern "C" int camli_convcl_wrw(const float* x0, int ldx0, int C0, const float* x1, int ldx1, int C1, const float* gy, int ldg,
                                float* workspace, int64_t workspace_bytes, float* gw, int accumulate, int B, int H, int W, int Cout,
                                int T, const signed char* dy, const signed char* dx, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_convcl_wrw";
    if (!x0 || !gy || !workspace || !gw || (C1 > 0 && !x1)) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (!taps_ok(what, T, dy, dx)) return CAMLI_EINVAL;
    const int Cin = C0 + C1;
    const int64_t P = (int64_t)B * H * W;
    const int mode = (B < 0 || H < 1 || W < 1 || C0 < 4 || C0 % 4 || C1 < 0 || C1 % 4) ? 0 : wrw_mode(C0, C1, Cout);
    if (!mode || ldx0 < C0 || ldx0 % 4 || (C1 > 0 && (ldx1 < C1 || ldx1 % 4)) || ldg < Cout || ldg % 4 ||
        P * (int64_t)ldg * 4 >= (int64_t)0x7FF00000 || P * (int64_t)ldx0 * 4 >= (int64_t)0x7FF00000 || P >= ((int64_t)1 << 30)) {
        camli_set_error("%s: unsupported shape B=%d %dx%d C0=%d C1=%d Cout=%d ld %d %d %d (input channels in multiples of 256 and output "
                        "of 128, or one input of a multiple of 128 channels and output channels in multiples of 256)",
                        what, B, H, W, C0, C1, Cout, ldx0, ldx1, ldg);
        return CAMLI_ENOTSUP;
    }
    if (!aligned16(x0) || !aligned16(x1) || !aligned16(gy) || !aligned16(workspace) || !aligned16(gw)) {
        camli_set_error("%s: pointers must be 16-byte aligned", what);
        return CAMLI_EINVAL;
    }
    wrw::Problem p;
    p.zero = zero_page();
    if (!p.zero) { camli_set_error("%s: cannot resolve the zero page", what); return CAMLI_ELAUNCH; }
    p.part = workspace;
    p.B = B; p.H = H; p.W = W; p.T = T;
    if (mode == 1) {
        p.x = x0; p.x1 = C1 > 0 ? x1 : x0; p.gy = gy;
        p.Cin = Cin; p.Cout = Cout; p.C0 = C0;
        p.ldx = ldx0; p.ldx1 = C1 > 0 ? ldx1 : ldx0; p.ldg = ldg;
        set_taps(p, T, dy, dx);
    } else {        // roles exchanged: the kernel's "x" is gy, its "gy" is x
        p.x = gy; p.x1 = gy; p.gy = x0;
        p.Cin = Cout; p.Cout = Cin; p.C0 = Cout;
        p.ldx = ldg; p.ldx1 = ldg; p.ldg = ldx0;
        signed char ndy[ccl::MAX_TAPS], ndx[ccl::MAX_TAPS];
        for (int t = 0; t < T; ++t) { ndy[t] = (signed char)-dy[t]; ndx[t] = (signed char)-dx[t]; }
        set_taps(p, T, ndy, ndx);
    }
    const int NB = p.Cout % 256 == 0 ? 256 : 128;
    p.tiles_m = p.Cin / 256; p.tiles_n = p.Cout / NB;
    wrw_split((int)P, T, p.tiles_m * p.tiles_n, p.S, p.ksplit);
    if (workspace_bytes < (int64_t)p.S * T * Cin * Cout * (int64_t)sizeof(float)) {
        camli_set_error("%s: workspace of %lld bytes, need %lld (camli_convcl_wrw_workspace_bytes)", what, (long long)workspace_bytes,
                        (long long)((int64_t)p.S * T * Cin * Cout * 4));
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rc = NB == 256 ? launch_wrw<8>(p, s) : launch_wrw<4>(p, s);
    if (rc != CAMLI_OK) return rc;
    const size_t n_el = (size_t)T * Cin * Cout;
    hipLaunchKernelGGL(wrw::wrw_reduce_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, workspace, gw, p.S, T, Cin, Cout,
                       accumulate ? 1 : 0, mode == 2 ? 1 : 0);
    return camli_check_launch(what);
}

// ---- GRU2D half-step: convolution + gate arithmetic in one launch (models/raft_core.py:124-130 / 132-138) -------------------
// All tensors dense NHWC: h, z, r, rh, q, h_new [P][HD] with HD = 128, x [P][CX], ctx_zr [P][2 HD], ctx_q [P][HD].
namespace {
int gru_args_ok(const char* what, int B, int H, int W, int CX, int T, const signed char* dy, const signed char* dx) {
    if (!taps_ok(what, T, dy, dx)) return 0;
    const int64_t P = (int64_t)B * H * W;
    if (B < 0 || H < 1 || W < 1 || CX < 16 || CX % 16 || P * (int64_t)(CX > 256 ? CX : 256) * 4 >= (int64_t)0x7FF00000 || (128 + CX) / 16 * T < NBUF - 1) {
        camli_set_error("%s: unsupported shape B=%d %dx%d CX=%d", what, B, H, W, CX);
        return 0;
    }
    return 1;
}
void gru_problem(ccl::Problem& p, const float* a0, const float* x, int CX, const float* wp, int Cout, int B, int H, int W, int T,
                 const signed char* dy, const signed char* dx) {
    const int64_t P = (int64_t)B * H * W;
    p.x = a0; p.x1 = x; p.w = wp;
    p.B = B; p.H = H; p.W = W; p.Cin = 128 + CX; p.Cout = Cout; p.T = T; p.C0 = 128; p.N0 = Cout;
    p.ldx = 128; p.ldx1 = CX;
    p.tiles_p = (int)((P + 255) / 256); p.tiles_n = 1; p.ldw = T * p.Cin;
    p.xk = p.wk = 16; p.xrec = (uint32_t)(P * p.ldx * 4); p.x1rec = (uint32_t)(P * p.ldx1 * 4); p.wrec = (uint32_t)((int64_t)Cout * p.ldw * 4);
    p.acc0 = p.acc1 = p.sanitize = 0;
    set_taps(p, T, dy, dx);
}
}  // namespace

// z, r*h, r = gates(conv(cat[h, x]; wp_zr [256][T][128 + CX]) + ctx_zr, h)
extern "C" int camli_convcl_gru_gates(const float* h, const float* x, int CX, const float* wp_zr, const float* ctx_zr, float* z,
                                      float* rh, float* r, int B, int H, int W, int T, const signed char* dy, const signed char* dx,
                                      void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_convcl_gru_gates";
    if (!h || !x || !wp_zr || !ctx_zr || !z || !rh || !r) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (!gru_args_ok(what, B, H, W, CX, T, dy, dx)) return CAMLI_ENOTSUP;
    if (!aligned16(h) || !aligned16(x) || !aligned16(wp_zr) || !aligned16(ctx_zr) || !aligned16(z) || !aligned16(rh) || !aligned16(r)) {
        camli_set_error("%s: pointers must be 16-byte aligned", what);
        return CAMLI_EINVAL;
    }
    ccl::Problem p;
    gru_problem(p, h, x, CX, wp_zr, 256, B, H, W, T, dy, dx);
    p.y = z; p.ldy = 128; p.y1 = rh; p.ldy1 = 128; p.y2 = r; p.ldy2 = 128;
    p.add = ctx_zr; p.ld_add = 256; p.h = h; p.ld_h = 128; p.z = h; p.ld_z = 128;
    int rc;
    if (narrow_tiles((int64_t)B * H * W, 1)) { p.tiles_n = 2; rc = launch_conv<4, ccl::EPI_GATES>(p, reinterpret_cast<hipStream_t>(stream)); }
    else rc = launch_conv<8, ccl::EPI_GATES>(p, reinterpret_cast<hipStream_t>(stream));
    if (rc != CAMLI_OK) return rc;
    return camli_check_launch(what);
}

// q = tanh(conv(cat[rh, x]; wp_q [128][T][128 + CX]) + ctx_q);  h_new = (1 - z) h + z q  (nan_to_num: followed by torch.nan_to_num)
extern "C" int camli_convcl_gru_blend(const float* rh, const float* x, int CX, const float* wp_q, const float* ctx_q, const float* z,
                                      const float* h, float* h_new, float* q, int nan_to_num, int B, int H, int W, int T,
                                      const signed char* dy, const signed char* dx, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_convcl_gru_blend";
    if (!rh || !x || !wp_q || !ctx_q || !z || !h || !h_new || !q) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (!gru_args_ok(what, B, H, W, CX, T, dy, dx)) return CAMLI_ENOTSUP;
    if (!aligned16(rh) || !aligned16(x) || !aligned16(wp_q) || !aligned16(ctx_q) || !aligned16(z) || !aligned16(h) || !aligned16(h_new) ||
        !aligned16(q)) {
        camli_set_error("%s: pointers must be 16-byte aligned", what);
        return CAMLI_EINVAL;
    }
    ccl::Problem p;
    gru_problem(p, rh, x, CX, wp_q, 128, B, H, W, T, dy, dx);
    p.y = h_new; p.ldy = 128; p.y1 = q; p.ldy1 = 128; p.y2 = q; p.ldy2 = 128;
    p.add = ctx_q; p.ld_add = 128; p.h = h; p.ld_h = 128; p.z = z; p.ld_z = 128;
    p.sanitize = nan_to_num ? 1 : 0;
    int rc;
    if (narrow_tiles((int64_t)B * H * W, 1)) { p.tiles_n = 2; rc = launch_conv<2, ccl::EPI_BLEND>(p, reinterpret_cast<hipStream_t>(stream)); }
    else rc = launch_conv<4, ccl::EPI_BLEND>(p, reinterpret_cast<hipStream_t>(stream));
    if (rc != CAMLI_OK) return rc;
    return camli_check_launch(what);
}

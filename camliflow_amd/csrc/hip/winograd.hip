// C-ABI of the Winograd F(M x M, 3 x 3) convolution, M = 2 | 4 (winograd.h + gemm_w128.h + winograd_wrw.h): the 3x3
// convolutions of the RAFT update block (models/raft_core.py:148-151 MotionEncoder2D.conv_c2 / conv, :173 FlowHead2D.conv1,
// :188 the mask head) forward, data gradient and weight gradient, replacing the library's fp32 Winograd / implicit-GEMM
// kernels (0.70-0.75 of the fp32 matrix rate counted as a direct convolution; F(2x2): 1.05-1.10, profiles/r06a_winograd_microbench.txt).
#include "camli_common.h"
#include "gemm_w128.h"
#include "winograd.h"
#include "winograd_wrw.h"
#include <stdlib.h>

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int cu_count() {       // rounded down to a multiple of 8: the plane GEMM deals its tiles in 8 chunks (one per XCD)
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        cus[dev] = n >= 8 ? n / 8 * 8 : 8;
    }
    return cus[dev];
}

constexpr int KS = 16, NBUF = 3;

bool tile_ok(int m) { return m == 2 || m == 4; }
int planes_of(int m) { return (m + 2) * (m + 2); }

// the P plane GEMMs Mo[t] = U[t]^T V[t]: one launch, batch = P
template <int GA, int GB, int WM>
int launch_planes(const float* U, const float* V, float* Mo, int Mp, int NT, int K, int P, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * KS * 512 * sizeof(float);
    auto kern = &w128::gemm_w128_kernel<KS, NBUF, 0, GA, GB, WM>;
    static unsigned long long reserved = 0;
    if (!camli_reserve_lds(reinterpret_cast<const void*>(kern), lds, reserved)) {
        camli_set_error("camli_wino_conv3x3: cannot reserve %zu bytes of LDS", lds);
        return CAMLI_ELAUNCH;
    }
    w128::Problem p;
    p.A = U; p.B = V; p.C = Mo; p.M = Mp; p.N = NT; p.K = K;
    p.lda = Mp; p.ldb = NT; p.ldc = NT;
    p.sa = (int64_t)K * Mp; p.sb = (int64_t)K * NT; p.sc = (int64_t)Mp * NT;
    p.alpha = 1.0f;
    p.tiles_m = camli_divup(Mp, w128::tile_m<GA, WM>());
    p.tiles_n = camli_divup(NT, w128::tile_n<GB, WM>());
    p.tiles = P * p.tiles_m * p.tiles_n;
    const int cus = cu_count();
    // the kernel's tile order deals 8 chunks (one per XCD) of gridDim.x / 8 consecutive tiles per round.  Every workgroup
    // walks ceil(tiles / workgroups) tiles whatever the count, so the launch takes the FEWEST workgroups that still finish in
    // that many rounds: 576 tiles (256 -> 192 at tile 4) are three rounds on 256 CUs and on 192 alike -- the other 64 CUs
    // stay free for the point lane's kernels (CAMLI_WINO_GRID=full: one workgroup per CU, A/B)
    static const bool full = []() { const char* e = getenv("CAMLI_WINO_GRID"); return e && e[0] == 'f'; }();
    const int rounds = camli_divup(p.tiles, cus);
    int nwg = p.tiles < cus ? (p.tiles + 7) / 8 * 8 : (full ? cus : (camli_divup(p.tiles, rounds) + 7) / 8 * 8);
    if (nwg > cus) nwg = cus;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, p);
    return CAMLI_OK;
}

int mp_of(int M) { return (M + 3) & ~3; }
int kp_of(int K) { return (K + KS - 1) / KS * KS; }

template <int M>
void launch_input(bool vec, bool chunked, const float* x, int64_t x_bs, int64_t plane, const unsigned char* bits, float* V, int C, int rows,
                  const wino::Geometry& g, hipStream_t s) {
    constexpr int TPT = 8 / M, GPC = 16 / TPT;
    const dim3 block(256);
    if (chunked) {
        const dim3 grid(g.NT / 16, camli_divup(rows, 256 / GPC));
        if (vec) hipLaunchKernelGGL((wino::input_transform_kernel<M, true, true>), grid, block, 0, s, x, x_bs, plane, bits, V, C, rows, g);
        else hipLaunchKernelGGL((wino::input_transform_kernel<M, false, true>), grid, block, 0, s, x, x_bs, plane, bits, V, C, rows, g);
    } else {
        const dim3 grid(camli_divup(g.NT / TPT, 256), rows);
        if (vec) hipLaunchKernelGGL((wino::input_transform_kernel<M, true, false>), grid, block, 0, s, x, x_bs, plane, bits, V, C, rows, g);
        else hipLaunchKernelGGL((wino::input_transform_kernel<M, false, false>), grid, block, 0, s, x, x_bs, plane, bits, V, C, rows, g);
    }
}

template <int M>
void launch_output(bool vec, const float* Mo, int Mp, const float* bias, float* y, int64_t y_bs, int64_t plane, int act, int accumulate,
                   unsigned char* y_bits, int N, const wino::Geometry& g, hipStream_t s) {
    constexpr int TPT = 8 / M;
    const dim3 grid(camli_divup(g.NT / TPT, 256), N), block(256);
    if (vec) hipLaunchKernelGGL((wino::output_transform_kernel<M, true>), grid, block, 0, s, Mo, Mp, bias, y, y_bs, plane, act, accumulate, y_bits, N, g);
    else hipLaunchKernelGGL((wino::output_transform_kernel<M, false>), grid, block, 0, s, Mo, Mp, bias, y, y_bs, plane, act, accumulate, y_bits, N, g);
}

template <int M>
void launch_grad(bool vec, const float* gy, int64_t gy_bs, int64_t plane, const unsigned char* bits, float* gM, int N, int rows,
                 const wino::Geometry& g, hipStream_t s) {
    constexpr int TPT = 8 / M, GPC = 16 / TPT;
    const dim3 grid(g.NT / 16, camli_divup(rows, 256 / GPC)), block(256);
    if (vec) hipLaunchKernelGGL((wino::grad_transform_kernel<M, true, true>), grid, block, 0, s, gy, gy_bs, plane, bits, gM, N, rows, g);
    else hipLaunchKernelGGL((wino::grad_transform_kernel<M, false, true>), grid, block, 0, s, gy, gy_bs, plane, bits, gM, N, rows, g);
}

}  // namespace

extern "C" int64_t camli_wino_weight_floats(int K, int M, int tile) {
    return K < 1 || M < 1 || !tile_ok(tile) ? 0 : (int64_t)planes_of(tile) * kp_of(K) * mp_of(M);
}

// U [P][Kp][Mp] from w [Cout][Cin][3][3]: flip = 0: K = Cin, M = Cout (forward); flip = 1: K = Cout, M = Cin, taps reversed
// (the data gradient's weights).  Kp = K rounded up to a multiple of 16, Mp = M to a multiple of 4, zeros beyond K / M.
extern "C" int camli_wino_weights(const float* w, float* U, int Cout, int Cin, int flip, int tile, void* stream) {
    const char* what = "camli_wino_weights";
    if (!w || !U) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (Cout < 1 || Cin < 1 || !tile_ok(tile)) { camli_set_error("%s: bad shape %d x %d, tile %d (2 | 4)", what, Cout, Cin, tile); return CAMLI_EINVAL; }
    const int Kp = kp_of(flip ? Cout : Cin), Mp = mp_of(flip ? Cin : Cout);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(camli_divup(Kp * Mp, 256)), block(256);
    if (tile == 2) hipLaunchKernelGGL(wino::weight_transform_kernel<2>, grid, block, 0, s, w, U, Cout, Cin, Kp, Mp, flip ? 1 : 0);
    else hipLaunchKernelGGL(wino::weight_transform_kernel<4>, grid, block, 0, s, w, U, Cout, Cin, Kp, Mp, flip ? 1 : 0);
    return camli_check_launch(what);
}

extern "C" int64_t camli_wino_workspace_bytes(int B, int C, int N, int H, int W, int tile) {
    if (B < 1 || C < 1 || N < 1 || H < 1 || W < 1 || !tile_ok(tile)) return 0;
    const wino::Geometry g = wino::make_geometry(B, H, W, tile);
    return (int64_t)planes_of(tile) * g.NT * ((int64_t)kp_of(C) + mp_of(N)) * (int64_t)sizeof(float);
}

extern "C" int64_t camli_wino_mask_bytes(int B, int C, int H, int W) {
    return B < 1 || C < 1 || H < 1 || W < 1 ? 0 : (int64_t)B * C * H * ((W + 7) / 8);
}

// y (= or +=) act(conv3x3(x) + bias).  x: B images of C planes H x W, image b at x + b * x_bs (planes dense: a channel slice
// of a wider NCHW tensor is fine); x_bits (optional): activation bits [B][C][H][ceil(W / 8)] bytes, x reads as zero where its
// bit is clear; U = camli_wino_weights(..., tile) [P][Cp][Mp]; y: N planes per image, image stride y_bs; y_bits (optional,
// act != 0): the activation bits of y [B][N][H][ceil(W / 8)], written.
extern "C" int camli_wino_conv3x3(const float* x, int64_t x_bs, const unsigned char* x_bits, const float* U, const float* bias,
                                  float* y, int64_t y_bs, unsigned char* y_bits, float* workspace, int64_t workspace_bytes, int B,
                                  int C, int N, int H, int W, int act, int accumulate, int tile, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino_conv3x3";
    if (!x || !U || !y || !workspace) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (B < 0 || H < 1 || W < 1 || C <= (NBUF - 1) * KS || N < 4 || act < wino::OUT_PLAIN || act > wino::OUT_RELU_FINITE || !tile_ok(tile)) {
        camli_set_error("%s: unsupported shape B=%d C=%d N=%d %dx%d act=%d tile=%d (more than %d input channels; act 0 | 1 | 2; tile 2 | 4)", what,
                        B, C, N, H, W, act, tile, (NBUF - 1) * KS);
        return CAMLI_ENOTSUP;
    }
    const wino::Geometry g = wino::make_geometry(B, H, W, tile);
    const int Mp = mp_of(N), Cp = kp_of(C), P = planes_of(tile);
    const int64_t need = camli_wino_workspace_bytes(B, C, N, H, W, tile);
    if (workspace_bytes < need) { camli_set_error("%s: workspace of %lld bytes, %lld needed", what, (long long)workspace_bytes, (long long)need); return CAMLI_EINVAL; }
    if ((int64_t)(Mp + 256) * g.NT * 4 >= (int64_t)0x7FF00000 || (int64_t)g.NT * P >= ((int64_t)1 << 30)) {
        camli_set_error("%s: a transform-domain plane beyond 2 GB (B=%d N=%d %dx%d)", what, B, N, H, W);
        return CAMLI_ENOTSUP;
    }
    if (!aligned16(U) || !aligned16(workspace)) { camli_set_error("%s: U / workspace must be 16-byte aligned", what); return CAMLI_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* V = workspace;
    float* Mo = workspace + (size_t)P * Cp * g.NT;
    const int64_t plane = (int64_t)H * W;
    const bool vec_in = W % 4 == 0 && aligned16(x) && x_bs % 4 == 0;
    if (tile == 2) launch_input<2>(vec_in, false, x, x_bs, plane, x_bits, V, C, Cp, g, s);
    else launch_input<4>(vec_in, false, x, x_bs, plane, x_bits, V, C, Cp, g, s);
    int rc;
    if (Mp <= 128) rc = launch_planes<2, 1, 1>(U, V, Mo, Mp, g.NT, Cp, P, s);
    else if (Mp <= 192) rc = launch_planes<3, 1, 1>(U, V, Mo, Mp, g.NT, Cp, P, s);
    else rc = launch_planes<2, 2, 2>(U, V, Mo, Mp, g.NT, Cp, P, s);
    if (rc != CAMLI_OK) return rc;
    const bool vec_out = W % 4 == 0 && aligned16(y) && y_bs % 4 == 0;
    if (tile == 2) launch_output<2>(vec_out, Mo, Mp, bias, y, y_bs, plane, act, accumulate ? 1 : 0, y_bits, N, g, s);
    else launch_output<4>(vec_out, Mo, Mp, bias, y, y_bs, plane, act, accumulate ? 1 : 0, y_bits, N, g, s);
    return camli_check_launch(what);
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------
namespace {

// column-side tile of the k-contiguous contraction (32 NTW channels: 128 / 192 / 256) and the padded row count it needs
void pad_cols(int n, int& ntw, int& padded) {
    const int p128 = camli_divup(n, 128) * 128, p192 = camli_divup(n, 192) * 192, p256 = camli_divup(n, 256) * 256;
    ntw = 8; padded = p256;
    if (p192 < padded) { ntw = 6; padded = p192; }
    if (p128 < padded) { ntw = 4; padded = p128; }
}

struct WrwPlan {
    bool swap;               // false: rows = input channels (V), columns = output channels (gM); true: the other way round
    int rows, cols, ntw;     // rows of the row-side operand (as allocated), padded columns, column tile
    int v_rows, g_rows;      // rows V / gM are allocated (and zero-padded) with
    int S, q;                // K splits, 16-float chunks per split
    int64_t part_floats;     // rows x cols
};

WrwPlan wrw_plan(int C, int N, const wino::Geometry& g) {
    WrwPlan pl;
    int ntw_a, pad_a, ntw_b, pad_b;
    pad_cols(N, ntw_a, pad_a);           // A: rows C, columns N
    pad_cols(C, ntw_b, pad_b);           // B: rows N, columns C
    const int64_t cost_a = (int64_t)camli_divup(C, 256) * 256 * pad_a, cost_b = (int64_t)camli_divup(N, 256) * 256 * pad_b;
    pl.swap = cost_b < cost_a;
    const int Cp = kp_of(C);
    if (!pl.swap) { pl.rows = Cp; pl.cols = pad_a; pl.ntw = ntw_a; pl.v_rows = Cp; pl.g_rows = pad_a; }
    else { pl.rows = N; pl.cols = pad_b; pl.ntw = ntw_b; pl.v_rows = pad_b; pl.g_rows = N; }
    const int tiles = camli_divup(pl.rows, 256) * (pl.cols / (32 * pl.ntw));
    const int total = g.NT / 16;
    int S = cu_count() / (planes_of(g.M) * tiles);
    if (S < 1) S = 1;
    if (S > total / 2) S = total / 2 > 0 ? total / 2 : 1;
    int q = camli_divup(total, S);
    S = camli_divup(total, q);
    while (S > 1 && total - (S - 1) * q < 2) { ++q; S = camli_divup(total, q); }      // every split at least the pipeline's depth
    pl.S = S; pl.q = q;
    pl.part_floats = (int64_t)pl.rows * pl.cols;
    return pl;
}

template <int NTW>
int launch_wrw_planes(const wino::WrwBatch& wb, int tiles, int zdim, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * (256 + 32 * NTW) * 16 * sizeof(float);
    auto kern = &wino::wrw_planes_kernel<NTW, NBUF>;
    static unsigned long long reserved = 0;
    if (!camli_reserve_lds(reinterpret_cast<const void*>(kern), lds, reserved)) {
        camli_set_error("camli_wino_wrw: cannot reserve %zu bytes of LDS", lds);
        return CAMLI_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, zdim), dim3(256), lds, s, wb);
    return CAMLI_OK;
}

}  // namespace

extern "C" int64_t camli_wino_wrw_workspace_bytes(int B, int C, int N, int H, int W, int tile) {
    if (B < 1 || C < 1 || N < 1 || H < 1 || W < 1 || !tile_ok(tile)) return 0;
    const wino::Geometry g = wino::make_geometry(B, H, W, tile);
    const WrwPlan pl = wrw_plan(C, N, g);
    const int P = planes_of(tile);
    return ((int64_t)P * g.NT * ((int64_t)pl.v_rows + pl.g_rows) + (int64_t)pl.S * P * pl.part_floats) * (int64_t)sizeof(float);
}

// gw [N][C][3][3] (= | +=) the weight gradient of y = conv3x3(x) for the output gradient gy (gy_bits optional: the
// activation bits [B][N][H][ceil(W / 8)] of the forward, gy reads as zero where its bit is clear).  x [B][C][H][W] (image
// stride x_bs), gy [B][N][H][W] (image stride gy_bs).
// gbias [N] (optional) (= | +=, gbias_accumulate) the bias gradient = the sum of the (masked) output gradient per channel.
extern "C" int camli_wino_wrw(const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, const unsigned char* gy_bits,
                              float* gw, float* gbias, float* workspace, int64_t workspace_bytes, int B, int C, int N, int H, int W,
                              int accumulate, int gbias_accumulate, int tile, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino_wrw";
    if (!x || !gy || !gw || !workspace) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    const int64_t need = camli_wino_wrw_workspace_bytes(B, C, N, H, W, tile);
    if (B < 0 || C < 1 || N < 1 || need == 0) { camli_set_error("%s: unsupported shape B=%d C=%d N=%d %dx%d tile=%d", what, B, C, N, H, W, tile); return CAMLI_ENOTSUP; }
    if (workspace_bytes < need) { camli_set_error("%s: workspace of %lld bytes, %lld needed", what, (long long)workspace_bytes, (long long)need); return CAMLI_EINVAL; }
    const wino::Geometry g = wino::make_geometry(B, H, W, tile);
    const WrwPlan pl = wrw_plan(C, N, g);
    const int P = planes_of(tile), A = tile + 2;
    if ((int64_t)(pl.v_rows > pl.g_rows ? pl.v_rows : pl.g_rows) * g.NT * 4 >= (int64_t)0x7FF00000 || pl.part_floats * 4 >= (int64_t)0x7FF00000) {
        camli_set_error("%s: a transform-domain plane beyond 2 GB (B=%d C=%d N=%d %dx%d)", what, B, C, N, H, W);
        return CAMLI_ENOTSUP;
    }
    if (!aligned16(workspace)) { camli_set_error("%s: workspace must be 16-byte aligned", what); return CAMLI_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* V = workspace;
    float* gM = V + (size_t)P * pl.v_rows * g.NT;
    float* parts = gM + (size_t)P * pl.g_rows * g.NT;
    const int64_t plane = (int64_t)H * W;
    // both operands in the chunk-major layout [P][NT / 16][rows][16]: a K chunk of all rows is one contiguous block
    const bool vec_x = W % 4 == 0 && aligned16(x) && x_bs % 4 == 0, vec_g = W % 4 == 0 && aligned16(gy) && gy_bs % 4 == 0;
    if (tile == 2) {
        launch_input<2>(vec_x, true, x, x_bs, plane, nullptr, V, C, pl.v_rows, g, s);
        launch_grad<2>(vec_g, gy, gy_bs, plane, gy_bits, gM, N, pl.g_rows, g, s);
    } else {
        launch_input<4>(vec_x, true, x, x_bs, plane, nullptr, V, C, pl.v_rows, g, s);
        launch_grad<4>(vec_g, gy, gy_bs, plane, gy_bits, gM, N, pl.g_rows, g, s);
    }
    if (gbias) hipLaunchKernelGGL(wino::bias_grad_kernel, dim3(N), dim3(256), 0, s, gM, A + 1, pl.g_rows, g.NT, gbias, gbias_accumulate ? 1 : 0);
    wino::WrwBatch wb;
    ccl::Problem& p = wb.base;
    p.x = p.x1 = pl.swap ? gM : V;
    p.w = pl.swap ? V : gM;
    p.y = p.y1 = parts;
    p.B = 1; p.H = 1; p.W = pl.rows; p.Cin = p.C0 = 16; p.Cout = p.N0 = pl.cols; p.T = 1;
    const int x_rows = pl.swap ? pl.g_rows : pl.v_rows, w_rows = pl.swap ? pl.v_rows : pl.g_rows;
    p.ldx = p.ldx1 = 16; p.ldw = 16; p.ldy = p.ldy1 = pl.cols;
    p.xk = x_rows * 16; p.wk = w_rows * 16;
    p.xrec = p.x1rec = (uint32_t)((int64_t)x_rows * g.NT * 4); p.wrec = (uint32_t)((int64_t)w_rows * g.NT * 4);
    p.tiles_p = camli_divup(pl.rows, 256); p.tiles_n = pl.cols / (32 * pl.ntw);
    p.add = p.h = p.z = V; p.y2 = parts; p.ld_add = p.ld_h = p.ld_z = p.ldy2 = 4;
    p.acc0 = p.acc1 = p.sanitize = 0;
    for (int t = 0; t < ccl::MAX_TAPS; ++t) p.dy[t] = p.dx[t] = 0;
    wb.planes = P;
    wb.q_chunks = pl.q; wb.total_chunks = g.NT / 16;
    wb.x_plane = (int64_t)x_rows * g.NT;
    wb.w_plane = (int64_t)w_rows * g.NT;
    wb.y_part = pl.part_floats;
    const int tiles = p.tiles_p * p.tiles_n;
    const int rc = pl.ntw == 8 ? launch_wrw_planes<8>(wb, tiles, P * pl.S, s) : pl.ntw == 6 ? launch_wrw_planes<6>(wb, tiles, P * pl.S, s)
                                                                                          : launch_wrw_planes<4>(wb, tiles, P * pl.S, s);
    if (rc != CAMLI_OK) return rc;
    // element (c, n) of a part: rows are c (V on the row side) or n (swapped)
    const int64_t sc = pl.swap ? 1 : pl.cols, sn = pl.swap ? pl.cols : 1;
    const dim3 rgrid(camli_divup(N, 64), C);
    if (tile == 2) hipLaunchKernelGGL(wino::wrw_reduce_kernel<2>, rgrid, dim3(256), 0, s, parts, pl.S, pl.part_floats, sc, sn, gw, C, N, accumulate ? 1 : 0);
    else hipLaunchKernelGGL(wino::wrw_reduce_kernel<4>, rgrid, dim3(256), 0, s, parts, pl.S, pl.part_floats, sc, sn, gw, C, N, accumulate ? 1 : 0);
    return camli_check_launch(what);
}

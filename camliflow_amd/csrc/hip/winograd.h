// Winograd F(M x M, 3 x 3), M = 2 | 4, for the 3x3, stride-1, "same" convolutions of the RAFT update block
// (models/raft_core.py:148-151 MotionEncoder2D.conv_c2 / conv, :173 FlowHead2D.conv1, :188 the mask head's first
// convolution), fp32, gfx950.
//
//     y[b][n][oy][ox] = sum_c sum_{i,j} x[b][c][oy + i - 1][ox + j - 1] * w[n][c][i][j]          zero outside the image
//
// as  Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A  per M x M output tile: (M + 2)^2 multiplications per M^2 outputs and
// channel pair instead of 9 M^2 -- 16 per 4 (2.25 x fewer) for M = 2, 36 per 16 (4 x fewer) for M = 4.  The sum over c for
// the P = (M + 2)^2 positions of the transform domain is P independent GEMMs
//
//     Mo[t][n][tile] = sum_c U[t][c][n] * V[t][c][tile]           t = 0 .. P - 1
//
// on the fp32 matrix cores (gemm_w128.h, batch = the P planes, M = output channels, N = tiles, K = input channels), paid
// for with HBM traffic: V is P / M^2 x the input (4 x | 2.25 x), Mo the same multiple of the output.  Three launches:
//
//   input_transform   x NCHW [B][C][H][W]  ->  V  [P][C][NT]       one thread = one channel x 8 output columns (4 | 2 tiles of a
//                     tile row): reads the (M + 2) x 10 input patch (two 16-byte loads + two 4-byte loads per row), writes
//                     P x 16 | 8 bytes, lanes along the tiles.  The data gradient's ReLU mask (activation bits) rides on the loads.
//   gemm_w128         U [P][C][Mp] x V -> Mo [P][Mp][NT]
//   output_transform  Mo -> y NCHW (+ bias, ReLU, activation bits, or += for a gradient accumulated in place): one thread = one
//                     output channel x 8 output columns: P loads, M rows of 8 pixels = 2 M x 16-byte stores.
//
// Tiles: B * TH * TWp of them, TH = ceil(H / M), TWp = ceil(W / M) rounded up to whole 8-column groups, NT = that count
// rounded up to a multiple of 16 (the weight gradient contracts over the tiles in steps of 16); the padding tiles hold zeros
// and are never written back.  The data gradient is the same three launches on the output gradient with the weights
// transposed and the taps reversed (weight_transform with `flip`).
//
// Weight gradient, also in the transform domain (the adjoint of the above with respect to U):
//
//     gU[t][c][n] = sum_tile V[t][c][tile] * gM[t][n][tile],   gM = A gy A^T per tile,      gw[n][c] = G^T gU[.][c][n] G
//
//   input_transform   x  -> V   (as in the forward, chunk-major)
//   grad_transform    gy -> gM  [P][N][NT]   (M x M -> (M + 2) x (M + 2) per tile; the activation bits ride on the loads)
//   wrw_planes        both operands have the contraction index (the tiles) contiguous: the k-contiguous contraction of
//                     convcl.h (64-byte LDS rows, direct-to-LDS loads), one workgroup per (plane, K split, tile), parts
//                     [S][P][rows][cols] written, no atomics
//   wrw_reduce        sum over the K splits in a fixed order, G^T . G, gw (= | +=)
//
// Numerics: not the direct form's summation order.  M = 2: the transforms use additions, subtractions and a multiplication by
// 0.5 (exact); the difference to an fp64 convolution is of the size of the direct fp32 form's own (1.4e-6 against 4.2e-6 max
// abs at 256 channels, unit-variance data).  M = 4: the interpolation points 0, +-1, +-2, inf put factors up to 8 and 1/24
// into the transforms -- the error is ~10 x larger (4e-5 max abs, 3e-6 relative L2 on the same data; cuDNN's fp32 default
// makes the same trade).  tests/test_winograd_gpu.py states the bounds; fused._WINO_TILE chooses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wino {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The 1-D transforms of F(M, 3): bt = B^T d (input), gg = G g (weights), at = A^T m (output), ag = A g (the adjoint of at:
// output gradient -> transform domain), gt = G^T u (transform-domain weight gradient -> taps).  `s` is the element stride of
// the vectors walked (the 2-D transforms apply them along rows, then columns).
template <int M> struct F;

template <> struct F<2> {
    static constexpr int A = 4;
    template <int S> static __device__ __forceinline__ void bt(const float* d, float* o) {
        o[0] = d[0] - d[2 * S]; o[1] = d[S] + d[2 * S]; o[2] = d[2 * S] - d[S]; o[3] = d[S] - d[3 * S];
    }
    static __device__ __forceinline__ void gg(const float* g, float* o) {
        o[0] = g[0]; o[1] = 0.5f * (g[0] + g[1] + g[2]); o[2] = 0.5f * (g[0] - g[1] + g[2]); o[3] = g[2];
    }
    static __device__ __forceinline__ void at(const float* m, float* o) {
        o[0] = m[0] + m[1] + m[2]; o[1] = m[1] - m[2] - m[3];
    }
    static __device__ __forceinline__ void ag(const float* g, float* o) {
        o[0] = g[0]; o[1] = g[0] + g[1]; o[2] = g[0] - g[1]; o[3] = -g[1];
    }
    static __device__ __forceinline__ void gt(const float* u, float* o) {
        o[0] = u[0] + 0.5f * (u[1] + u[2]); o[1] = 0.5f * (u[1] - u[2]); o[2] = 0.5f * (u[1] + u[2]) + u[3];
    }
};

template <> struct F<4> {
    static constexpr int A = 6;
    template <int S> static __device__ __forceinline__ void bt(const float* d, float* o) {
        const float d0 = d[0], d1 = d[S], d2 = d[2 * S], d3 = d[3 * S], d4 = d[4 * S], d5 = d[5 * S];
        o[0] = 4.f * d0 - 5.f * d2 + d4;
        o[1] = (d3 + d4) - 4.f * (d1 + d2);
        o[2] = (d4 - d3) + 4.f * (d1 - d2);
        o[3] = (d4 - d2) + 2.f * (d3 - d1);
        o[4] = (d4 - d2) - 2.f * (d3 - d1);
        o[5] = 4.f * d1 - 5.f * d3 + d5;
    }
    static __device__ __forceinline__ void gg(const float* g, float* o) {
        o[0] = 0.25f * g[0];
        o[1] = (-1.f / 6.f) * (g[0] + g[1] + g[2]);
        o[2] = (-1.f / 6.f) * (g[0] - g[1] + g[2]);
        o[3] = (1.f / 24.f) * g[0] + (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
        o[4] = (1.f / 24.f) * g[0] - (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
        o[5] = g[2];
    }
    static __device__ __forceinline__ void at(const float* m, float* o) {
        const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        o[0] = m[0] + s12 + s34;
        o[1] = d12 + 2.f * d34;
        o[2] = s12 + 4.f * s34;
        o[3] = d12 + 8.f * d34 + m[5];
    }
    static __device__ __forceinline__ void ag(const float* g, float* o) {
        o[0] = g[0];
        o[1] = (g[0] + g[2]) + (g[1] + g[3]);
        o[2] = (g[0] + g[2]) - (g[1] + g[3]);
        o[3] = (g[0] + 4.f * g[2]) + (2.f * g[1] + 8.f * g[3]);
        o[4] = (g[0] + 4.f * g[2]) - (2.f * g[1] + 8.f * g[3]);
        o[5] = g[3];
    }
    static __device__ __forceinline__ void gt(const float* u, float* o) {
        o[0] = 0.25f * u[0] - (1.f / 6.f) * (u[1] + u[2]) + (1.f / 24.f) * (u[3] + u[4]);
        o[1] = (1.f / 6.f) * (u[2] - u[1]) + (1.f / 12.f) * (u[3] - u[4]);
        o[2] = (1.f / 6.f) * ((u[3] + u[4]) - (u[1] + u[2])) + u[5];
    }
};

struct Geometry {
    int B, H, W;
    int M;                   // output tile edge (2 | 4)
    int TH, TWp;             // tile rows, tile columns padded to whole 8-column groups
    int tiles;               // B * TH * TWp
    int NT;                  // tiles rounded up to a multiple of 16: the row length of V / Mo / gM
};

__host__ __device__ inline Geometry make_geometry(int B, int H, int W, int M) {
    Geometry g;
    g.B = B; g.H = H; g.W = W; g.M = M;
    g.TH = (H + M - 1) / M;
    g.TWp = ((W + 7) / 8) * (8 / M);
    g.tiles = B * g.TH * g.TWp;
    g.NT = (g.tiles + 15) & ~15;
    return g;
}

// ---- weights: U[t][k][m] = (G g G^T)[t / A][t % A],  g = w[m][k] (forward: k = input channel, m = output channel) or
// g = w[k][m] rotated by 180 degrees (data gradient: k = output channel, m = input channel).  U is [P][Kp][Mp], Kp >= K,
// Mp >= M: zeros beyond K / M.  w is [Cout][Cin][3][3].
template <int M>
__global__ void weight_transform_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Kp, int Mp, int flip) {
    constexpr int A = F<M>::A;
    const int K = flip ? Cout : Cin, Mr = flip ? Cin : Cout;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kp * Mp) return;
    const int k = i / Mp, m = i - k * Mp;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.f;
            if (m < Mr && k < K) v = flip ? w[((size_t)k * Cin + m) * 9 + (2 - a) * 3 + (2 - b)] : w[((size_t)m * Cin + k) * 9 + a * 3 + b];
            g[a][b] = v;
        }
    float t[A][3];      // G g along the first tap index
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float col[3] = {g[0][b], g[1][b], g[2][b]};
        float o[A];
        F<M>::gg(col, o);
#pragma unroll
        for (int a = 0; a < A; ++a) t[a][b] = o[a];
    }
    const size_t plane = (size_t)Kp * Mp;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        float o[A];
        F<M>::gg(t[a], o);
#pragma unroll
        for (int b = 0; b < A; ++b) U[(size_t)(A * a + b) * plane + i] = o[b];
    }
}

// Where a thread's tiles go.  A thread owns one channel row and one GROUP of 8 output columns = TPT = 8 / M adjacent tiles.
//   CHUNKED = false: [P][rows][NT], thread = (channel blockIdx.y, group): lanes along the tiles (1 KB | 512 bytes per store
//                    instruction);
//   CHUNKED = true:  [P][NT / 16][rows][16] -- the 16 tiles of a K chunk of the weight gradient's contraction contiguous per
//                    row, consecutive rows 64 bytes apart (what its direct-to-LDS loads fetch 1 KB at a time); a block is one
//                    chunk x (256 / GPC) rows, GPC = 16 / TPT groups per chunk: a wave stores whole 64-byte rows.
template <int M, bool CHUNKED>
struct Slot {
    static constexpr int TPT = 8 / M, GPC = 16 / TPT;
    int row, group;
    __device__ __forceinline__ Slot() {
        if (CHUNKED) { row = (256 / GPC) * blockIdx.y + threadIdx.x / GPC; group = GPC * blockIdx.x + threadIdx.x % GPC; }
        else { row = blockIdx.y; group = blockIdx.x * blockDim.x + threadIdx.x; }
    }
    // first of the thread's TPT floats in plane t
    __device__ __forceinline__ size_t at(int t, int rows, int NT) const {
        if (CHUNKED) return (((size_t)t * (NT >> 4) + group / GPC) * rows + row) * 16 + TPT * (group % GPC);
        return ((size_t)t * rows + row) * NT + (size_t)TPT * group;
    }
};

template <int M> struct TileVec;
template <> struct TileVec<2> { typedef f32x4 type; };
template <> struct TileVec<4> { typedef f32x2 type; };

// ---- input: V[t][c][tile] = (B^T d B)[t / A][t % A], d = the A x A patch of x around output tile `tile`
// x: channel c of image b at x + b * sxb + c * sxc (H x W, contiguous rows).  mask (optional): the activation bits an
// output transform left behind -- [B][C][H][W8] bytes, W8 = ceil(W / 8), bit j of byte s = pixel 8 s + j "passes the
// gradient" -- x reads as zero where its bit is clear (the ReLU adjoint at 1/32 of the bytes of re-reading the output).
// VEC: W % 4 == 0 and 16-byte aligned rows -> 16-byte loads.  `rows` >= C channel rows are written (zeros for c >= C).
template <int M, bool VEC, bool CHUNKED = false>
__global__ __launch_bounds__(256) void input_transform_kernel(const float* __restrict__ x, int64_t sxb, int64_t sxc,
                                                             const unsigned char* __restrict__ mask,
                                                             float* __restrict__ V, int C, int rows, Geometry g) {
    constexpr int A = F<M>::A, TPT = 8 / M;
    typedef typename TileVec<M>::type vec;
    const Slot<M, CHUNKED> slot;
    const int i = slot.group, c = slot.row;
    if (i >= g.NT / TPT || c >= rows) return;
    if (c >= C || TPT * i >= g.tiles) {               // padding channel of the contraction / padding tiles
        vec z;
#pragma unroll
        for (int e = 0; e < TPT; ++e) z[e] = 0.f;
#pragma unroll
        for (int t = 0; t < A * A; ++t) *reinterpret_cast<vec*>(V + slot.at(t, rows, g.NT)) = z;
        return;
    }
    const int qpr = g.TWp / TPT;                      // groups per tile row
    const int tq = i % qpr, ty = (i / qpr) % g.TH, b = i / (qpr * g.TH);
    const float* xc = x + b * sxb + c * sxc;
    const int W8 = (g.W + 7) >> 3;
    const unsigned char* mc = mask ? mask + ((size_t)b * C + c) * g.H * W8 : nullptr;
    const int col0 = 8 * tq;                          // first output column of the group; the patch spans col0 - 1 .. col0 + 8
    float d[A][10];
#pragma unroll
    for (int rr = 0; rr < A; ++rr) {
        const int row = M * ty - 1 + rr;
        const bool rok = (unsigned)row < (unsigned)g.H;
        const float* xr = xc + (int64_t)row * g.W;
        if (VEC && rok && col0 + 8 <= g.W) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + col0), bq = *reinterpret_cast<const f32x4*>(xr + col0 + 4);
            d[rr][0] = col0 > 0 ? xr[col0 - 1] : 0.f;
            d[rr][9] = col0 + 8 < g.W ? xr[col0 + 8] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[rr][1 + e] = a[e]; d[rr][5 + e] = bq[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 10; ++e) {
                const int col = col0 - 1 + e;
                d[rr][e] = rok && (unsigned)col < (unsigned)g.W ? xr[col] : 0.f;
            }
        }
        if (mc && rok) {
            // bits of columns col0 - 1 .. col0 + 8: the top bit of the byte to the left, this group's byte, bit 0 of the next
            const unsigned char* mrow = mc + (size_t)row * W8;
            const unsigned left = tq > 0 ? mrow[tq - 1] : 0u, mid = tq < W8 ? mrow[tq] : 0u, right = tq + 1 < W8 ? mrow[tq + 1] : 0u;
            const unsigned bits = (left >> 7) | (mid << 1) | ((right & 1u) << 9);
#pragma unroll
            for (int e = 0; e < 10; ++e)
                if (!((bits >> e) & 1u)) d[rr][e] = 0.f;
        }
    }
    // B^T d along the rows of the patch (per column), then along the columns of each tile
    float rws[A][10];
#pragma unroll
    for (int e = 0; e < 10; ++e) {
        float o[A];
        F<M>::template bt<10>(&d[0][e], o);
#pragma unroll
        for (int a = 0; a < A; ++a) rws[a][e] = o[a];
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
        vec v[A];
#pragma unroll
        for (int tl = 0; tl < TPT; ++tl) {
            float o[A];
            F<M>::template bt<1>(&rws[a][M * tl], o);
#pragma unroll
            for (int bq = 0; bq < A; ++bq) v[bq][tl] = o[bq];
        }
#pragma unroll
        for (int bq = 0; bq < A; ++bq) *reinterpret_cast<vec*>(V + slot.at(A * a + bq, rows, g.NT)) = v[bq];
    }
}

// ---- output: y[b][n][M ty + i][M tx + j] = (A^T m A)[i][j] (+ bias[n]) (ReLU) (+= y), m = Mo[.][n][tile] as A x A
// Mo [P][Mp][NT]; y: channel n of image b at y + b * syb + n * syc.  One thread: channel n, one group of 8 output columns.
// bits (optional, with an activation): [B][N][H][W8] bytes, bit j of byte s = output pixel 8 s + j passes the gradient
// (pre-activation > 0; OUT_RELU_FINITE: and finite) -- the mask the adjoint's transforms read.
// OUT_RELU_FINITE: ReLU, then torch.nan_to_num (models/raft_core.py:163-164): NaN -> 0 (fmaxf drops it), +inf -> FLT_MAX
enum { OUT_PLAIN = 0, OUT_RELU = 1, OUT_RELU_FINITE = 2 };
template <int M, bool VEC>
__global__ __launch_bounds__(256) void output_transform_kernel(const float* __restrict__ Mo, int Mp, const float* __restrict__ bias,
                                                              float* __restrict__ y, int64_t syb, int64_t syc, int act,
                                                              int accumulate, unsigned char* __restrict__ bits, int N, Geometry g) {
    constexpr int A = F<M>::A, TPT = 8 / M;
    typedef typename TileVec<M>::type vec;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (TPT * i >= g.tiles) return;
    const int qpr = g.TWp / TPT;
    const int tq = i % qpr, ty = (i / qpr) % g.TH, b = i / (qpr * g.TH);
    const size_t plane = (size_t)Mp * g.NT;
    const float* in = Mo + (size_t)n * g.NT + (size_t)TPT * i;
    // r[i][b][tl] = sum_a A^T[i][a] m[a][b][tl]: the first index of the transform domain folded into the M output rows
    float r[M][A][TPT];
#pragma unroll
    for (int bq = 0; bq < A; ++bq) {
        vec m[A];
#pragma unroll
        for (int a = 0; a < A; ++a) m[a] = *reinterpret_cast<const vec*>(in + (size_t)(A * a + bq) * plane);
#pragma unroll
        for (int tl = 0; tl < TPT; ++tl) {
            float col[A], o[M];
#pragma unroll
            for (int a = 0; a < A; ++a) col[a] = m[a][tl];
            F<M>::at(col, o);
#pragma unroll
            for (int ri = 0; ri < M; ++ri) r[ri][bq][tl] = o[ri];
        }
    }
    const float bv = bias ? bias[n] : 0.f;
    float* yc = y + b * syb + n * syc;
    const int col0 = 8 * tq;
#pragma unroll
    for (int ri = 0; ri < M; ++ri) {
        const int row = M * ty + ri;
        float px[8];
#pragma unroll
        for (int tl = 0; tl < TPT; ++tl) {
            float col[A], o[M];
#pragma unroll
            for (int bq = 0; bq < A; ++bq) col[bq] = r[ri][bq][tl];
            F<M>::at(col, o);
#pragma unroll
            for (int j = 0; j < M; ++j) px[M * tl + j] = o[j] + bv;
        }
        if (row >= g.H) continue;
        float* yr = yc + (int64_t)row * g.W + col0;
        if (accumulate) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col0 + e < g.W) px[e] += yr[e];
        }
        if (act != OUT_PLAIN) {
            unsigned mbits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mbits |= (px[e] > 0.f && (act != OUT_RELU_FINITE || px[e] <= 3.402823466e+38f) ? 1u : 0u) << e;
                px[e] = fmaxf(px[e], 0.f);
                if (act == OUT_RELU_FINITE) px[e] = fminf(px[e], 3.402823466e+38f);
            }
            if (bits) {
                const int W8 = (g.W + 7) >> 3;
                const int live = g.W - col0;            // pixels of this byte inside the image: the bits beyond stay clear
                if (live < 8) mbits &= (1u << (live > 0 ? live : 0)) - 1u;
                if (tq < W8) bits[(((size_t)b * N + n) * g.H + row) * W8 + tq] = (unsigned char)mbits;
            }
        }
        if (VEC && col0 + 8 <= g.W) {
            const f32x4 lo = {px[0], px[1], px[2], px[3]}, hi = {px[4], px[5], px[6], px[7]};
            *reinterpret_cast<f32x4*>(yr) = lo;
            *reinterpret_cast<f32x4*>(yr + 4) = hi;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col0 + e < g.W) yr[e] = px[e];
        }
    }
}

// ---- output gradient: gM[t][n][tile] = (A gy A^T)[t / A][t % A] (A = the (M + 2) x M matrix whose transpose finishes the forward)
// gy: channel n of image b at gy + b * sgb + n * sgc; mask optional: the activation bits [B][N][H][W8] of the forward's
// output transform, gy reads as zero where its bit is clear.
// gM has `rows` >= N channel rows (zeros for n >= N and for the padding tiles) in either layout of input_transform_kernel.
template <int M, bool VEC, bool CHUNKED = false>
__global__ __launch_bounds__(256) void grad_transform_kernel(const float* __restrict__ gy, int64_t sgb, int64_t sgc,
                                                            const unsigned char* __restrict__ mask,
                                                            float* __restrict__ gM, int N, int rows, Geometry g) {
    constexpr int A = F<M>::A, TPT = 8 / M;
    typedef typename TileVec<M>::type vec;
    const Slot<M, CHUNKED> slot;
    const int i = slot.group, n = slot.row;
    if (i >= g.NT / TPT || n >= rows) return;
    if (n >= N || TPT * i >= g.tiles) {
        vec z;
#pragma unroll
        for (int e = 0; e < TPT; ++e) z[e] = 0.f;
#pragma unroll
        for (int t = 0; t < A * A; ++t) *reinterpret_cast<vec*>(gM + slot.at(t, rows, g.NT)) = z;
        return;
    }
    const int qpr = g.TWp / TPT;
    const int tq = i % qpr, ty = (i / qpr) % g.TH, b = i / (qpr * g.TH);
    const float* gc = gy + b * sgb + n * sgc;
    const int W8 = (g.W + 7) >> 3;
    const unsigned char* mc = mask ? mask + ((size_t)b * N + n) * g.H * W8 : nullptr;
    const int col0 = 8 * tq;
    float d[M][8];
#pragma unroll
    for (int rr = 0; rr < M; ++rr) {
        const int row = M * ty + rr;
        const bool rok = row < g.H;
        const float* gr = gc + (int64_t)row * g.W + col0;
        if (VEC && rok && col0 + 8 <= g.W) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(gr), bq = *reinterpret_cast<const f32x4*>(gr + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[rr][e] = a[e]; d[rr][4 + e] = bq[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) d[rr][e] = rok && col0 + e < g.W ? gr[e] : 0.f;
        }
        if (mc && rok && tq < W8) {
            const unsigned m = mc[(size_t)row * W8 + tq];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (!((m >> e) & 1u)) d[rr][e] = 0.f;
        }
    }
    // A g along the rows (per column), then along the columns of each tile
    float rw[A][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float col[M], o[A];
#pragma unroll
        for (int rr = 0; rr < M; ++rr) col[rr] = d[rr][e];
        F<M>::ag(col, o);
#pragma unroll
        for (int a = 0; a < A; ++a) rw[a][e] = o[a];
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
        vec v[A];
#pragma unroll
        for (int tl = 0; tl < TPT; ++tl) {
            float o[A];
            F<M>::ag(&rw[a][M * tl], o);
#pragma unroll
            for (int bq = 0; bq < A; ++bq) v[bq][tl] = o[bq];
        }
#pragma unroll
        for (int bq = 0; bq < A; ++bq) *reinterpret_cast<vec*>(gM + slot.at(A * a + bq, rows, g.NT)) = v[bq];
    }
}

// ---- gbias[n] (= | +=) sum over every pixel of the (masked) output gradient = the row sum of plane (1, 1) of gM: row 1 of A
// is all ones for both tile sizes, so (A g A^T)[1][1] is the sum of the tile's pixels.  gM chunk-major [P][NT / 16][rows][16].
// One block per channel, fixed summation tree.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ gM, int sum_plane, int rows, int NT, float* __restrict__ gbias,
                                                       int accumulate) {
    __shared__ float part[256];
    const int n = blockIdx.x, chunks = NT >> 4;
    const float* q = gM + (size_t)sum_plane * rows * NT + (size_t)n * 16;
    // a thread takes whole K chunks (the channel's 16 tiles of a chunk are 64 contiguous bytes): four independent 16-byte loads
    // per trip (first form: one float per thread and trip, 64 dependent trips on 192 blocks: the launch took longer than the
    // masked transforms it rides behind)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < chunks; k += 256) {
        const f32x4* c4 = reinterpret_cast<const f32x4*>(q + (size_t)k * rows * 16);
        acc += (c4[0] + c4[1]) + (c4[2] + c4[3]);
    }
    part[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) gbias[n] = accumulate ? gbias[n] + part[0] : part[0];
}

// ---- gw[n][c][i][j] (= | +=) sum_{a,b} G[a][i] G[b][j] * sum_s part[s][A a + b][.]: element (c, n) of a part at
// c * sc + n * sn (the contraction ran with either operand on its row side).  Block = 64 n x 4 plane groups: a thread sums
// the S parts of its P / 4 planes (independent loads, lanes along n), the plane sums of an n meet in LDS, the first plane
// group finishes with G^T . G.  (r6, first form: one thread per (c, n) walking all P S loads in turn -- 69 us for 47 MB,
// a latency chain on 768 waves.)
template <int M>
__global__ __launch_bounds__(256) void wrw_reduce_kernel(const float* __restrict__ parts, int S, int64_t part_floats, int64_t sc, int64_t sn,
                                                        float* __restrict__ gw, int C, int N, int accumulate) {
    constexpr int A = F<M>::A, P = A * A, PG = P / 4;
    __shared__ float us[P][64];
    const int nl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl, c = blockIdx.y;
    float acc[PG];
#pragma unroll
    for (int k = 0; k < PG; ++k) acc[k] = 0.f;
    if (n < N) {
        const float* q = parts + (size_t)(PG * tg) * part_floats + c * sc + n * sn;
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int k = 0; k < PG; ++k) acc[k] += q[((size_t)s * P + k) * part_floats];
        }
    }
#pragma unroll
    for (int k = 0; k < PG; ++k) us[PG * tg + k][nl] = acc[k];
    __syncthreads();
    if (tg != 0 || n >= N) return;
    float r[3][A];      // G^T applied to the first index
#pragma unroll
    for (int b = 0; b < A; ++b) {
        float col[A], o[3];
#pragma unroll
        for (int a = 0; a < A; ++a) col[a] = us[A * a + b][nl];
        F<M>::gt(col, o);
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i][b] = o[i];
    }
    float* out = gw + ((size_t)n * C + c) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float o[3];
        F<M>::gt(r[i], o);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (accumulate) out[3 * i + j] += o[j];
            else out[3 * i + j] = o[j];
        }
    }
}

}  // namespace wino

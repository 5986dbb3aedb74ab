// Winograd F(2x2, 3x3) for the 3x3, stride-1, "same" convolutions of the RAFT update block (models/raft_core.py:148-151
// MotionEncoder2D.conv_c2 / conv, :173 FlowHead2D.conv1, :188 the mask head's first convolution), fp32, gfx950.
//
//     y[b][n][oy][ox] = sum_c sum_{i,j} x[b][c][oy + i - 1][ox + j - 1] * w[n][c][i][j]          zero outside the image
//
// as  Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A  per 2 x 2 output tile: 16 multiplications per 4 outputs and channel pair
// instead of 36.  The sum over c for the 16 positions of the 4 x 4 transform domain is 16 independent GEMMs
//
//     Mo[t][n][tile] = sum_c U[t][c][n] * V[t][c][tile]           t = 0 .. 15
//
// on the fp32 matrix cores (gemm_w128.h, batch = the 16 planes, M = output channels, N = tiles, K = input channels):
// 2.25 x fewer MFMAs than the 9-tap implicit GEMM (convcl.h), paid for with HBM traffic -- V is 4 x the input, Mo 4 x the
// output.  Three launches:
//
//   input_transform   x NCHW [B][C][H][W]  ->  V  [16][C][NT]      one thread = one channel x 4 adjacent tiles: reads the
//                     4 x 10 input patch (two 16-byte loads + two 4-byte loads per row), writes 16 x 16 bytes, lanes along
//                     the tiles: every store instruction is 1 KB contiguous.  The data gradient's ReLU mask (gy * (y > 0))
//                     rides on the loads.
//   gemm_w128         U [16][C][Mp] x V -> Mo [16][Mp][NT]
//   output_transform  Mo -> y NCHW (+ bias, ReLU, or += for a gradient accumulated in place): one thread = one output
//                     channel x 4 adjacent tiles: 16 x 16-byte loads, two rows of 8 pixels = 4 x 16-byte stores.
//
// Tiles: B * TH * TWp of them, TH = ceil(H / 2), TWp = ceil(W / 2) rounded up to a multiple of 4 (a thread's 4 tiles share a
// tile row), NT = that count rounded up to a multiple of 16 (the weight gradient contracts over the tiles in steps of 16);
// the padding tiles hold zeros and are never written back.  The data gradient is the same three launches on the output
// gradient with the weights transposed and the taps reversed (weight_transform with `flip`).
//
// Weight gradient, also in the transform domain (the adjoint of the above with respect to U):
//
//     gU[t][c][n] = sum_tile V[t][c][tile] * gM[t][n][tile],   gM = A gy A^T per tile,      gw[n][c] = G^T gU[.][c][n] G
//
//   input_transform   x  -> V   (as in the forward)
//   grad_transform    gy -> gM  [16][N][NT]   (2 x 2 -> 4 x 4 per tile; the ReLU mask rides on the loads)
//   wrw_planes        both operands have the contraction index (the tiles) contiguous: the k-contiguous contraction of
//                     convcl.h (64-byte LDS rows, direct-to-LDS loads), one workgroup per (plane, K split, tile), parts
//                     [S][16][rows][cols] written, no atomics
//   wrw_reduce        sum over the K splits in a fixed order, G^T . G, gw (= | +=)
//
// Numerics: not the direct form's summation order.  For |x|, |w| ~ 1 and C = 256 the difference to the fp64 convolution
// is of the size of the direct fp32 form's own (tests/test_winograd_gpu.py states the bound); the transforms use only
// additions, subtractions and a multiplication by 0.5 (exact), the contraction is the k-ascending fmaf chain of gemm_w128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wino {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Geometry {
    int B, H, W;             // image
    int TH, TWp;             // tile rows, padded tile columns (multiple of 4)
    int tiles;               // B * TH * TWp
    int NT;                  // tiles rounded up to a multiple of 16: the row length of V / Mo / gM
};

__host__ __device__ inline Geometry make_geometry(int B, int H, int W) {
    Geometry g;
    g.B = B; g.H = H; g.W = W;
    g.TH = (H + 1) / 2;
    g.TWp = ((W + 1) / 2 + 3) & ~3;
    g.tiles = B * g.TH * g.TWp;
    g.NT = (g.tiles + 15) & ~15;
    return g;
}

// ---- weights: U[t][k][m] = (G g G^T)[t / 4][t % 4],  g = w[m][k] (forward: k = input channel, m = output channel) or
// g = w[k][m] rotated by 180 degrees (data gradient: k = output channel, m = input channel).  U is [16][Kp][Mp], Kp >= K,
// Mp >= M: zeros beyond K / M.  w is [Cout][Cin][3][3].
__global__ void weight_transform_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Kp, int Mp, int flip) {
    const int K = flip ? Cout : Cin, M = flip ? Cin : Cout;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kp * Mp) return;
    const int k = i / Mp, m = i - k * Mp;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.f;
            if (m < M && k < K) v = flip ? w[((size_t)k * Cin + m) * 9 + (2 - a) * 3 + (2 - b)] : w[((size_t)m * Cin + k) * 9 + a * 3 + b];
            g[a][b] = v;
        }
    float t[4][3];      // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
        const size_t plane = (size_t)Kp * Mp;
        U[(size_t)(4 * a + 0) * plane + i] = u0;
        U[(size_t)(4 * a + 1) * plane + i] = u1;
        U[(size_t)(4 * a + 2) * plane + i] = u2;
        U[(size_t)(4 * a + 3) * plane + i] = u3;
    }
}

// ---- input: V[t][c][tile] = (B^T d B)[t / 4][t % 4], d = the 4 x 4 patch of x around output tile `tile`
// x: channel c of image b at x + b * sxb + c * sxc (H x W, contiguous rows).  mask (optional): the activation bits an
// output transform left behind -- [B][C][H][W8] bytes, W8 = ceil(W / 8), bit j of byte s = pixel 8 s + j "passes the
// gradient" -- x reads as zero where its bit is clear (the ReLU adjoint at 1/32 of the bytes of re-reading the output).
// One thread: 4 tiles (b, ty, 4 tq .. 4 tq + 3).
// VEC: W % 4 == 0 and 16-byte aligned rows -> 16-byte loads.  Output layout, `rows` >= C channel rows (zeros for c >= C):
//   CHUNKED = false: V [16][rows][NT], thread = (channel blockIdx.y, quad): lanes along the tiles, 1 KB per store instruction;
//   CHUNKED = true:  V [16][NT / 16][rows][16] -- the 16 tiles of a K chunk of the weight gradient's contraction contiguous per
//                    row, consecutive rows 64 bytes apart (what its direct-to-LDS loads fetch 1 KB at a time); thread =
//                    (channel 64 blockIdx.y + tid / 4, quad 4 blockIdx.x + tid % 4): a wave stores 16 rows x 64 bytes = 1 KB.
template <bool CHUNKED>
struct Slot {
    int row, quad;          // channel row, quad of tiles
    __device__ __forceinline__ Slot() {
        if (CHUNKED) { row = 64 * blockIdx.y + (threadIdx.x >> 2); quad = 4 * blockIdx.x + (threadIdx.x & 3); }
        else { row = blockIdx.y; quad = blockIdx.x * blockDim.x + threadIdx.x; }
    }
    // first of the thread's 4 floats in plane t
    __device__ __forceinline__ size_t at(int t, int rows, int NT) const {
        if (CHUNKED) return (((size_t)t * (NT >> 4) + (quad >> 2)) * rows + row) * 16 + 4 * (quad & 3);
        return ((size_t)t * rows + row) * NT + 4 * (size_t)quad;
    }
};

template <bool VEC, bool CHUNKED = false>
__global__ __launch_bounds__(256) void input_transform_kernel(const float* __restrict__ x, int64_t sxb, int64_t sxc,
                                                             const unsigned char* __restrict__ mask,
                                                             float* __restrict__ V, int C, int rows, Geometry g) {
    const Slot<CHUNKED> slot;
    const int i = slot.quad, c = slot.row;
    if (i >= (g.NT >> 2) || c >= rows) return;
    if (c >= C || 4 * i >= g.tiles) {                 // padding channel of the contraction / padding tiles
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(V + slot.at(t, rows, g.NT)) = z;
        return;
    }
    const int qpr = g.TWp >> 2;                       // quads per tile row
    const int tq = i % qpr, ty = (i / qpr) % g.TH, b = i / (qpr * g.TH);
    const float* xc = x + b * sxb + c * sxc;
    const int W8 = (g.W + 7) >> 3;
    const unsigned char* mc = mask ? mask + ((size_t)b * C + c) * g.H * W8 : nullptr;
    const int col0 = 8 * tq;                          // first output column of the quad; the patch spans col0 - 1 .. col0 + 8
    float d[4][10];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = 2 * ty - 1 + rr;
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
            // bits of columns col0 - 1 .. col0 + 8: the top bit of the byte to the left, this quad's byte, bit 0 of the next
            const unsigned char* mrow = mc + (size_t)row * W8;
            const unsigned left = tq > 0 ? mrow[tq - 1] : 0u, mid = tq < W8 ? mrow[tq] : 0u, right = tq + 1 < W8 ? mrow[tq + 1] : 0u;
            const unsigned bits = (left >> 7) | (mid << 1) | ((right & 1u) << 9);
#pragma unroll
            for (int e = 0; e < 10; ++e)
                if (!((bits >> e) & 1u)) d[rr][e] = 0.f;
        }
    }
    // rows: B^T d  (t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3), then the same along the columns per tile
    float rws[4][10];
#pragma unroll
    for (int e = 0; e < 10; ++e) {
        rws[0][e] = d[0][e] - d[2][e];
        rws[1][e] = d[1][e] + d[2][e];
        rws[2][e] = d[2][e] - d[1][e];
        rws[3][e] = d[1][e] - d[3][e];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        f32x4 v0, v1, v2, v3;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const float e0 = rws[a][2 * tl], e1 = rws[a][2 * tl + 1], e2 = rws[a][2 * tl + 2], e3 = rws[a][2 * tl + 3];
            v0[tl] = e0 - e2; v1[tl] = e1 + e2; v2[tl] = e2 - e1; v3[tl] = e1 - e3;
        }
        *reinterpret_cast<f32x4*>(V + slot.at(4 * a + 0, rows, g.NT)) = v0;
        *reinterpret_cast<f32x4*>(V + slot.at(4 * a + 1, rows, g.NT)) = v1;
        *reinterpret_cast<f32x4*>(V + slot.at(4 * a + 2, rows, g.NT)) = v2;
        *reinterpret_cast<f32x4*>(V + slot.at(4 * a + 3, rows, g.NT)) = v3;
    }
}

// ---- output: y[b][n][2 ty + i][2 tx + j] = (A^T m A)[i][j] (+ bias[n]) (ReLU) (+= y), m = Mo[.][n][tile] as 4 x 4
// Mo [16][Mp][NT]; y: channel n of image b at y + b * syb + n * syc.  One thread: channel n, 4 tiles.
// bits (optional, with an activation): [B][N][H][W8] bytes, bit j of byte s = output pixel 8 s + j passes the gradient
// (pre-activation > 0; OUT_RELU_FINITE: and finite) -- the mask the adjoint's transforms read.
// OUT_RELU_FINITE: ReLU, then torch.nan_to_num (models/raft_core.py:163-164): NaN -> 0 (fmaxf drops it), +inf -> FLT_MAX
enum { OUT_PLAIN = 0, OUT_RELU = 1, OUT_RELU_FINITE = 2 };
template <bool VEC>
__global__ __launch_bounds__(256) void output_transform_kernel(const float* __restrict__ Mo, int Mp, const float* __restrict__ bias,
                                                              float* __restrict__ y, int64_t syb, int64_t syc, int act,
                                                              int accumulate, unsigned char* __restrict__ bits, int N, Geometry g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (4 * i >= g.tiles) return;
    const int qpr = g.TWp >> 2;
    const int tq = i % qpr, ty = (i / qpr) % g.TH, b = i / (qpr * g.TH);
    const size_t plane = (size_t)Mp * g.NT;
    const float* in = Mo + (size_t)n * g.NT + 4 * (size_t)i;
    f32x4 m[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) m[t] = *reinterpret_cast<const f32x4*>(in + (size_t)t * plane);
    const float bv = bias ? bias[n] : 0.f;
    float* yc = y + b * syb + n * syc;
    const int col0 = 8 * tq;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const int row = 2 * ty + ri;
        // A^T m: row 0 = m0 + m1 + m2, row 1 = m1 - m2 - m3 (along the first index), then the same along the second
        f32x4 s[4];
#pragma unroll
        for (int cidx = 0; cidx < 4; ++cidx)
            s[cidx] = ri == 0 ? m[cidx] + m[4 + cidx] + m[8 + cidx] : m[4 + cidx] - m[8 + cidx] - m[12 + cidx];
        const f32x4 o0 = s[0] + s[1] + s[2] + bv, o1 = s[1] - s[2] - s[3] + bv;       // columns 2 tx, 2 tx + 1 of the 4 tiles
        float px[8] = {o0[0], o1[0], o0[1], o1[1], o0[2], o1[2], o0[3], o1[3]};
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

// ---- output gradient: gM[t][n][tile] = (A gy A^T)[t / 4][t % 4] (A = the 4 x 2 matrix whose transpose finishes the forward)
// gy: channel n of image b at gy + b * sgb + n * sgc; mask optional: the activation bits [B][N][H][W8] of the forward's
// output transform, gy reads as zero where its bit is clear.
// gM has `rows` >= N channel rows (zeros for n >= N and for the padding tiles) in either layout of input_transform_kernel.
template <bool VEC, bool CHUNKED = false>
__global__ __launch_bounds__(256) void grad_transform_kernel(const float* __restrict__ gy, int64_t sgb, int64_t sgc,
                                                            const unsigned char* __restrict__ mask,
                                                            float* __restrict__ gM, int N, int rows, Geometry g) {
    const Slot<CHUNKED> slot;
    const int i = slot.quad, n = slot.row;
    if (i >= (g.NT >> 2) || n >= rows) return;
    if (n >= N || 4 * i >= g.tiles) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(gM + slot.at(t, rows, g.NT)) = z;
        return;
    }
    const int qpr = g.TWp >> 2;
    const int tq = i % qpr, ty = (i / qpr) % g.TH, b = i / (qpr * g.TH);
    const float* gc = gy + b * sgb + n * sgc;
    const int W8 = (g.W + 7) >> 3;
    const unsigned char* mc = mask ? mask + ((size_t)b * N + n) * g.H * W8 : nullptr;
    const int col0 = 8 * tq;
    float d[2][8];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * ty + rr;
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
    // A g: rows g0, g0 + g1, g0 - g1, -g1; then the same along the columns of each tile
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float rw[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rw[e] = a == 0 ? d[0][e] : a == 1 ? d[0][e] + d[1][e] : a == 2 ? d[0][e] - d[1][e] : -d[1][e];
        f32x4 v0, v1, v2, v3;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const float e0 = rw[2 * tl], e1 = rw[2 * tl + 1];
            v0[tl] = e0; v1[tl] = e0 + e1; v2[tl] = e0 - e1; v3[tl] = -e1;
        }
        *reinterpret_cast<f32x4*>(gM + slot.at(4 * a + 0, rows, g.NT)) = v0;
        *reinterpret_cast<f32x4*>(gM + slot.at(4 * a + 1, rows, g.NT)) = v1;
        *reinterpret_cast<f32x4*>(gM + slot.at(4 * a + 2, rows, g.NT)) = v2;
        *reinterpret_cast<f32x4*>(gM + slot.at(4 * a + 3, rows, g.NT)) = v3;
    }
}

// ---- gbias[n] (= | +=) sum over every pixel of the (masked) output gradient = the row sum of plane 5 of gM: (A g A^T)[1][1]
// is g00 + g01 + g10 + g11 of the tile.  gM chunk-major [16][NT / 16][rows][16].  One block per channel, fixed summation tree.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ gM, int rows, int NT, float* __restrict__ gbias, int accumulate) {
    __shared__ float part[256];
    const int n = blockIdx.x, chunks = NT >> 4;
    const float* q = gM + (size_t)5 * rows * NT + (size_t)n * 16;
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

// ---- gw[n][c][i][j] (= | +=) sum_{a,b} G[a][i] G[b][j] * sum_s part[s][4 a + b][.]: element (c, n) of a part at
// c * sc + n * sn (the contraction ran with either operand on its row side).  Block = 64 n x 4 plane groups: a thread sums
// the S parts of its 4 planes (independent loads, lanes along n), the 16 plane sums of an n meet in LDS, the first plane
// group finishes with G^T . G.  (r6, first form: one thread per (c, n) walking all 16 S loads in turn -- 69 us for 47 MB,
// a latency chain on 768 waves.)
__global__ __launch_bounds__(256) void wrw_reduce_kernel(const float* __restrict__ parts, int S, int64_t part_floats, int64_t sc, int64_t sn,
                                                        float* __restrict__ gw, int C, int N, int accumulate) {
    __shared__ float us[16][64];
    const int nl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl, c = blockIdx.y;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const float* q = parts + (size_t)(4 * tg) * part_floats + c * sc + n * sn;
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += q[((size_t)s * 16 + k) * part_floats];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) us[4 * tg + k][nl] = acc[k];
    __syncthreads();
    if (tg != 0 || n >= N) return;
    float u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) u[t] = us[t][nl];
    // G^T u G, G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
    float r[3][4];      // rows: G^T applied to the first index
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        r[0][b] = u[b] + 0.5f * (u[4 + b] + u[8 + b]);
        r[1][b] = 0.5f * (u[4 + b] - u[8 + b]);
        r[2][b] = 0.5f * (u[4 + b] + u[8 + b]) + u[12 + b];
    }
    float* o = gw + ((size_t)n * C + c) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float o0 = r[a][0] + 0.5f * (r[a][1] + r[a][2]), o1 = 0.5f * (r[a][1] - r[a][2]), o2 = 0.5f * (r[a][1] + r[a][2]) + r[a][3];
        if (accumulate) { o[3 * a] += o0; o[3 * a + 1] += o1; o[3 * a + 2] += o2; }
        else { o[3 * a] = o0; o[3 * a + 1] = o1; o[3 * a + 2] = o2; }
    }
}

}  // namespace wino

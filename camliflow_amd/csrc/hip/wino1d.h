// 1-D Winograd F(4, 5) for GRU2D's separable convolutions (models/raft_core.py:110-140: nn.Conv2d(hidden + input, hidden,
// (1, 5) | (5, 1), padding (0, 2) | (2, 0))), channels-last, fp32, gfx950.
//
//     y[p][n] = sum_k sum_c x[p + (k - 2) e][c] * w[n][k][c]          e = one pixel along the convolution's axis; zero outside
//
// as  Y = A^T [ sum_c (G g) . (B^T d) ]  per tile of 4 consecutive outputs along the axis: 8 multiplications per 4 outputs
// and channel pair instead of 20 (2.5 x fewer), interpolation points 0, +-1, +-2, +-1/2, inf.  The sum over c for the 8
// positions of the transform domain is 8 independent contractions with the contraction index (the channels) contiguous
// in both operands -- the k-contiguous core of convcl.h with one tap:
//
//     Mo[t][tile][n] = sum_c V[t][tile][c] * U[t][n][c]           t = 0 .. 7
//
//   input_transform_1d    x NHWC (one or two tensors = cat[h, motion])  ->  V [8][tiles][C]      one wave = one tile: lanes
//                         along the channels (4 each), 8 positions x 16 bytes in, 8 planes x 16 bytes out, all 1 KB runs
//   planes_cl_kernel      convcl_body<NTW> on (V[t], U[t]) per plane t (blockIdx.y), Mo [8][tiles][Cout]
//   output_transform_1d   Mo -> the 4 outputs of the tile with the epilogue of the convolution they replace: PLAIN (the data
//                         gradient: two output tensors, = or +=), GATES (z, r, r h), BLEND (q, h'); the arithmetic is that
//                         of convcl.h's epilogues
//
// The transform domain is 2 x the image domain either side (8 values per 4 pixels), against 4 x for the 2-D F(2x2,3x3) and
// 2.25 x for F(4x4,3x3) of winograd.h.  Numerics: the 1-D transforms carry factors up to 21/4 and 8; measured on unit-
// variance data at 256 channels: 8e-6 max abs / 1.3e-6 relative L2 against an fp64 convolution (the direct fp32 chain:
// 4e-6 / 9e-7), profiles/r06_experiments.txt item 14.
#pragma once
#include "wrwcl.h"

namespace w1d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 1-D lines of the image along the convolution's axis.  axis 0: along W (1 x 5), a line = one image row; axis 1: along H
// (5 x 1), a line = one image column.  Tile = 4 consecutive positions of a line; tiles are numbered line-major.
struct Lines {
    int len;             // positions per line (W | H)
    int tpl;             // tiles per line = ceil(len / 4)
    int nlines;          // B * H | B * W
    int tiles;           // nlines * tpl
    int step;            // pixels between consecutive positions (1 | W)
    int W, HW;           // image geometry (axis 1: line (b, x) starts at pixel b * HW + x)
    int axis;
};

__host__ __device__ inline Lines make_lines(int B, int H, int W, int axis) {
    Lines l;
    l.axis = axis; l.W = W; l.HW = H * W;
    l.len = axis == 0 ? W : H;
    l.tpl = (l.len + 3) / 4;
    l.nlines = axis == 0 ? B * H : B * W;
    l.tiles = l.nlines * l.tpl;
    l.step = axis == 0 ? 1 : W;
    return l;
}

// first pixel of line `line`
__device__ __forceinline__ int line_start(const Lines& l, int line) {
    return l.axis == 0 ? line * l.W : (line / l.W) * l.HW + line % l.W;
}

// B^T d: 8 -> 8
__device__ __forceinline__ void bt8(const f32x4 (&d)[8], f32x4 (&t)[8]) {
    t[0] = (d[6] - d[0]) + 5.25f * (d[2] - d[4]);
    f32x4 a = d[2] - 4.25f * d[4] + d[6], b = d[1] - 4.25f * d[3] + d[5];
    t[1] = a + b; t[2] = a - b;
    a = 0.25f * d[2] - 1.25f * d[4] + d[6]; b = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    t[3] = a + b; t[4] = a - b;
    a = 4.f * d[2] - 5.f * d[4] + d[6]; b = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    t[5] = a + b; t[6] = a - b;
    t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
}

// A^T m: 8 -> 4
__device__ __forceinline__ void at8(const f32x4 (&m)[8], f32x4 (&y)[4]) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
    y[0] = m[0] + s12 + s34 + s56;
    y[1] = d12 + 2.f * d34 + 0.5f * d56;
    y[2] = s12 + 4.f * s34 + 0.25f * s56;
    y[3] = d12 + 8.f * d34 + 0.125f * d56 + m[7];
}

// ---- weights: U[t][n][c] = sum_k G[t][k] * wp[n][flip ? 4 - k : k][c]          wp = the packed [N][5][C] weights of convcl
// (flip: the data gradient = the same convolution of the output gradient on the transposed packing with the taps reversed)
__global__ void weight_transform_1d_kernel(const float* __restrict__ wp, float* __restrict__ U, int N, int C, int flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float g[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) g[k] = wp[((size_t)n * 5 + (flip ? 4 - k : k)) * C + c];
    const float se = g[0] + g[2] + g[4], so = g[1] + g[3];
    const float e3 = (1.f / 90.f) * g[0] + (2.f / 45.f) * g[2] + (8.f / 45.f) * g[4], o3 = (1.f / 45.f) * g[1] + (4.f / 45.f) * g[3];
    const float e5 = (32.f / 45.f) * g[0] + (8.f / 45.f) * g[2] + (2.f / 45.f) * g[4], o5 = (16.f / 45.f) * g[1] + (4.f / 45.f) * g[3];
    const size_t plane = (size_t)N * C;
    U[i] = -g[0];
    U[plane + i] = (-2.f / 9.f) * (se + so);
    U[2 * plane + i] = (-2.f / 9.f) * (se - so);
    U[3 * plane + i] = e3 + o3;
    U[4 * plane + i] = e3 - o3;
    U[5 * plane + i] = e5 + o5;
    U[6 * plane + i] = e5 - o5;
    U[7 * plane + i] = g[4];
}

// ---- input: V[t][tile][c] = (B^T d)[t], d[j] = x[pos0 - 2 + j] along the line (zero outside it)
// x = cat[x0 (C0 channels, ldx0 floats per pixel), x1 (C1 channels)]; block 256 = 4 waves, 64 lanes x 4 channels = 256
// channels per pass; one tile per wave and pass
// V has `rows` >= tiles rows per plane (the weight gradient pads the tile count so that its K splits do not straddle planes):
// rows beyond the tiles are written as zeros.
__global__ __launch_bounds__(256) void input_transform_1d_kernel(const float* __restrict__ x0, int ldx0, int C0, const float* __restrict__ x1,
                                                                int ldx1, int C1, float* __restrict__ V, int rows, Lines l) {
    const int C = C0 + C1;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= rows) return;
    if (tile >= l.tiles) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int c = 4 * lane; c < C; c += 256)
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(V + ((size_t)j * rows + tile) * C + c) = z;
        return;
    }
    const int line = tile / l.tpl, pos0 = 4 * (tile - line * l.tpl);
    const int p0 = line_start(l, line);
    for (int c = 4 * lane; c < C; c += 256) {
        const bool second = c >= C0;
        const float* src = second ? x1 + (c - C0) : x0 + c;
        const int ld = second ? ldx1 : ldx0;
        f32x4 d[8], t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = pos0 - 2 + j;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            d[j] = (unsigned)pos < (unsigned)l.len ? *reinterpret_cast<const f32x4*>(src + (size_t)(p0 + pos * l.step) * ld) : z;
        }
        bt8(d, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(V + ((size_t)j * rows + tile) * C + c) = t[j];
    }
}

// ---- the 8 plane contractions: convcl_body on one slice of operands per plane
struct PlanesBatch {
    ccl::Problem base;            // x = V, w = U, y = Mo of plane 0; B = H = 1, W = tiles; T = 1
    int64_t x_plane, w_plane, y_plane;
};

template <int NTW, int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void planes_cl_kernel(PlanesBatch pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ccl::Problem& p = pb.base;
    const int t = blockIdx.y;
    float* y = p.y + (int64_t)t * pb.y_plane;
    const float* x = p.x + (int64_t)t * pb.x_plane;
    const ccl::Operands o = {x, x, p.w + (int64_t)t * pb.w_plane, y, y, p.Cin, p.C0};
    ccl::convcl_body<NTW, NBUF, ccl::EPI_PLAIN>(p, o, lds);
}

// ---- output: the tile's 4 outputs = A^T Mo[.][tile][n], then the epilogue of the convolution this replaces (convcl.h)
struct Epilogue {
    int N, N0;                    // output channels; PLAIN: channels [0, N0) -> y, [N0, N) -> y1
    float* y; float* y1; float* y2;
    int ldy, ldy1, ldy2;
    int acc0, acc1;               // PLAIN: += instead of =
    const float* add; const float* h; const float* z;
    int ld_add, ld_h, ld_z;
    int sanitize;
};

template <int EPI>
__global__ __launch_bounds__(256) void output_transform_1d_kernel(const float* __restrict__ Mo, Epilogue e, Lines l) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= l.tiles) return;
    const int line = tile / l.tpl, pos0 = 4 * (tile - line * l.tpl);
    const int p0 = line_start(l, line);
    const size_t plane = (size_t)l.tiles * e.N;
    for (int n = 4 * lane; n < e.N; n += 256) {
        f32x4 m[8], v[4];
#pragma unroll
        for (int t = 0; t < 8; ++t) m[t] = *reinterpret_cast<const f32x4*>(Mo + (size_t)t * plane + (size_t)tile * e.N + n);
        at8(m, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pos0 + i >= l.len) break;
            const size_t pix = (size_t)(p0 + (pos0 + i) * l.step);
            if (EPI == ccl::EPI_PLAIN) {
                const bool second = n >= e.N0;
                float* dst = second ? e.y1 + pix * e.ldy1 + (n - e.N0) : e.y + pix * e.ldy + n;
                f32x4 o = v[i];
                if (second ? e.acc1 : e.acc0) o += *reinterpret_cast<const f32x4*>(dst);
                *reinterpret_cast<f32x4*>(dst) = o;
            } else if (EPI == ccl::EPI_GATES) {
                const f32x4 pre = v[i] + *reinterpret_cast<const f32x4*>(e.add + pix * e.ld_add + n);
                f32x4 g;
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = ccl::sigmoid_(pre[q]);
                if (n < 128) {
                    *reinterpret_cast<f32x4*>(e.y + pix * e.ldy + n) = g;                     // z
                } else {
                    const f32x4 hh = *reinterpret_cast<const f32x4*>(e.h + pix * e.ld_h + (n - 128));
                    *reinterpret_cast<f32x4*>(e.y2 + pix * e.ldy2 + (n - 128)) = g;           // r
                    *reinterpret_cast<f32x4*>(e.y1 + pix * e.ldy1 + (n - 128)) = g * hh;      // r h
                }
            } else {
                const f32x4 pre = v[i] + *reinterpret_cast<const f32x4*>(e.add + pix * e.ld_add + n);
                const f32x4 zq = *reinterpret_cast<const f32x4*>(e.z + pix * e.ld_z + n), hq = *reinterpret_cast<const f32x4*>(e.h + pix * e.ld_h + n);
                f32x4 qv, hn;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    qv[q] = tanhf(pre[q]);
                    hn[q] = (1.0f - zq[q]) * hq[q] + zq[q] * qv[q];
                    if (e.sanitize) hn[q] = ccl::nan_to_num_(hn[q]);
                }
                *reinterpret_cast<f32x4*>(e.y1 + pix * e.ldy1 + n) = qv;
                *reinterpret_cast<f32x4*>(e.y + pix * e.ldy + n) = hn;
            }
        }
    }
}

// ---- weight gradient in the transform domain:  gU[t][c][n] = sum_tile V[t][tile][c] * gM[t][tile][n],  gM = A g per tile (A = the
// transpose of at8's matrix), then gw[n][c][k] (= | +=) sum_t G[t][k] gU[t][c][n].  The contraction runs over the tiles with both
// operands tile-major and channel-contiguous -- the layout wrwcl.h contracts over pixels -- so it IS wrw::wrw_kernel with one tap
// on the 8 planes laid end to end as 8 * rows "pixels": `rows` = the tile count padded to a multiple of the K split, so that no
// split straddles two planes; part s = plane s / Sp, split s % Sp.
__device__ __forceinline__ void ag8(const f32x4 (&g)[4], f32x4 (&t)[8]) {
    const f32x4 e = g[0] + g[2], o = g[1] + g[3];
    t[0] = g[0];
    t[1] = e + o;
    t[2] = e - o;
    const f32x4 e2 = g[0] + 4.f * g[2], o2 = 2.f * g[1] + 8.f * g[3];
    t[3] = e2 + o2;
    t[4] = e2 - o2;
    const f32x4 e3 = g[0] + 0.25f * g[2], o3 = 0.5f * g[1] + 0.125f * g[3];
    t[5] = e3 + o3;
    t[6] = e3 - o3;
    t[7] = g[3];
}

// gy [P][ldg] (N channels) -> gM [8][rows][N]; one wave per tile, lanes along the channels
__global__ __launch_bounds__(256) void grad_transform_1d_kernel(const float* __restrict__ gy, int ldg, int N, float* __restrict__ gM, int rows, Lines l) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= rows) return;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (tile >= l.tiles) {
        for (int n = 4 * lane; n < N; n += 256)
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(gM + ((size_t)j * rows + tile) * N + n) = z;
        return;
    }
    const int line = tile / l.tpl, pos0 = 4 * (tile - line * l.tpl);
    const int p0 = line_start(l, line);
    for (int n = 4 * lane; n < N; n += 256) {
        f32x4 g[4], t[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            g[i] = pos0 + i < l.len ? *reinterpret_cast<const f32x4*>(gy + (size_t)(p0 + (pos0 + i) * l.step) * ldg + n) : z;
        ag8(g, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(gM + ((size_t)j * rows + tile) * N + n) = t[j];
    }
}

// gw[(n * C + c) * 5 + k] (= | +=) sum_t G[t][k] * sum_{j < Sp} part[t * Sp + j][c][n].  One block per (c, 128 output channels):
// 32 lanes x float4 along n, 8 groups of threads sharing the Sp partial sums of each plane (group g takes j = g, g + 8, ...: with
// one thread per element the 8 * Sp dependent loads of a thread were a 66 us latency chain for 67 MB), combined through LDS in a
// fixed order, then one thread per n applies G.  N a multiple of 128.
__global__ __launch_bounds__(256) void wrw_reduce_1d_kernel(const float* __restrict__ part, int Sp, float* __restrict__ gw, int C, int N, int accumulate) {
    __shared__ f32x4 sm[8][8][32];
    const int nb = N / 128;
    const int c = blockIdx.x / nb, n0 = (blockIdx.x % nb) * 128;
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t el = (size_t)C * N;
    const float* src = part + (size_t)c * N + n0 + 4 * lane;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        f32x4 a0 = z, a1 = z;
        int j = g;
        for (; j + 8 < Sp; j += 16) {
            a0 += *reinterpret_cast<const f32x4*>(src + ((size_t)t * Sp + j) * el);
            a1 += *reinterpret_cast<const f32x4*>(src + ((size_t)t * Sp + j + 8) * el);
        }
        if (j < Sp) a0 += *reinterpret_cast<const f32x4*>(src + ((size_t)t * Sp + j) * el);
        sm[g][t][lane] = a0 + a1;
    }
    __syncthreads();
    if (threadIdx.x >= 128) return;
    const int n = threadIdx.x;
    float u[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += reinterpret_cast<const float*>(&sm[q][t][0])[n];
        u[t] = acc;
    }
    const float s12 = u[1] + u[2], d12 = u[1] - u[2], s34 = u[3] + u[4], d34 = u[3] - u[4], s56 = u[5] + u[6], d56 = u[5] - u[6];
    float o[5];
    o[0] = -u[0] - (2.f / 9.f) * s12 + (1.f / 90.f) * s34 + (32.f / 45.f) * s56;
    o[1] = -(2.f / 9.f) * d12 + (1.f / 45.f) * d34 + (16.f / 45.f) * d56;
    o[2] = -(2.f / 9.f) * s12 + (2.f / 45.f) * s34 + (8.f / 45.f) * s56;
    o[3] = -(2.f / 9.f) * d12 + (4.f / 45.f) * d34 + (4.f / 45.f) * d56;
    o[4] = -(2.f / 9.f) * s12 + (8.f / 45.f) * s34 + (2.f / 45.f) * s56 + u[7];
    float* dst = gw + ((size_t)(n0 + n) * C + c) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) dst[k] = accumulate ? dst[k] + o[k] : o[k];
}

}  // namespace w1d

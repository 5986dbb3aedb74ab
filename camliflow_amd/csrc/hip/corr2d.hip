// PWC-style local correlation (cost volume over a (2*md+1)^2 displacement window), gfx950.
//
// Replaces models/csrc/correlation/correlation_forward_kernel.cu:11-55 and
// correlation_backward_kernel.cu:4-89 of the reference.  Layouts are those of the reference's
// native symbols (correlation.cpp:11-35): inputs NHWC, cost volume NCHW; the backward writes
// NHWC gradients directly (the reference writes NCHW and wrapper.py:34-35 permutes + copies).
//
//   out[n, (dy+md)*Dd + (dx+md), y, x] = (1/C) * sum_c in1[n,y,x,c] * in2[n,y+dy,x+dx,c]
//
// HBM-bound op (9-17 flop/B).  Forward: a workgroup owns 64 consecutive pixels of one image row;
// the in1 tile and, per dy, the in2 row tile (+-md halo) are staged into LDS with coalesced
// channel-contiguous loads (row stride C+1 dwords -> conflict-free column reads); the four waves
// split the channel range, keep the 2*md+1 dx-accumulators in registers and combine through LDS,
// so each cost-volume row is written as 64 contiguous floats.  in1/in2 are read from HBM once
// per (row, dy) instead of 81 x per pixel.  Backward: lanes run along the contiguous channel
// axis (coalesced), the grad_output scalar is wave-uniform per pixel.
#include "camli_common.h"

#include <stdlib.h>

namespace {

// CAMLI_CORR2D_TILE=0 selects the one-row forward kernel (A/B measurements); default: the 2-D tile kernel
bool camli_corr2d_use_tile() {
    static const bool on = [] { const char* e = getenv("CAMLI_CORR2D_TILE"); return !(e && e[0] == '0'); }();
    return on;
}

constexpr int CT_PX = 64;     // pixels per workgroup (one wave-width)
constexpr int CT_CMAX = 256;  // in1 channels resident in LDS per pass
constexpr int CT_CH = 64;     // in2 channels staged per chunk
#ifndef CAMLI_CORR2D_SPLIT_BELOW
#define CAMLI_CORR2D_SPLIT_BELOW 2048
#endif

// grid (ceil(W/64), H, B), block 256.  VEC: C % 4 == 0 -> 16-byte global loads, ds_read_b128 in the
// inner loop (row stride = channels + 4 dwords keeps the 16-lane b128 phases conflict-free).
// LDS: in1 tile [64][cs+PAD] stays for all dy; per dy the in2 row tile is streamed in CT_CH-channel
// chunks [64+2md][CT_CH+PAD]; red [4][Dd][64] combines the four waves' channel quarters.
template <int MD, bool VEC>
__global__ __launch_bounds__(256) void corr2d_fwd_kernel(const float* __restrict__ in1,
                                                          const float* __restrict__ in2,
                                                          float* __restrict__ out, int C, int H, int W, int zsplit) {
    constexpr int DD = 2 * MD + 1;
    constexpr int HALO = CT_PX + 2 * MD;
    constexpr int PAD = VEC ? 4 : 1;
    constexpr int LD2 = CT_CH + PAD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int px = tid & 63;
    const int cq = tid >> 6;
    // zsplit > 1 (small pyramid levels, round 5): the displacement rows dy are dealt to zsplit workgroups per (row, image) --
    // a 9 x 15 level is 9 workgroups walking 9 dy x 3 channel chunks of stage -> barrier -> accumulate one after the other
    // (55 us for 2 MFLOP); the per-element arithmetic and its order are unchanged
    const int x0 = blockIdx.x * CT_PX, y = blockIdx.y, n = blockIdx.z / zsplit, part = blockIdx.z - n * zsplit;
    const float inv_c = 1.0f / (float)C;
    const int cs_max = min(C, CT_CMAX);
    const int ld1 = cs_max + PAD;
    float* s1 = smem;                      // [CT_PX][ld1]
    float* s2 = s1 + CT_PX * ld1;          // [HALO][LD2]
    float* red = s2 + HALO * LD2;          // [4][DD][CT_PX]

    for (int c0 = 0; c0 < C; c0 += CT_CMAX) {
        const int cs = min(CT_CMAX, C - c0);
        // ---- in1 tile: pixels x0..x0+63 of row y, channels c0..c0+cs ----
        if (VEC) {
            const int q4 = cs >> 2;
            for (int e = tid; e < CT_PX * q4; e += 256) {
                const int p = e / q4, c = (e - p * q4) << 2;
                const int x = x0 + p;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (x < W) v = *reinterpret_cast<const float4*>(in1 + (((size_t)n * H + y) * W + x) * C + c0 + c);
                *reinterpret_cast<float4*>(s1 + p * ld1 + c) = v;
            }
        } else {
            for (int e = tid; e < CT_PX * cs; e += 256) {
                const int p = e / cs, c = e - p * cs;
                const int x = x0 + p;
                s1[p * ld1 + c] = (x < W) ? in1[(((size_t)n * H + y) * W + x) * C + c0 + c] : 0.0f;
            }
        }

        for (int dyi = part; dyi < DD; dyi += zsplit) {
            const int y2 = y + dyi - MD;
            if (y2 < 0 || y2 >= H) {   // whole displacement row is outside: zeros (block-uniform branch)
                if (c0 == 0)
                    for (int e = tid; e < DD * CT_PX; e += 256) {
                        const int d = e >> 6, x = x0 + (e & 63);
                        if (x < W) out[(((size_t)n * DD * DD + dyi * DD + d) * H + y) * W + x] = 0.0f;
                    }
                continue;
            }
            float acc[DD];
#pragma unroll
            for (int d = 0; d < DD; ++d) acc[d] = 0.0f;
            for (int k0 = 0; k0 < cs; k0 += CT_CH) {
                const int ck = min(CT_CH, cs - k0);
                __syncthreads();   // consumers of the previous s2 chunk / red are done; s1 is staged
                if (VEC) {
                    const int q4 = ck >> 2;
                    for (int e = tid; e < HALO * q4; e += 256) {
                        const int p = e / q4, c = (e - p * q4) << 2;
                        const int x = x0 - MD + p;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (x >= 0 && x < W)
                            v = *reinterpret_cast<const float4*>(in2 + (((size_t)n * H + y2) * W + x) * C + c0 + k0 + c);
                        *reinterpret_cast<float4*>(s2 + p * LD2 + c) = v;
                    }
                } else {
                    for (int e = tid; e < HALO * ck; e += 256) {
                        const int p = e / ck, c = e - p * ck;
                        const int x = x0 - MD + p;
                        s2[p * LD2 + c] =
                            (x >= 0 && x < W) ? in2[(((size_t)n * H + y2) * W + x) * C + c0 + k0 + c] : 0.0f;
                    }
                }
                __syncthreads();
                if (VEC) {
                    const int q4 = ck >> 2;
                    const int qb = (q4 * cq) / 4, qe = (q4 * (cq + 1)) / 4;
                    for (int q = qb; q < qe; ++q) {
                        const float4 a = *reinterpret_cast<const float4*>(s1 + px * ld1 + k0 + 4 * q);
#pragma unroll
                        for (int d = 0; d < DD; ++d) {
                            const float4 b = *reinterpret_cast<const float4*>(s2 + (px + d) * LD2 + 4 * q);
                            float t = __builtin_fmaf(a.x, b.x, acc[d]);
                            t = __builtin_fmaf(a.y, b.y, t);
                            t = __builtin_fmaf(a.z, b.z, t);
                            acc[d] = __builtin_fmaf(a.w, b.w, t);
                        }
                    }
                } else {
                    const int cb = (ck * cq) / 4, ce = (ck * (cq + 1)) / 4;
                    for (int c = cb; c < ce; ++c) {
                        const float a = s1[px * ld1 + k0 + c];
#pragma unroll
                        for (int d = 0; d < DD; ++d) acc[d] = __builtin_fmaf(a, s2[(px + d) * LD2 + c], acc[d]);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < DD; ++d) red[(cq * DD + d) * CT_PX + px] = acc[d];
            __syncthreads();
            for (int e = tid; e < DD * CT_PX; e += 256) {
                const int d = e >> 6, p = e & 63;
                const int x = x0 + p;
                if (x < W) {
                    float s = (red[(0 * DD + d) * CT_PX + p] + red[(1 * DD + d) * CT_PX + p]) +
                              (red[(2 * DD + d) * CT_PX + p] + red[(3 * DD + d) * CT_PX + p]);
                    const size_t o = (((size_t)n * DD * DD + dyi * DD + d) * H + y) * W + x;
                    s *= inv_c;
                    out[o] = (c0 == 0) ? s : out[o] + s;
                }
            }
        }
        __syncthreads();
    }
}

// Generic-md forward (any md): one thread per output element, used only for md values without a
// specialisation.
__global__ void corr2d_fwd_generic_kernel(const float* __restrict__ in1, const float* __restrict__ in2,
                                          float* __restrict__ out, int B, int C, int H, int W, int md) {
    const int Dd = 2 * md + 1;
    const size_t total = (size_t)B * Dd * Dd * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(e % W);
        int y = (int)((e / W) % H);
        int tc = (int)((e / ((size_t)W * H)) % (Dd * Dd));
        int n = (int)(e / ((size_t)W * H * Dd * Dd));
        int y2 = y + tc / Dd - md, x2 = x + tc % Dd - md;
        float s = 0.0f;
        if (x2 >= 0 && y2 >= 0 && x2 < W && y2 < H) {
            const float* a = in1 + (((size_t)n * H + y) * W + x) * C;
            const float* b = in2 + (((size_t)n * H + y2) * W + x2) * C;
            for (int c = 0; c < C; ++c) s = __builtin_fmaf(a[c], b[c], s);
            s = s / (float)C;
        }
        out[e] = s;
    }
}

// Backward.  which = blockIdx.z & 1: 0 -> grad wrt in1, 1 -> grad wrt in2.
//   g1[n,y,x,c]   = (1/C) sum_{dy,dx} gout[n,tc,y,x]       * in2[n,y+dy,x+dx,c]
//   g2[n,y2,x2,c] = (1/C) sum_{dy,dx} gout[n,tc,y2-dy,x2-dx] * in1[n,y2-dy,x2-dx,c]
// thread = (pixel, channel); channel is the fastest thread axis so loads/stores are coalesced.
__global__ __launch_bounds__(256) void corr2d_bwd_kernel(const float* __restrict__ gout,
                                                          const float* __restrict__ in1,
                                                          const float* __restrict__ in2,
                                                          float* __restrict__ g1, float* __restrict__ g2, int B,
                                                          int C, int H, int W, int md) {
    const int Dd = 2 * md + 1;
    const int which = blockIdx.y;
    const float* __restrict__ other = which == 0 ? in2 : in1;
    float* __restrict__ gdst = which == 0 ? g1 : g2;
    const float inv_c = 1.0f / (float)C;
    const size_t total = (size_t)B * H * W * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t pix = e / C;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((size_t)W * H));
        float s = 0.0f;
        for (int dyi = 0; dyi < Dd; ++dyi) {
            const int dy = dyi - md;
            const int yo = which == 0 ? y + dy : y - dy;  // row of the other tensor / of gout's pixel
            if (yo < 0 || yo >= H) continue;
            for (int dxi = 0; dxi < Dd; ++dxi) {
                const int dx = dxi - md;
                const int xo = which == 0 ? x + dx : x - dx;
                if (xo < 0 || xo >= W) continue;
                const int tc = dyi * Dd + dxi;
                // which==0: gout at (y,x), other=in2 at (yo,xo); which==1: gout at (yo,xo), other=in1 at (yo,xo)
                const int gy = which == 0 ? y : yo, gx = which == 0 ? x : xo;
                const float g = gout[(((size_t)n * Dd * Dd + tc) * H + gy) * W + gx];
                const float v = other[(((size_t)n * H + yo) * W + xo) * C + c];
                s = __builtin_fmaf(g, v, s);
            }
        }
        gdst[e] = s * inv_c;
    }
}

// Tiled backward for md <= 4.  WHICH = 0: grad wrt in1 (other = in2), 1: grad wrt in2 (other = in1).
// With window position (r,q), other-pixel (y+r-md, x+q-md):
//   WHICH 0: weight = gout[r*Dd+q][y][x]                      (gout at the output pixel)
//   WHICH 1: weight = gout[Dd*Dd-1-(r*Dd+q)][y+r-md][x+q-md]  (gout at the other pixel)
// grid (ceil(W/16), H, B), block = 64 * min(4, ceil(C/64)).  A wave owns 16 consecutive pixels x 64
// channels: lanes run along the channels (coalesced NHWC loads and stores), the 16 accumulators and
// the current window row (16 + 2md values) live in registers.  Each window row is 16+2md independent
// 256-byte wave loads straight from L2/HBM (no staging barrier: the kernel is latency-bound, so
// occupancy and loads in flight matter more than LDS reuse); each loaded value feeds up to Dd FMAs.
// The Dd*Dd x 16 weight table gw[r*Dd+q][k] is staged once per workgroup in LDS, already shifted per
// q and zeroed where the reference skips the displacement, and read as broadcast ds_read_b128.
constexpr int CB_TW = 16;    // pixels per workgroup

template <int MD, int WHICH>
__device__ __forceinline__ void corr2d_bwd_tiled_body(const float* __restrict__ gout, const float* __restrict__ other,
                                                      float* __restrict__ gdst, int C, int H, int W, int n, float* gw) {
    constexpr int DD = 2 * MD + 1;
    constexpr int HW = CB_TW + 2 * MD;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6, nw = blockDim.x >> 6;
    const int x0 = blockIdx.x * CB_TW, y = blockIdx.y;
    const float inv_c = 1.0f / (float)C;
    const size_t plane = (size_t)H * W;
    const float* __restrict__ gn = gout + (size_t)n * DD * DD * plane;

    for (int e = tid; e < DD * DD * CB_TW; e += blockDim.x) {
        const int t = e / CB_TW, k = e - t * CB_TW;
        const int r = t / DD, q = t - r * DD;
        const int oy = y + r - MD, ox = x0 + k + q - MD;          // the other tensor's pixel
        const bool ok = (x0 + k < W) & (oy >= 0) & (oy < H) & (ox >= 0) & (ox < W);
        const int tc = WHICH == 0 ? t : DD * DD - 1 - t;
        const int gy = WHICH == 0 ? y : oy, gx = WHICH == 0 ? x0 + k : ox;
        gw[e] = ok ? gn[(size_t)tc * plane + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();

    for (int cbase = w * 64; cbase < C; cbase += 64 * nw) {
        const bool cok = cbase + lane < C;
        float acc[CB_TW];
#pragma unroll
        for (int k = 0; k < CB_TW; ++k) acc[k] = 0.0f;
#pragma unroll 1   // one window row in registers at a time (unrolling makes the compiler hoist all Dd rows)
        for (int r = 0; r < DD; ++r) {
            const int yy = y + r - MD;
            if (yy < 0 || yy >= H) continue;   // block-uniform
            const float* __restrict__ row = other + ((size_t)n * H + yy) * W * C + cbase + lane;
            float v[HW];
#pragma unroll
            for (int i = 0; i < HW; ++i) {
                const int xx = x0 + i - MD;
                v[i] = (cok & (xx >= 0) & (xx < W)) ? row[(size_t)xx * C] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < DD; ++q) {
#pragma unroll
                for (int k4 = 0; k4 < CB_TW / 4; ++k4) {
                    const float4 g = *reinterpret_cast<const float4*>(gw + (r * DD + q) * CB_TW + 4 * k4);
                    acc[4 * k4 + 0] = __builtin_fmaf(g.x, v[q + 4 * k4 + 0], acc[4 * k4 + 0]);
                    acc[4 * k4 + 1] = __builtin_fmaf(g.y, v[q + 4 * k4 + 1], acc[4 * k4 + 1]);
                    acc[4 * k4 + 2] = __builtin_fmaf(g.z, v[q + 4 * k4 + 2], acc[4 * k4 + 2]);
                    acc[4 * k4 + 3] = __builtin_fmaf(g.w, v[q + 4 * k4 + 3], acc[4 * k4 + 3]);
                }
            }
        }
        if (cok) {
#pragma unroll
            for (int k = 0; k < CB_TW; ++k) {
                const int x = x0 + k;
                if (x < W) gdst[(((size_t)n * H + y) * W + x) * C + cbase + lane] = acc[k] * inv_c;
            }
        }
    }
}

// Both gradients in ONE launch (round 5): blockIdx.z = 2 n + which.  They were two launches in a row, each a chain of
// 9 window rows x 24 loads per wave -- on the coarse PWC levels (9 to 600 workgroups) the second only started when the first
// had drained: 30-57 us of kernel time for a few MFLOP.  Same arithmetic per element.
template <int MD>
__global__ __launch_bounds__(256) void corr2d_bwd_tiled_kernel(const float* __restrict__ gout, const float* __restrict__ in1,
                                                                const float* __restrict__ in2, float* __restrict__ g1,
                                                                float* __restrict__ g2, int C, int H, int W) {
    constexpr int DD = 2 * MD + 1;
    __shared__ __attribute__((aligned(16))) float gw[DD * DD * CB_TW];
    const int n = blockIdx.z >> 1;
    if (blockIdx.z & 1) corr2d_bwd_tiled_body<MD, 1>(gout, in1, g2, C, H, W, n, gw);
    else corr2d_bwd_tiled_body<MD, 0>(gout, in2, g1, C, H, W, n, gw);
}

// ---------------------------------------------------------------------------------------------------------
// Forward, md = 4, C % 4 == 0: 2-D output tile per workgroup.
//
// The one-row kernel above re-streams every in2 row for each of the 9 dy that use it (9x the L2 -> LDS bytes) and
// synchronises twice per (dy, 64-channel chunk) with nothing in flight: at the reference's self-check shape it sat
// at 0.65 TB/s, bound by exposed load latency and by LDS read bandwidth (3.6 FMA per ds_read_b128).  Here a
// 512-thread workgroup owns TH = 8 rows x TW = 64 columns of output pixels:
//   * per 16-channel pass the in1 tile [8][64] and the in2 halo tile [16][72] are staged ONCE into LDS as channel
//     planes (x contiguous): an in2 row is fetched once per tile instead of once per (row, dy) -- 2.25x the
//     algorithmic in2 bytes instead of 9x.  A pass covers 64 contiguous bytes of every pixel (half an L2 line;
//     narrower passes multiply the L2 -> L1 line traffic), the next pass's global loads are issued before the
//     current pass's math and land in registers while it runs (one barrier pair per pass)
//   * a thread owns 2 adjacent pixels of one row and half of the dy range (5 or 4 of the 9; wave = (2-row block,
//     dy half), one heavy and one light wave per SIMD): 90 / 72 accumulators live in VGPRs across the whole
//     channel loop.  Per channel it reads its 2 in1 values (one ds_read_b64) and, per dy, the 10-wide in2 window
//     (five ds_read_b64) for 18 FMAs; lanes run along x, so every 32-lane read group covers 64 distinct banks
//   * plane stride == 2 (mod 32) floats makes the transposing staging stores (4 lanes = 4 channel quads of one
//     pixel) hit 32 distinct banks; cost-volume rows are written as contiguous float2 per lane.
// grid (ceil(W/64), ceil(H/8), B), block 512, 108 KB of LDS (one workgroup = 2 waves per SIMD).
struct CorrTile {
    static constexpr int XG = 32, TW = 64, TH = 8, CC = 16, Q = CC / 4, NT = 512;
    static constexpr int XS1 = TW, XS2 = TW + 8;
    static constexpr int PS1 = TH * XS1 + 2;                  // == 2 (mod 32)
    static constexpr int PS2 = (TH + 8) * XS2 + 2;            // 1154 == 2 (mod 32)
    static constexpr int N1 = TH * TW * Q, N2 = (TH + 8) * (TW + 8) * Q;
    static constexpr int L1 = (N1 + NT - 1) / NT, L2 = (N2 + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = (size_t)CC * (PS1 + PS2) * sizeof(float);
};

__global__ __launch_bounds__(512) void corr2d_fwd_tile_kernel(const float* __restrict__ in1,
                                                               const float* __restrict__ in2,
                                                               float* __restrict__ out, int C, int H, int W) {
    using T = CorrTile;
    constexpr int MD = 4, DD = 9, Q = T::Q;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s1 = smem;                       // [CC][PS1]  rows of XS1
    float* s2 = smem + T::CC * T::PS1;      // [CC][PS2]  rows of XS2
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane & 31;
    const int half = wave >> 2;                       // waves 0-3: dy 0..4, waves 4-7: dy 5..8
    const int r = (wave & 3) * 2 + (lane >> 5);       // row of the tile
    const int d0 = half * 5, nd = 5 - half;
    const int x0t = blockIdx.x * T::TW, y0t = blockIdx.y * T::TH, n = blockIdx.z;
    const size_t img = (size_t)n * H * W;

    float acc[5][2][DD];
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
        for (int k = 0; k < DD; ++k) acc[d][0][k] = acc[d][1][k] = 0.0f;

    float4 pre1[T::L1], pre2[T::L2];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < T::L1; ++i) {
            const int e = tid + i * T::NT;
            const int q = e % Q, px = (e / Q) % T::TW, row = e / (Q * T::TW);
            const int y = y0t + row, x = x0t + px, c = c0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < T::N1 && y < H && x < W && c < C) v = *reinterpret_cast<const float4*>(in1 + (img + (size_t)y * W + x) * C + c);
            pre1[i] = v;
        }
#pragma unroll
        for (int i = 0; i < T::L2; ++i) {
            const int e = tid + i * T::NT;
            const int q = e % Q, px = (e / Q) % (T::TW + 8), row = e / (Q * (T::TW + 8));
            const int y = y0t - MD + row, x = x0t - MD + px, c = c0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < T::N2 && y >= 0 && y < H && x >= 0 && x < W && c < C)
                v = *reinterpret_cast<const float4*>(in2 + (img + (size_t)y * W + x) * C + c);
            pre2[i] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < T::L1; ++i) {
            const int e = tid + i * T::NT;
            if (e < T::N1) {
                const int q = e % Q, px = (e / Q) % T::TW, row = e / (Q * T::TW);
                float* d = s1 + (4 * q) * T::PS1 + row * T::XS1 + px;
                d[0] = pre1[i].x; d[T::PS1] = pre1[i].y; d[2 * T::PS1] = pre1[i].z; d[3 * T::PS1] = pre1[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < T::L2; ++i) {
            const int e = tid + i * T::NT;
            if (e < T::N2) {
                const int q = e % Q, px = (e / Q) % (T::TW + 8), row = e / (Q * (T::TW + 8));
                float* d = s2 + (4 * q) * T::PS2 + row * T::XS2 + px;
                d[0] = pre2[i].x; d[T::PS2] = pre2[i].y; d[2 * T::PS2] = pre2[i].z; d[3 * T::PS2] = pre2[i].w;
            }
        }
    };

    gload(0);
    for (int c0 = 0; c0 < C; c0 += T::CC) {
        __syncthreads();          // every wave is done reading the previous pass
        lstore();
        __syncthreads();
        if (c0 + T::CC < C) gload(c0 + T::CC);      // in flight while this pass is consumed
        const float* a_row = s1 + r * T::XS1 + 2 * g;
        const float* b_row = s2 + (r + d0) * T::XS2 + 2 * g;
#pragma unroll 1
        for (int c = 0; c < T::CC; ++c) {
            const float2 a = *reinterpret_cast<const float2*>(a_row + c * T::PS1);
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                if (d < nd) {          // wave-uniform
                    const float* bp = b_row + c * T::PS2 + d * T::XS2;
                    float b[10];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const float2 t = *reinterpret_cast<const float2*>(bp + 2 * j);
                        b[2 * j] = t.x;
                        b[2 * j + 1] = t.y;
                    }
#pragma unroll
                    for (int k = 0; k < DD; ++k) {
                        acc[d][0][k] = __builtin_fmaf(a.x, b[k], acc[d][0][k]);
                        acc[d][1][k] = __builtin_fmaf(a.y, b[k + 1], acc[d][1][k]);
                    }
                }
            }
        }
    }

    const int y = y0t + r, x = x0t + 2 * g;
    if (y < H && x < W) {
        const float inv_c = 1.0f / (float)C;
        const size_t plane = (size_t)H * W;
        float* o = out + (size_t)n * DD * DD * plane + (size_t)y * W + x;
        const bool vec = ((W & 1) == 0);      // x is even: a float2 store stays inside the row and aligned
#pragma unroll
        for (int d = 0; d < 5; ++d)
            if (d < nd) {
#pragma unroll
                for (int k = 0; k < DD; ++k) {
                    float* op = o + (size_t)((d0 + d) * DD + k) * plane;
                    if (vec) {
                        *reinterpret_cast<float2*>(op) = make_float2(acc[d][0][k] * inv_c, acc[d][1][k] * inv_c);
                    } else {
                        op[0] = acc[d][0][k] * inv_c;
                        if (x + 1 < W) op[1] = acc[d][1][k] * inv_c;
                    }
                }
            }
    }
}

int launch_fwd_tile(const float* in1, const float* in2, float* out, int B, int C, int H, int W, hipStream_t stream) {
    using T = CorrTile;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr2d_fwd_tile_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        attr_set = true;
    }
    dim3 grid(camli_divup(W, T::TW), camli_divup(H, T::TH), B);
    hipLaunchKernelGGL(corr2d_fwd_tile_kernel, grid, dim3(T::NT), T::LDS_BYTES, stream, in1, in2, out, C, H, W);
    return camli_check_launch("camli_corr2d_fwd(tile)");
}

template <int MD>
int launch_bwd(const float* gout, const float* in1, const float* in2, float* g1, float* g2, int B, int C, int H, int W,
               hipStream_t stream) {
    dim3 grid(camli_divup(W, CB_TW), H, 2 * B);
    const int nw = camli_divup(C, 64) < 4 ? camli_divup(C, 64) : 4;
    hipLaunchKernelGGL((corr2d_bwd_tiled_kernel<MD>), grid, dim3(64 * nw), 0, stream, gout, in1, in2, g1, g2, C, H, W);
    return camli_check_launch("camli_corr2d_bwd");
}

template <int MD>
int launch_fwd(const float* in1, const float* in2, float* out, int B, int C, int H, int W, hipStream_t stream) {
    constexpr int DD = 2 * MD + 1;
    const bool vec = (C % 4) == 0;
    const int pad = vec ? 4 : 1;
    const int cs = C < CT_CMAX ? C : CT_CMAX;
    const size_t lds = ((size_t)CT_PX * (cs + pad) + (size_t)(CT_PX + 2 * MD) * (CT_CH + pad) + 4 * DD * CT_PX) * sizeof(float);
    // few workgroups (the coarse levels of the PWC pyramid): one workgroup per displacement row as well
    const long long wgs = (long long)camli_divup(W, CT_PX) * H * B;
    const int zsplit = (wgs < CAMLI_CORR2D_SPLIT_BELOW && (long long)B * DD <= 65535) ? DD : 1;
    dim3 grid(camli_divup(W, CT_PX), H, B * zsplit);
    if (vec)
        hipLaunchKernelGGL((corr2d_fwd_kernel<MD, true>), grid, dim3(256), lds, stream, in1, in2, out, C, H, W, zsplit);
    else
        hipLaunchKernelGGL((corr2d_fwd_kernel<MD, false>), grid, dim3(256), lds, stream, in1, in2, out, C, H, W, zsplit);
    return camli_check_launch("camli_corr2d_fwd");
}

}  // namespace

extern "C" int camli_corr2d_fwd(const float* in1_nhwc, const float* in2_nhwc, float* out_nchw, int B, int C, int H,
                                int W, int md, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!in1_nhwc || !in2_nhwc || !out_nchw) {
        camli_set_error("camli_corr2d_fwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || H < 1 || W < 1 || md < 0 || B > 65535 || H > 65535) {
        camli_set_error("camli_corr2d_fwd: bad shape B=%d C=%d H=%d W=%d md=%d", B, C, H, W, md);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (md) {
        case 1: return launch_fwd<1>(in1_nhwc, in2_nhwc, out_nchw, B, C, H, W, s);
        case 2: return launch_fwd<2>(in1_nhwc, in2_nhwc, out_nchw, B, C, H, W, s);
        case 3: return launch_fwd<3>(in1_nhwc, in2_nhwc, out_nchw, B, C, H, W, s);
        case 4:
            // the 8 x 64 tile kernel needs enough tiles to fill the chip (one 512-thread workgroup per CU); small
            // pyramid levels keep the one-row kernel, whose 64-pixel workgroups spread over more CUs
            if ((C & 3) == 0 && camli_corr2d_use_tile() &&
                (long long)B * camli_divup(H, 8) * camli_divup(W, 64) >= 192) {
                return launch_fwd_tile(in1_nhwc, in2_nhwc, out_nchw, B, C, H, W, s);
            }
            return launch_fwd<4>(in1_nhwc, in2_nhwc, out_nchw, B, C, H, W, s);
        default: {
            hipLaunchKernelGGL(corr2d_fwd_generic_kernel, dim3(2048), dim3(256), 0, s, in1_nhwc, in2_nhwc, out_nchw,
                               B, C, H, W, md);
            return camli_check_launch("camli_corr2d_fwd(generic)");
        }
    }
}

extern "C" int camli_corr2d_bwd(const float* gout_nchw, const float* in1_nhwc, const float* in2_nhwc, float* g1_nhwc,
                                float* g2_nhwc, int B, int C, int H, int W, int md, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gout_nchw || !in1_nhwc || !in2_nhwc || !g1_nhwc || !g2_nhwc) {
        camli_set_error("camli_corr2d_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || H < 1 || W < 1 || md < 0) {
        camli_set_error("camli_corr2d_bwd: bad shape B=%d C=%d H=%d W=%d md=%d", B, C, H, W, md);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (B > 32767 || H > 65535) {       // grid z = 2 B (both gradients in one launch)
        camli_set_error("camli_corr2d_bwd: bad shape B=%d H=%d", B, H);
        return CAMLI_EINVAL;
    }
    switch (md) {
        case 1: return launch_bwd<1>(gout_nchw, in1_nhwc, in2_nhwc, g1_nhwc, g2_nhwc, B, C, H, W, s);
        case 2: return launch_bwd<2>(gout_nchw, in1_nhwc, in2_nhwc, g1_nhwc, g2_nhwc, B, C, H, W, s);
        case 3: return launch_bwd<3>(gout_nchw, in1_nhwc, in2_nhwc, g1_nhwc, g2_nhwc, B, C, H, W, s);
        case 4: return launch_bwd<4>(gout_nchw, in1_nhwc, in2_nhwc, g1_nhwc, g2_nhwc, B, C, H, W, s);
        default: break;
    }
    const size_t total = (size_t)B * H * W * C;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(corr2d_bwd_kernel, dim3(blocks, 2), dim3(256), 0, s, gout_nchw, in1_nhwc, in2_nhwc, g1_nhwc,
                       g2_nhwc, B, C, H, W, md);
    return camli_check_launch("camli_corr2d_bwd(generic)");
}

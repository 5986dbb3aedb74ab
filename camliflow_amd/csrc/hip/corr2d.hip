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

namespace {

constexpr int CT_PX = 64;     // pixels per workgroup (one wave-width)
constexpr int CT_CMAX = 256;  // channels staged per pass

// grid (ceil(W/64), H, B), block 256
template <int MD>
__global__ __launch_bounds__(256) void corr2d_fwd_kernel(const float* __restrict__ in1,
                                                          const float* __restrict__ in2,
                                                          float* __restrict__ out, int C, int H, int W) {
    constexpr int DD = 2 * MD + 1;
    constexpr int HALO = CT_PX + 2 * MD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int px = tid & 63;
    const int cq = tid >> 6;
    const int x0 = blockIdx.x * CT_PX, y = blockIdx.y, n = blockIdx.z;
    const float inv_c = 1.0f / (float)C;

    for (int c0 = 0; c0 < C; c0 += CT_CMAX) {
        const int cc = min(CT_CMAX, C - c0);
        const int ld = cc + 1;
        float* s1 = smem;                    // [CT_PX][ld]
        float* s2 = s1 + CT_PX * ld;         // [HALO][ld]
        float* red = s2 + HALO * ld;         // [4][DD][CT_PX]

        // stage in1 tile: pixels x0..x0+63 of row y, channels c0..c0+cc
        for (int e = tid; e < CT_PX * cc; e += 256) {
            int p = e / cc, c = e - p * cc;
            int x = x0 + p;
            s1[p * ld + c] = (x < W) ? in1[(((size_t)n * H + y) * W + x) * C + c0 + c] : 0.0f;
        }
        const int cbeg = (cc * cq) / 4, cend = (cc * (cq + 1)) / 4;

        for (int dyi = 0; dyi < DD; ++dyi) {
            const int y2 = y + dyi - MD;
            const bool row_ok = (y2 >= 0 && y2 < H);
            __syncthreads();  // previous dy's consumers done with s2/red (and s1 staged)
            if (row_ok) {
                for (int e = tid; e < HALO * cc; e += 256) {
                    int p = e / cc, c = e - p * cc;
                    int x = x0 - MD + p;
                    s2[p * ld + c] =
                        (x >= 0 && x < W) ? in2[(((size_t)n * H + y2) * W + x) * C + c0 + c] : 0.0f;
                }
            }
            __syncthreads();
            float acc[DD];
#pragma unroll
            for (int d = 0; d < DD; ++d) acc[d] = 0.0f;
            if (row_ok) {
                for (int c = cbeg; c < cend; ++c) {
                    float a = s1[px * ld + c];
#pragma unroll
                    for (int d = 0; d < DD; ++d) acc[d] = __builtin_fmaf(a, s2[(px + d) * ld + c], acc[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < DD; ++d) red[(cq * DD + d) * CT_PX + px] = acc[d];
            __syncthreads();
            for (int e = tid; e < DD * CT_PX; e += 256) {
                int d = e >> 6, p = e & 63;
                int x = x0 + p;
                if (x < W) {
                    float s = (red[(0 * DD + d) * CT_PX + p] + red[(1 * DD + d) * CT_PX + p]) +
                              (red[(2 * DD + d) * CT_PX + p] + red[(3 * DD + d) * CT_PX + p]);
                    size_t o = (((size_t)n * DD * DD + dyi * DD + d) * H + y) * W + x;
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

template <int MD>
int launch_fwd(const float* in1, const float* in2, float* out, int B, int C, int H, int W, hipStream_t stream) {
    constexpr int DD = 2 * MD + 1;
    const int cc = C < CT_CMAX ? C : CT_CMAX;
    size_t lds = ((size_t)(CT_PX + CT_PX + 2 * MD) * (cc + 1) + 4 * DD * CT_PX) * sizeof(float);
    dim3 grid(camli_divup(W, CT_PX), H, B);
    hipLaunchKernelGGL((corr2d_fwd_kernel<MD>), grid, dim3(256), lds, stream, in1, in2, out, C, H, W);
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
        case 4: return launch_fwd<4>(in1_nhwc, in2_nhwc, out_nchw, B, C, H, W, s);
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
    const size_t total = (size_t)B * H * W * C;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(corr2d_bwd_kernel, dim3(blocks, 2), dim3(256), 0, s, gout_nchw, in1_nhwc, in2_nhwc, g1_nhwc,
                       g2_nhwc, B, C, H, W, md);
    return camli_check_launch("camli_corr2d_bwd");
}

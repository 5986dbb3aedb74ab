// Convex up-sampling of a flow field (RAFT) and its adjoint, gfx950.
//
// Replaces the composed path of models/utils.py:191-204 of the reference (mask.view -> softmax over the
// 9 taps -> unfold(flow * S) -> multiply -> sum -> permute -> reshape: seven passes over the
// [B, 9*S*S, h, w] mask tensor forward, more backward) and the `0.25 *` pre-scaling of
// models/raft_core.py:195.
//
//   p_k(i,j;y,x)  = softmax_k( mask_scale * mask[b, k*S*S + i*S + j, y, x] ),  k = (dy+1)*3 + (dx+1)
//   out[b,c,y*S+i,x*S+j] = sum_k p_k * S * flow[b,c,y+dy_k,x+dx_k]           (zero outside the image)
//
// HBM-bound on streaming the mask once (4*9*S*S bytes per coarse pixel).  A workgroup owns 64
// consecutive coarse pixels of one row; lanes run along x so every mask plane read is a coalesced
// 256-byte row; the S fine columns of a coarse pixel are interleaved through LDS so each fine
// output row leaves as one contiguous 64*S-float segment.  The adjoint recomputes the softmax from
// the mask (one more read), writes the mask gradient plane by plane (coalesced) and scatters the 18
// neighbour-flow gradients of a pixel with float atomics.
#include "camli_common.h"

namespace {

// grid (ceil(w/64), h, B * IG), block 64: the S fine rows of a coarse row are independent, so IG workgroups share them
// (S/IG rows each).  One wave walking all S*S sub-pixel positions is a chain of 64 dependent (9 loads -> softmax) groups
// on a launch of ~1,000 waves -- one per SIMD, nothing to overlap the load latency with: 160 us for 183 MB.
template <int S, bool BACKWARD>
__global__ __launch_bounds__(64) void convex_upsample_kernel(const float* __restrict__ flow,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ out_or_gout,
                                                              float* __restrict__ gflow, float* __restrict__ gmask,
                                                              int h, int w, float mask_scale, int IG) {
    __shared__ float tile[2][S][64 + 1];    // [channel][j][x_local]
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, x = x0 + lane, y = blockIdx.y, b = blockIdx.z / IG;
    const int i_beg = (blockIdx.z % IG) * (S / IG), i_end = i_beg + S / IG;
    const bool valid = x < w;
    const int xc = valid ? x : w - 1;
    const size_t plane = (size_t)h * w;
    const size_t pix = (size_t)y * w + xc;
    const float* __restrict__ mrow = mask + (size_t)b * 9 * S * S * plane + pix;
    const int W = w * S;                                  // fine width
    const int ncols = min(64, w - x0) * S;                // fine columns this block covers

    // the 9 neighbour flows of this coarse pixel (already multiplied by S)
    float f[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = xc + k % 3 - 1;
        const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            f[k][c] = in ? flow[((size_t)b * 2 + c) * plane + (size_t)yy * w + xx] * (float)S : 0.0f;
    }
    float gf[9][2];
    if (BACKWARD) {
#pragma unroll
        for (int k = 0; k < 9; ++k) gf[k][0] = gf[k][1] = 0.0f;
    }

    for (int i = i_beg; i < i_end; ++i) {
        float* __restrict__ orow0 = out_or_gout + (((size_t)b * 2 + 0) * h * S + (size_t)y * S + i) * W + (size_t)x0 * S;
        float* __restrict__ orow1 = out_or_gout + (((size_t)b * 2 + 1) * h * S + (size_t)y * S + i) * W + (size_t)x0 * S;
        if (BACKWARD) {
            // bring the fine gradient row segment in, coalesced, and de-interleave it: tile[c][j][x]
            for (int e = lane; e < ncols; e += 64) {
                tile[0][e % S][e / S] = orow0[e];
                tile[1][e % S][e / S] = orow1[e];
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float p[9];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                p[k] = mrow[(size_t)(k * S * S + i * S + j) * plane] * mask_scale;
                mx = fmaxf(mx, p[k]);
            }
            float den = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                p[k] = __expf(p[k] - mx);
                den += p[k];
            }
            const float inv = 1.0f / den;
            if (!BACKWARD) {
                float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float pk = p[k] * inv;
                    o0 = __builtin_fmaf(pk, f[k][0], o0);
                    o1 = __builtin_fmaf(pk, f[k][1], o1);
                }
                tile[0][j][lane] = o0;
                tile[1][j][lane] = o1;
            } else {
                const float g0 = tile[0][j][lane], g1 = tile[1][j][lane];
                float gk[9], dot = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    p[k] *= inv;
                    gk[k] = g0 * f[k][0] + g1 * f[k][1];            // d out / d p_k
                    dot = __builtin_fmaf(p[k], gk[k], dot);
                    gf[k][0] = __builtin_fmaf(p[k], g0, gf[k][0]);   // d out / d flow (per neighbour)
                    gf[k][1] = __builtin_fmaf(p[k], g1, gf[k][1]);
                }
                if (valid) {
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        gmask[(size_t)b * 9 * S * S * plane + (size_t)(k * S * S + i * S + j) * plane + pix] =
                            mask_scale * p[k] * (gk[k] - dot);
                }
            }
        }
        __syncthreads();
        if (!BACKWARD) {
            // interleave: fine column e = x_local*S + j
            for (int e = lane; e < ncols; e += 64) {
                orow0[e] = tile[0][e % S][e / S];
                orow1[e] = tile[1][e % S][e / S];
            }
            __syncthreads();
        }
    }
    if (BACKWARD && valid) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    unsafeAtomicAdd(gflow + ((size_t)b * 2 + c) * plane + (size_t)yy * w + xx, gf[k][c] * (float)S);
            }
        }
    }
}

template <bool BACKWARD>
int launch_upsample(const float* flow, const float* mask, float* io, float* gflow, float* gmask, int B, int h, int w,
                    int S, float mask_scale, hipStream_t s, const char* what) {
    if (B < 0 || h < 1 || w < 1 || (S != 4 && S != 8) || B > 65535 || h > 65535) {
        camli_set_error("%s: bad shape B=%d h=%d w=%d scale=%d (scale must be 4 or 8)", what, B, h, w, S);
        return CAMLI_EINVAL;
    }
    // split the fine rows until the launch carries ~4 waves per SIMD (4096 waves)
    int ig = 1;
    while (ig < S && (long long)camli_divup(w, 64) * h * B * ig < 4096 && (long long)B * ig * 2 <= 65535) ig *= 2;
    dim3 grid(camli_divup(w, 64), h, B * ig);
    if (S == 8)
        hipLaunchKernelGGL((convex_upsample_kernel<8, BACKWARD>), grid, dim3(64), 0, s, flow, mask, io, gflow, gmask, h, w,
                           mask_scale, ig);
    else
        hipLaunchKernelGGL((convex_upsample_kernel<4, BACKWARD>), grid, dim3(64), 0, s, flow, mask, io, gflow, gmask, h, w,
                           mask_scale, ig);
    return camli_check_launch(what);
}

}  // namespace

extern "C" int camli_convex_upsample_fwd(const float* flow, const float* mask, float* out, int B, int h, int w,
                                         int scale, float mask_scale, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!flow || !mask || !out) { camli_set_error("camli_convex_upsample_fwd: null pointer"); return CAMLI_EINVAL; }
    return launch_upsample<false>(flow, mask, out, nullptr, nullptr, B, h, w, scale, mask_scale,
                                  reinterpret_cast<hipStream_t>(stream), "camli_convex_upsample_fwd");
}

extern "C" int camli_convex_upsample_bwd(const float* gout, const float* flow, const float* mask, float* gflow,
                                         float* gmask, int B, int h, int w, int scale, float mask_scale,
                                         void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gout || !flow || !mask || !gflow || !gmask) {
        camli_set_error("camli_convex_upsample_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    return launch_upsample<true>(flow, mask, const_cast<float*>(gout), gflow, gmask, B, h, w, scale, mask_scale,
                                 reinterpret_cast<hipStream_t>(stream), "camli_convex_upsample_bwd");
}

// Convex up-sampling of a flow field (RAFT) and its adjoint, gfx950.
//
// Replaces the composed path of models/utils.py:191-204 of the reference (mask.view -> softmax over the
// 9 taps -> unfold(flow * S) -> multiply -> sum -> permute -> reshape: seven passes over the
// [B, 9*S*S, h, w] mask tensor forward, more backward) and the `0.25 *` pre-scaling of
// models/raft_core.py:195.
//
//   p_k(i,j;y,x)  = softmax_k( mask_scale * (mask[b, ch, y, x] + mask_bias[ch]) ),  ch = k*S*S + i*S + j,  k = (dy+1)*3 + (dx+1)
//   (mask_bias: optional, the bias of the mask head's last 1x1 convolution (raft_core.py:187-189) -- added here, the
//   [B, 9*S*S, h, w] tensor is not passed over once more for it; its gradient is the per-channel sum of the mask gradient)
//   out[b,c,y*S+i,x*S+j] = sum_k p_k * S * flow[b,c,y+dy_k,x+dx_k]           (zero outside the image)
//
// HBM-bound on streaming the mask once (4*9*S*S bytes per coarse pixel).  A workgroup owns 64
// consecutive coarse pixels of one row; lanes run along x so every mask plane read is a coalesced
// 256-byte row; the S fine columns of a coarse pixel are interleaved through LDS so each fine
// output row leaves as one contiguous 64*S-float segment.  The adjoint recomputes the softmax from
// the mask (one more read), writes the mask gradient plane by plane (coalesced) and adds the neighbour-flow
// gradients, summed per target column in LDS first, with float atomics.
#include "camli_common.h"

namespace {

// grid (ceil(w/64), h, B * IG), block 64: the S fine rows of a coarse row are independent, so IG workgroups share them
// (S/IG rows each).  One wave walking all S*S sub-pixel positions is a chain of 64 dependent (9 loads -> softmax) groups
// on a launch of ~1,000 waves -- one per SIMD, nothing to overlap the load latency with: 160 us for 183 MB.
template <int S>
__global__ __launch_bounds__(64) void convex_upsample_kernel(const float* __restrict__ flow,
                                                              const float* __restrict__ mask,
                                                              const float* __restrict__ mask_bias,
                                                              float* __restrict__ out, int h, int w, float mask_scale,
                                                              int IG, int out_h) {
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
    for (int i = i_beg; i < i_end; ++i) {
        // out keeps the first out_h fine rows only (the caller's un-padding of a bottom-padded image, folded in): block-uniform
        if (y * S + i >= out_h) break;
        float* __restrict__ orow0 = out + (((size_t)b * 2 + 0) * out_h + (size_t)y * S + i) * W + (size_t)x0 * S;
        float* __restrict__ orow1 = out + (((size_t)b * 2 + 1) * out_h + (size_t)y * S + i) * W + (size_t)x0 * S;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float p[9];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                p[k] = (mrow[(size_t)(k * S * S + i * S + j) * plane] + (mask_bias ? mask_bias[k * S * S + i * S + j] : 0.0f)) * mask_scale;
                mx = fmaxf(mx, p[k]);
            }
            float den = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                p[k] = __expf(p[k] - mx);
                den += p[k];
            }
            const float inv = 1.0f / den;
            float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float pk = p[k] * inv;
                o0 = __builtin_fmaf(pk, f[k][0], o0);
                o1 = __builtin_fmaf(pk, f[k][1], o1);
            }
            tile[0][j][lane] = o0;
            tile[1][j][lane] = o1;
        }
        __syncthreads();
        for (int e = lane; e < ncols; e += 64) {     // interleave: fine column e = x_local*S + j
            orow0[e] = tile[0][e % S][e / S];
            orow1[e] = tile[1][e % S][e / S];
        }
        __syncthreads();
    }
}

// The adjoint, round 3.  Same walk as the forward, but the IG row groups of a coarse row are the IG waves of ONE workgroup
// (block (64, IG)) and the 18 neighbour-flow gradients of a pixel are summed over the row groups and over the three
// lanes that hit the same target column in LDS before they leave: 6 float atomics per coarse pixel (+ 12 per workgroup for
// the two columns next to its 64) instead of 18 * IG -- 4.7 M -> 0.43 M atomics at 8 x 68 x 120, S = 8.
template <int S>
__global__ __launch_bounds__(256) void convex_upsample_bwd_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                                  const float* __restrict__ mask_bias,
                                                                  const float* __restrict__ gout, float* __restrict__ gflow,
                                                                  float* __restrict__ gmask, int h, int w, float mask_scale,
                                                                  int out_h) {
    constexpr int MAXG = 4;
    __shared__ float tile[MAXG][2][S][64 + 1];    // [row group][channel][j][x_local]
    __shared__ float gsum[MAXG][18][64];          // [row group][k*2 + c][x_local]
    const int lane = threadIdx.x, grp = threadIdx.y, IG = blockDim.y;
    const int x0 = blockIdx.x * 64, x = x0 + lane, y = blockIdx.y, b = blockIdx.z;
    const int i_beg = grp * (S / IG), i_end = i_beg + S / IG;
    const bool valid = x < w;
    const int xc = valid ? x : w - 1;
    const size_t plane = (size_t)h * w;
    const size_t pix = (size_t)y * w + xc;
    const float* __restrict__ mrow = mask + (size_t)b * 9 * S * S * plane + pix;
    float* __restrict__ gmrow = gmask + (size_t)b * 9 * S * S * plane + pix;
    const int W = w * S;
    const int ncols = min(64, w - x0) * S;

    float f[9][2], gf[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = xc + k % 3 - 1;
        const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f[k][c] = in ? flow[((size_t)b * 2 + c) * plane + (size_t)yy * w + xx] * (float)S : 0.0f;
            gf[k][c] = 0.0f;
        }
    }
    for (int i = i_beg; i < i_end; ++i) {
        const bool kept = y * S + i < out_h;        // a cropped fine row carries no gradient (its mask gradient is written as 0)
        const float* __restrict__ grow0 = gout + (((size_t)b * 2 + 0) * out_h + (size_t)y * S + i) * W + (size_t)x0 * S;
        const float* __restrict__ grow1 = gout + (((size_t)b * 2 + 1) * out_h + (size_t)y * S + i) * W + (size_t)x0 * S;
        for (int e = lane; e < ncols; e += 64) {     // fine gradient row segment in, coalesced, de-interleaved
            tile[grp][0][e % S][e / S] = kept ? grow0[e] : 0.0f;
            tile[grp][1][e % S][e / S] = kept ? grow1[e] : 0.0f;
        }
        __syncthreads();
#pragma unroll 1      // 118 VGPRs (4 waves per SIMD, the whole launch resident) vs 222 fully unrolled: 99 vs 112 us
        for (int j = 0; j < S; ++j) {
            float p[9];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                p[k] = (mrow[(size_t)(k * S * S + i * S + j) * plane] + (mask_bias ? mask_bias[k * S * S + i * S + j] : 0.0f)) * mask_scale;
                mx = fmaxf(mx, p[k]);
            }
            float den = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                p[k] = __expf(p[k] - mx);
                den += p[k];
            }
            const float inv = 1.0f / den;
            const float g0 = tile[grp][0][j][lane], g1 = tile[grp][1][j][lane];
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
                    gmrow[(size_t)(k * S * S + i * S + j) * plane] = mask_scale * p[k] * (gk[k] - dot);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        gsum[grp][k * 2 + 0][lane] = valid ? gf[k][0] * (float)S : 0.0f;
        gsum[grp][k * 2 + 1][lane] = valid ? gf[k][1] * (float)S : 0.0f;
    }
    __syncthreads();
    // (neighbour row r, channel c): 6 combinations shared by the IG waves; lane L owns target column x0 + L
    for (int q = grp; q < 6; q += IG) {
        const int r = q >> 1, c = q & 1, yy = y + r - 1;
        if (yy < 0 || yy >= h) continue;
        float* __restrict__ grow = gflow + ((size_t)b * 2 + c) * plane + (size_t)yy * w;
        float t = 0.0f, lo = 0.0f, hi = 0.0f;
        for (int g = 0; g < IG; ++g) {
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int src = lane - dx;
                if (src >= 0 && src < 64) t += gsum[g][(r * 3 + dx + 1) * 2 + c][src];
            }
            lo += gsum[g][(r * 3 + 0) * 2 + c][0];        // lane 0's dx = -1 term: target column x0 - 1
            hi += gsum[g][(r * 3 + 2) * 2 + c][63];       // lane 63's dx = +1 term: target column x0 + 64
        }
        if (valid) unsafeAtomicAdd(grow + x, t);
        if (lane == 0 && x0 > 0) unsafeAtomicAdd(grow + x0 - 1, lo);
        if (lane == 63 && x0 + 64 < w) unsafeAtomicAdd(grow + x0 + 64, hi);
    }
}

bool upsample_shape_ok(const char* what, int B, int h, int w, int S) {
    if (B < 0 || h < 1 || w < 1 || (S != 4 && S != 8) || B > 65535 || h > 65535) {
        camli_set_error("%s: bad shape B=%d h=%d w=%d scale=%d (scale must be 4 or 8)", what, B, h, w, S);
        return false;
    }
    return true;
}

// row groups per coarse row: split the fine rows until the launch carries ~4 waves per SIMD (4096 waves)
int upsample_row_groups(int B, int h, int w, int S, int cap) {
    int ig = 1;
    while (ig < S && ig < cap && (long long)camli_divup(w, 64) * h * B * ig < 4096 && (long long)B * ig * 2 <= 65535) ig *= 2;
    return ig;
}

}  // namespace

extern "C" int camli_convex_upsample_rows_fwd(const float* flow, const float* mask, const float* mask_bias, float* out, int B,
                                              int h, int w, int scale, int out_rows, float mask_scale, void* stream);
extern "C" int camli_convex_upsample_rows_bwd(const float* gout, const float* flow, const float* mask, const float* mask_bias,
                                              float* gflow, float* gmask, int B, int h, int w, int scale, int out_rows,
                                              float mask_scale, void* stream);

extern "C" int camli_convex_upsample_fwd(const float* flow, const float* mask, const float* mask_bias, float* out, int B,
                                         int h, int w, int scale, float mask_scale, void* stream) {
    return camli_convex_upsample_rows_fwd(flow, mask, mask_bias, out, B, h, w, scale, h * scale, mask_scale, stream);
}

// out [B,2,out_rows,w*S]: the first out_rows of the h*S fine rows (an image padded at the bottom to a multiple of 8 is
// un-padded by the up-sampling itself instead of a slice + copy each way)
extern "C" int camli_convex_upsample_rows_fwd(const float* flow, const float* mask, const float* mask_bias, float* out, int B,
                                              int h, int w, int scale, int out_rows, float mask_scale, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!flow || !mask || !out) { camli_set_error("camli_convex_upsample_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!upsample_shape_ok("camli_convex_upsample_fwd", B, h, w, scale)) return CAMLI_EINVAL;
    if (out_rows < 1 || out_rows > h * scale) { camli_set_error("camli_convex_upsample_fwd: out_rows %d outside [1, %d]", out_rows, h * scale); return CAMLI_EINVAL; }
    const int ig = upsample_row_groups(B, h, w, scale, 8);
    const dim3 grid(camli_divup(w, 64), h, B * ig);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (scale == 8)
        hipLaunchKernelGGL(convex_upsample_kernel<8>, grid, dim3(64), 0, s, flow, mask, mask_bias, out, h, w, mask_scale, ig, out_rows);
    else
        hipLaunchKernelGGL(convex_upsample_kernel<4>, grid, dim3(64), 0, s, flow, mask, mask_bias, out, h, w, mask_scale, ig, out_rows);
    return camli_check_launch("camli_convex_upsample_fwd");
}

extern "C" int camli_convex_upsample_bwd(const float* gout, const float* flow, const float* mask, const float* mask_bias,
                                         float* gflow, float* gmask, int B, int h, int w, int scale, float mask_scale,
                                         void* stream) {
    return camli_convex_upsample_rows_bwd(gout, flow, mask, mask_bias, gflow, gmask, B, h, w, scale, h * scale, mask_scale, stream);
}

extern "C" int camli_convex_upsample_rows_bwd(const float* gout, const float* flow, const float* mask, const float* mask_bias,
                                              float* gflow, float* gmask, int B, int h, int w, int scale, int out_rows,
                                              float mask_scale, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (out_rows < 1 || out_rows > h * scale) { camli_set_error("camli_convex_upsample_bwd: out_rows %d outside [1, %d]", out_rows, h * scale); return CAMLI_EINVAL; }
    if (!gout || !flow || !mask || !gflow || !gmask) {
        camli_set_error("camli_convex_upsample_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (!upsample_shape_ok("camli_convex_upsample_bwd", B, h, w, scale)) return CAMLI_EINVAL;
    const dim3 grid(camli_divup(w, 64), h, B), block(64, upsample_row_groups(B, h, w, scale, 4));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (scale == 8)
        hipLaunchKernelGGL(convex_upsample_bwd_kernel<8>, grid, block, 0, s, flow, mask, mask_bias, gout, gflow, gmask, h, w,
                           mask_scale, out_rows);
    else
        hipLaunchKernelGGL(convex_upsample_bwd_kernel<4>, grid, block, 0, s, flow, mask, mask_bias, gout, gflow, gmask, h, w,
                           mask_scale, out_rows);
    return camli_check_launch("camli_convex_upsample_bwd");
}

// Depth-wise point set-convolution core ("PointConvDW" gather * weight -> max over neighbours)
// and its adjoint, gfx950.
//
// Replaces the composed tail of models/point_conv.py:122-128 of the reference:
//     features = batch_indexing(features, knn_indices)        # materialises [B,C,N,k]
//     features = features * self.weight_net(knn_offset)       # another [B,C,N,k]
//     features = torch.max(features, dim=-1)[0]
// whose backward scatter-adds a [B,C,N,k] tensor back into [B,C,M] with one atomic per element.
//
//   out[b,c,n] = max_j feat[b,c,idx[b,n,j]] * weight[b,c,n,j]      (first maximum wins ties)
//
// HBM-bound on streaming `weight` once (4*B*C*N*k bytes); the gathered feature rows (M floats per
// (b,c)) stay in L1/L2.  Lane = query point n (coalesced out/arg stores, each lane streams its own
// contiguous k-float weight row with 16-byte loads), 4 channels per thread so the index row is
// read once per 4 channels.  The adjoint touches ONE neighbour per (b,c,n): one float atomic into
// the feature gradient and one plain read-modify-write into the (persistent, caller-zeroed)
// weight-gradient buffer -- k times fewer atomics than the composed path.
#include "camli_common.h"

namespace {

constexpr int SC_CPT = 4;    // channels per thread
constexpr int SC_CGRP = 4;   // channel groups per block (waves)

// grid (ceil(N/64), ceil(C/16), B), block 256
template <int K>
__global__ __launch_bounds__(256) void pointconv_dw_fwd_kernel(const float* __restrict__ feat,
                                                                const float* __restrict__ weight,
                                                                const int64_t* __restrict__ idx, int idx_stride,
                                                                float* __restrict__ out,
                                                                unsigned char* __restrict__ arg, int C, int M,
                                                                int N, int k_rt) {
    const int k = K > 0 ? K : k_rt;
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = (blockIdx.y * SC_CGRP + (threadIdx.x >> 6)) * SC_CPT;
    const int b = blockIdx.z;
    if (n >= N || c0 >= C) return;
    const int64_t* __restrict__ irow = idx + ((size_t)b * N + n) * idx_stride;

    float best[SC_CPT];
    int barg[SC_CPT];
    const float* frow[SC_CPT];
    const float* wrow[SC_CPT];
#pragma unroll
    for (int u = 0; u < SC_CPT; ++u) {
        const int c = min(c0 + u, C - 1);
        best[u] = -INFINITY;
        barg[u] = 0;
        frow[u] = feat + ((size_t)b * C + c) * M;
        wrow[u] = weight + (((size_t)b * C + c) * N + n) * (size_t)k;
    }
    if (K > 0 && (K % 4) == 0) {
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += 4) {
            int m[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) m[t] = (int)irow[j0 + t];
#pragma unroll
            for (int u = 0; u < SC_CPT; ++u) {
                const float4 w4 = *reinterpret_cast<const float4*>(wrow[u] + j0);
                const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float p = frow[u][m[t]] * wv[t];
                    const bool gt = p > best[u];
                    best[u] = gt ? p : best[u];
                    barg[u] = gt ? j0 + t : barg[u];
                }
            }
        }
    } else {
        for (int j = 0; j < k; ++j) {
            const int m = (int)irow[j];
#pragma unroll
            for (int u = 0; u < SC_CPT; ++u) {
                const float p = frow[u][m] * wrow[u][j];
                const bool gt = p > best[u];
                best[u] = gt ? p : best[u];
                barg[u] = gt ? j : barg[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SC_CPT; ++u) {
        if (c0 + u < C) {
            const size_t o = ((size_t)b * C + c0 + u) * N + n;
            out[o] = best[u];
            arg[o] = (unsigned char)barg[u];
        }
    }
}

// thread = (b, c, n), n fastest
__global__ __launch_bounds__(256) void pointconv_dw_bwd_kernel(const float* __restrict__ gout,
                                                                const float* __restrict__ feat,
                                                                const float* __restrict__ weight,
                                                                const int64_t* __restrict__ idx, int idx_stride,
                                                                const unsigned char* __restrict__ arg,
                                                                float* __restrict__ gfeat, float* __restrict__ gweight,
                                                                int B, int C, int M, int N, int k) {
    const size_t total = (size_t)B * C * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % N);
        const size_t bc = e / N;
        const int b = (int)(bc / C);
        const float g = gout[e];
        const int j = arg[e];
        const int m = (int)idx[((size_t)b * N + n) * idx_stride + j];
        const size_t wpos = e * (size_t)k + j;
        const float w = weight[wpos];
        const float f = feat[bc * M + m];
        if (gfeat) unsafeAtomicAdd(gfeat + bc * M + m, g * w);
        if (gweight) gweight[wpos] += g * f;
    }
}

}  // namespace

extern "C" int camli_pointconv_dw_fwd(const float* feat, const float* weight, const int64_t* idx, int idx_stride,
                                      float* out, unsigned char* arg, int B, int C, int M, int N, int k,
                                      void* stream) {
    if (!feat || !weight || !idx || !out || !arg) {
        camli_set_error("camli_pointconv_dw_fwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 1 || k < 1 || k > 255 || idx_stride < k || B > 65535) {
        camli_set_error("camli_pointconv_dw_fwd: bad shape B=%d C=%d M=%d N=%d k=%d idx_stride=%d", B, C, M, N, k,
                        idx_stride);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(camli_divup(N, 64), camli_divup(C, SC_CPT * SC_CGRP), B);
#define CAMLI_DW_LAUNCH(KK)                                                                                    \
    hipLaunchKernelGGL((pointconv_dw_fwd_kernel<KK>), grid, dim3(256), 0, s, feat, weight, idx, idx_stride, out, \
                       arg, C, M, N, k)
    switch (k) {
        case 4: CAMLI_DW_LAUNCH(4); break;
        case 8: CAMLI_DW_LAUNCH(8); break;
        case 16: CAMLI_DW_LAUNCH(16); break;
        case 32: CAMLI_DW_LAUNCH(32); break;
        default: CAMLI_DW_LAUNCH(0); break;
    }
#undef CAMLI_DW_LAUNCH
    return camli_check_launch("camli_pointconv_dw_fwd");
}

extern "C" int camli_pointconv_dw_bwd(const float* gout, const float* feat, const float* weight, const int64_t* idx,
                                      int idx_stride, const unsigned char* arg, float* gfeat, float* gweight, int B,
                                      int C, int M, int N, int k, void* stream) {
    if (!gout || !feat || !weight || !idx || !arg || (!gfeat && !gweight)) {
        camli_set_error("camli_pointconv_dw_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 1 || k < 1 || k > 255 || idx_stride < k) {
        camli_set_error("camli_pointconv_dw_bwd: bad shape B=%d C=%d M=%d N=%d k=%d", B, C, M, N, k);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    const size_t total = (size_t)B * C * N;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(pointconv_dw_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gout,
                       feat, weight, idx, idx_stride, arg, gfeat, gweight, B, C, M, N, k);
    return camli_check_launch("camli_pointconv_dw_bwd");
}

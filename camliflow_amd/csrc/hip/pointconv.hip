// PointConv neighbourhood mixing and its adjoint, gfx950.
//
// Replaces the composed core of models/point_conv.py:60-66 of the reference:
//     knn_features = batch_indexing(features_cl, knn_indices, layout='channel_last')   # [B,n,k,CH]
//     out = torch.matmul(weights.transpose(1, 2), knn_features)                          # [B,n,Wn,CH]
// (CH = in_channels + 3, Wn = 16) whose backward goes through index_put_(accumulate=True) -- a
// device-wide sort + segmented reduction of the materialised [B,n,k,CH] gradient.
//
//   out[b,n,w,ch] = sum_j wgt[b,w,n,j] * feat[b, idx[b,n,j], ch]
//
// One workgroup owns one sampled point (b, n).  The k neighbour rows are channel-contiguous, so
// lanes run along ch: every gather is a coalesced row read and the Wn*k weights of the point are
// wave-uniform (scalar loads).  Nothing of size [B,n,k,CH] ever exists in HBM.
// Adjoint: gout[n] (Wn x CH) and the k gathered rows are staged in LDS once; phase A gives the
// weight gradient (one thread per (w, j), a CH-long dot product), phase B the feature gradient
// (lanes along ch again, Wn FMAs per neighbour) added with row-coalesced float atomics.
#include "camli_common.h"

namespace {

constexpr int PC_WN_MAX = 16;

// grid (N, B), block = 64 * ceil(CH / 64)
__global__ __launch_bounds__(256) void pointconv_mix_fwd_kernel(const float* __restrict__ feat /*[B,M,CH]*/,
                                                                 const float* __restrict__ wgt /*[B,Wn,N,k]*/,
                                                                 const int64_t* __restrict__ idx, int idx_stride,
                                                                 float* __restrict__ out /*[B,N,Wn,CH]*/, int M, int N,
                                                                 int CH, int Wn, int k) {
    const int n = blockIdx.x, b = blockIdx.y;
    const int64_t* __restrict__ irow = idx + ((size_t)b * N + n) * idx_stride;
    const float* __restrict__ wbase = wgt + (size_t)b * Wn * N * k + (size_t)n * k;   // + w*N*k + j
    for (int ch = threadIdx.x; ch < CH; ch += blockDim.x) {
        float acc[PC_WN_MAX];
#pragma unroll
        for (int w = 0; w < PC_WN_MAX; ++w) acc[w] = 0.0f;
        for (int j = 0; j < k; ++j) {
            const int m = (int)irow[j];
            const float f = feat[((size_t)b * M + m) * CH + ch];
#pragma unroll
            for (int w = 0; w < PC_WN_MAX; ++w)
                if (w < Wn) acc[w] = __builtin_fmaf(wbase[(size_t)w * N * k + j], f, acc[w]);
        }
        float* __restrict__ o = out + (((size_t)b * N + n) * Wn) * CH + ch;
#pragma unroll
        for (int w = 0; w < PC_WN_MAX; ++w)
            if (w < Wn) o[(size_t)w * CH] = acc[w];
    }
}

// grid (N, B), block 256.  dynamic LDS: (Wn + k) * CH floats
__global__ __launch_bounds__(256) void pointconv_mix_bwd_kernel(const float* __restrict__ gout /*[B,N,Wn,CH]*/,
                                                                 const float* __restrict__ feat,
                                                                 const float* __restrict__ wgt,
                                                                 const int64_t* __restrict__ idx, int idx_stride,
                                                                 float* __restrict__ gfeat /*[B,M,CH], zeroed*/,
                                                                 float* __restrict__ gwgt /*[B,Wn,N,k]*/, int M, int N,
                                                                 int CH, int Wn, int k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sg = lds;             // [Wn][CH]
    float* sf = lds + Wn * CH;   // [k][CH]
    const int n = blockIdx.x, b = blockIdx.y;
    const int64_t* __restrict__ irow = idx + ((size_t)b * N + n) * idx_stride;
    const float* __restrict__ g = gout + (((size_t)b * N + n) * Wn) * CH;
    for (int e = threadIdx.x; e < Wn * CH; e += blockDim.x) sg[e] = g[e];
    for (int e = threadIdx.x; e < k * CH; e += blockDim.x) {
        const int j = e / CH, ch = e - j * CH;
        sf[e] = feat[((size_t)b * M + (int)irow[j]) * CH + ch];
    }
    __syncthreads();
    // phase A: gwgt[b,w,n,j] = <gout[n,w,:], feat[idx_j,:]>
    if (gwgt) {
        for (int t = threadIdx.x; t < Wn * k; t += blockDim.x) {
            const int w = t / k, j = t - w * k;
            float acc = 0.0f;
            for (int ch = 0; ch < CH; ++ch) acc = __builtin_fmaf(sg[w * CH + ch], sf[j * CH + ch], acc);
            gwgt[((size_t)b * Wn + w) * N * k + (size_t)n * k + j] = acc;
        }
    }
    // phase B: gfeat[b,idx_j,ch] += sum_w wgt[b,w,n,j] * gout[n,w,ch]
    if (gfeat) {
        const float* __restrict__ wbase = wgt + (size_t)b * Wn * N * k + (size_t)n * k;
        for (int ch = threadIdx.x; ch < CH; ch += blockDim.x) {
            float gcol[PC_WN_MAX];
#pragma unroll
            for (int w = 0; w < PC_WN_MAX; ++w) gcol[w] = w < Wn ? sg[w * CH + ch] : 0.0f;
            for (int j = 0; j < k; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int w = 0; w < PC_WN_MAX; ++w)
                    if (w < Wn) acc = __builtin_fmaf(wbase[(size_t)w * N * k + j], gcol[w], acc);
                unsafeAtomicAdd(gfeat + ((size_t)b * M + (int)irow[j]) * CH + ch, acc);
            }
        }
    }
}

int mix_args_ok(const char* what, int B, int M, int N, int CH, int Wn, int k, int idx_stride) {
    if (B < 0 || M < 1 || N < 1 || CH < 1 || Wn < 1 || Wn > PC_WN_MAX || k < 1 || idx_stride < k || B > 65535) {
        camli_set_error("%s: bad shape B=%d M=%d N=%d CH=%d Wn=%d (<= %d) k=%d", what, B, M, N, CH, Wn, PC_WN_MAX, k);
        return 0;
    }
    return 1;
}

}  // namespace

extern "C" int camli_pointconv_mix_fwd(const float* feat_cl, const float* wgt, const int64_t* idx, int idx_stride,
                                       float* out, int B, int M, int N, int CH, int Wn, int k, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!feat_cl || !wgt || !idx || !out) { camli_set_error("camli_pointconv_mix_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!mix_args_ok("camli_pointconv_mix_fwd", B, M, N, CH, Wn, k, idx_stride)) return CAMLI_EINVAL;
    if (B == 0) return CAMLI_OK;
    const int threads = 64 * (CH > 192 ? 4 : camli_divup(CH, 64));
    hipLaunchKernelGGL(pointconv_mix_fwd_kernel, dim3(N, B), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream),
                       feat_cl, wgt, idx, idx_stride, out, M, N, CH, Wn, k);
    return camli_check_launch("camli_pointconv_mix_fwd");
}

extern "C" int camli_pointconv_mix_bwd(const float* gout, const float* feat_cl, const float* wgt, const int64_t* idx,
                                       int idx_stride, float* gfeat_cl, float* gwgt, int B, int M, int N, int CH, int Wn,
                                       int k, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gout || !feat_cl || !wgt || !idx || (!gfeat_cl && !gwgt)) {
        camli_set_error("camli_pointconv_mix_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (!mix_args_ok("camli_pointconv_mix_bwd", B, M, N, CH, Wn, k, idx_stride)) return CAMLI_EINVAL;
    const size_t lds = (size_t)(Wn + k) * CH * sizeof(float);
    if (lds > 150 * 1024) {
        camli_set_error("camli_pointconv_mix_bwd: (Wn + k) * CH = %d floats exceeds the LDS tile", (Wn + k) * CH);
        return CAMLI_ENOTSUP;
    }
    if (B == 0) return CAMLI_OK;
    hipLaunchKernelGGL(pointconv_mix_bwd_kernel, dim3(N, B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), gout,
                       feat_cl, wgt, idx, idx_stride, gfeat_cl, gwgt, M, N, CH, Wn, k);
    return camli_check_launch("camli_pointconv_mix_bwd");
}

// PointPWC learnable cost volume (PWC-style Correlation3D), gfx950.
//
// Replaces the composed body of models/camlipwc_l_core.py:53-106 of the reference:
//     cat([f1 expanded over k | gather(f2, knn) | knn_xyz2 - xyz1])  ->  [B, 2C+3, N, k]      (materialised)
//     p2p = cost_mlp(cat)            MLP2d(2C+3 -> C -> C, leaky 0.1)
//     p2n = sum_k weight_net2(dxyz) * p2p
//     out = sum_k weight_net1(dxyz1) * gather(p2n, knn1)
//
// The first cost_mlp layer is linear in the concatenated channels, so it splits by input block:
//     W1 . [f1(n) | f2(q) | d]  =  (W1a . f1)[:, n]  +  (W1b . f2)[:, q]  +  (W1c . d + b1)[:, n, j]
// The two feature terms are per-POINT 1x1 convolutions (N*C*C instead of N*k*(2C+3)*C multiply-adds, done by the
// caller on the matrix cores) and the [B,2C+3,N,k] tensor never exists.  Kernels here (layout [B,C,N,k], k fastest;
// lanes run along the flattened (n, j) axis so every global access except the one gather is coalesced):
//   pair    h1[c,n,j]  = leaky( A[c,n] + Bm[c, idx[n,j]] + E[c,n,j] )                       (E = W1c.d + b1)
//           adjoint:     gpre = gh1 * leaky'(h1)  (written once: it is gE and the source of the sorted scatter that
//                        gives gBm), gA[c,n] = sum_j gpre  (16-lane DPP reduction)
//   ksum    out[c,n]   = sum_j w[c,n,j] * h[c,n,j]                       adjoint: gw = g*h, gh = g*w   (one pass)
//   gwsum   out[c,n]   = sum_j w[c,n,j] * feat[c, idx[n,j]]              adjoint: gw = g*feat[idx], t = g*w (-> scatter)
// k must be a power of two <= 64 (the models use 16).
#include "camli_common.h"

namespace {

__device__ __forceinline__ float group_sum(float v, int k) {
    for (int off = k >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// grid (ceil(N*k / 256), CY, B), block 256
template <bool BACKWARD>
__global__ __launch_bounds__(256) void pwc3d_pair_kernel(const float* __restrict__ A /*[B,C,N]*/,
                                                          const float* __restrict__ Bm /*[B,C,M]*/,
                                                          const float* __restrict__ E /*[B,C,N,k]  (bwd: gh1)*/,
                                                          const int64_t* __restrict__ idx /*[B,N,k]*/,
                                                          float* __restrict__ h1 /*fwd out; bwd: h1 in*/,
                                                          float* __restrict__ gpre /*bwd out [B,C,N,k]*/,
                                                          float* __restrict__ gA /*bwd out [B,C,N]*/, int C, int M, int N,
                                                          int k, float slope) {
    const int e = blockIdx.x * 256 + threadIdx.x;          // flat (n, j)
    const int b = blockIdx.z;
    const int nk = N * k;
    const bool ok = e < nk;
    const int ec = ok ? e : nk - 1;
    const int n = ec / k;
    const size_t plane = (size_t)nk;
    if (!BACKWARD) {
        const int m = (int)idx[(size_t)b * nk + ec];
        for (int c = blockIdx.y; c < C; c += gridDim.y) {
            const size_t row = (size_t)b * C + c;
            const float v = A[row * N + n] + Bm[row * M + m] + E[row * plane + ec];
            if (ok) h1[row * plane + ec] = v > 0.0f ? v : slope * v;
        }
    } else {
        for (int c = blockIdx.y; c < C; c += gridDim.y) {
            const size_t row = (size_t)b * C + c;
            const float g = ok ? E[row * plane + ec] * (h1[row * plane + ec] > 0.0f ? 1.0f : slope) : 0.0f;
            if (ok) gpre[row * plane + ec] = g;
            const float s = group_sum(g, k);
            if (ok && (ec % k) == 0) gA[row * N + n] = s;
        }
    }
}

// out[c,n] = sum_j w[c,n,j] * h[c,n,j];  h = GATHER ? feat[c, idx[n,j]] : hk[c,n,j]
template <bool GATHER>
__global__ __launch_bounds__(256) void ksum_fwd_kernel(const float* __restrict__ w, const float* __restrict__ hk,
                                                        const float* __restrict__ feat, const int64_t* __restrict__ idx,
                                                        float* __restrict__ out, int C, int M, int N, int k) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    const int nk = N * k;
    const bool ok = e < nk;
    const int ec = ok ? e : nk - 1;
    const int n = ec / k;
    const int m = GATHER ? (int)idx[(size_t)b * nk + ec] : 0;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        const size_t row = (size_t)b * C + c;
        const float h = GATHER ? feat[row * M + m] : hk[row * nk + ec];
        const float v = ok ? w[row * nk + ec] * h : 0.0f;
        const float s = group_sum(v, k);
        if (ok && (ec % k) == 0) out[row * N + n] = s;
    }
}

// gw[c,n,j] = g[c,n] * h;  gh_or_t[c,n,j] = g[c,n] * w[c,n,j]
template <bool GATHER>
__global__ __launch_bounds__(256) void ksum_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                        const float* __restrict__ hk, const float* __restrict__ feat,
                                                        const int64_t* __restrict__ idx, float* __restrict__ gw,
                                                        float* __restrict__ gh, int C, int M, int N, int k) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    const int nk = N * k;
    if (e >= nk) return;
    const int n = e / k;
    const int m = GATHER ? (int)idx[(size_t)b * nk + e] : 0;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        const size_t row = (size_t)b * C + c;
        const float gv = g[row * N + n];
        const float h = GATHER ? feat[row * M + m] : hk[row * nk + e];
        if (gw) gw[row * nk + e] = gv * h;
        if (gh) gh[row * nk + e] = gv * w[row * nk + e];
    }
}

int pw_args_ok(const char* what, int B, int C, int M, int N, int k) {
    if (B < 0 || C < 1 || M < 1 || N < 1 || k < 1 || k > 64 || (k & (k - 1)) != 0 || B > 65535 ||
        (long long)N * k > 2147483647LL) {
        camli_set_error("%s: bad shape B=%d C=%d M=%d N=%d k=%d (k must be a power of two <= 64)", what, B, C, M, N, k);
        return 0;
    }
    return 1;
}

dim3 pw_grid(int B, int C, int N, int k) { return dim3(camli_divup(N * k, 256), C < 32 ? C : 32, B); }

}  // namespace

extern "C" int camli_pwc3d_pair_fwd(const float* a, const float* bm, const float* e, const int64_t* idx, float* h1, int B,
                                    int C, int M, int N, int k, float slope, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!a || !bm || !e || !idx || !h1) { camli_set_error("camli_pwc3d_pair_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!pw_args_ok("camli_pwc3d_pair_fwd", B, C, M, N, k)) return CAMLI_EINVAL;
    hipLaunchKernelGGL((pwc3d_pair_kernel<false>), pw_grid(B, C, N, k), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a,
                       bm, e, idx, h1, nullptr, nullptr, C, M, N, k, slope);
    return camli_check_launch("camli_pwc3d_pair_fwd");
}

extern "C" int camli_pwc3d_pair_bwd(const float* gh1, const float* h1, float* gpre, float* ga, int B, int C, int N, int k,
                                    float slope, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gh1 || !h1 || !gpre || !ga) { camli_set_error("camli_pwc3d_pair_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!pw_args_ok("camli_pwc3d_pair_bwd", B, C, 1, N, k)) return CAMLI_EINVAL;
    hipLaunchKernelGGL((pwc3d_pair_kernel<true>), pw_grid(B, C, N, k), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       nullptr, nullptr, gh1, nullptr, const_cast<float*>(h1), gpre, ga, C, 1, N, k, slope);
    return camli_check_launch("camli_pwc3d_pair_bwd");
}

extern "C" int camli_ksum_fwd(const float* w, const float* h, float* out, int B, int C, int N, int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!w || !h || !out) { camli_set_error("camli_ksum_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!pw_args_ok("camli_ksum_fwd", B, C, 1, N, k)) return CAMLI_EINVAL;
    hipLaunchKernelGGL((ksum_fwd_kernel<false>), pw_grid(B, C, N, k), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, h,
                       nullptr, nullptr, out, C, 1, N, k);
    return camli_check_launch("camli_ksum_fwd");
}

extern "C" int camli_ksum_bwd(const float* g, const float* w, const float* h, float* gw, float* gh, int B, int C, int N,
                              int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!g || !w || !h || (!gw && !gh)) { camli_set_error("camli_ksum_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!pw_args_ok("camli_ksum_bwd", B, C, 1, N, k)) return CAMLI_EINVAL;
    hipLaunchKernelGGL((ksum_bwd_kernel<false>), pw_grid(B, C, N, k), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, w,
                       h, nullptr, nullptr, gw, gh, C, 1, N, k);
    return camli_check_launch("camli_ksum_bwd");
}

extern "C" int camli_gather_wsum_fwd(const float* w, const float* feat, const int64_t* idx, float* out, int B, int C, int M,
                                     int N, int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!w || !feat || !idx || !out) { camli_set_error("camli_gather_wsum_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!pw_args_ok("camli_gather_wsum_fwd", B, C, M, N, k)) return CAMLI_EINVAL;
    hipLaunchKernelGGL((ksum_fwd_kernel<true>), pw_grid(B, C, N, k), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                       nullptr, feat, idx, out, C, M, N, k);
    return camli_check_launch("camli_gather_wsum_fwd");
}

extern "C" int camli_gather_wsum_bwd(const float* g, const float* w, const float* feat, const int64_t* idx, float* gw,
                                     float* t, int B, int C, int M, int N, int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!g || !w || !feat || !idx || (!gw && !t)) { camli_set_error("camli_gather_wsum_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!pw_args_ok("camli_gather_wsum_bwd", B, C, M, N, k)) return CAMLI_EINVAL;
    hipLaunchKernelGGL((ksum_bwd_kernel<true>), pw_grid(B, C, N, k), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, w,
                       nullptr, feat, idx, gw, t, C, M, N, k);
    return camli_check_launch("camli_gather_wsum_bwd");
}

// Selective-kernel fusion of the 2-D and 3-D branches (models/clfm.py:170-213), gfx950.
//
// After the two aligning convolutions the reference runs, on full-size [B,C,P] tensors,
//   s = mean_p(a + b);  w = softmax(gate(s)) in R^{B x C x 2};  out = a * w0 + b * w1
// as add, mean, mul, mul, add (11 tensor passes) and about twice that in the backward.  Here the
// full-size work is four streaming kernels (5 + 6 passes); the tiny gate MLP on [B,C] stays in torch.
//
//   sk_pool_fwd   s[b,c]   = (1/P) sum_p (a + b)
//   sk_mix_fwd    out      = a * w0[b,c] + b * w1[b,c]
//   sk_mix_bwd_w  gw0[b,c] = sum_p g * a,   gw1[b,c] = sum_p g * b
//   sk_mix_bwd_x  ga = g * w0 + gs[b,c] / P,  gb = g * w1 + gs[b,c] / P     (mix and pool adjoints together)
//
// All HBM-bound: one workgroup per (b,c) row (or row chunk), 16-byte accesses when P % 4 == 0.
#include "camli_common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// grid B*C, block 256
__global__ __launch_bounds__(256) void sk_pool_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ s, int P) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * P;
    float acc = 0.0f;
    if ((P & 3) == 0) {
        const float4* a4 = reinterpret_cast<const float4*>(a + base);
        const float4* b4 = reinterpret_cast<const float4*>(b + base);
        for (int i = threadIdx.x; i < (P >> 2); i += 256) {
            const float4 x = a4[i], y = b4[i];
            acc += ((x.x + y.x) + (x.y + y.y)) + ((x.z + y.z) + (x.w + y.w));
        }
    } else {
        for (int i = threadIdx.x; i < P; i += 256) acc += a[base + i] + b[base + i];
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) s[blockIdx.x] = tot / (float)P;
}

// grid (ceil(P/1024), B*C), block 256: 4 elements per thread
__global__ __launch_bounds__(256) void sk_mix_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ w, float* __restrict__ out, int P) {
    const int row = blockIdx.y;
    const float w0 = w[2 * row], w1 = w[2 * row + 1];
    const size_t base = (size_t)row * P;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if ((P & 3) == 0) {
        if (i < P) {
            const float4 x = *reinterpret_cast<const float4*>(a + base + i), y = *reinterpret_cast<const float4*>(b + base + i);
            float4 o;
            o.x = x.x * w0 + y.x * w1;
            o.y = x.y * w0 + y.y * w1;
            o.z = x.z * w0 + y.z * w1;
            o.w = x.w * w0 + y.w * w1;
            *reinterpret_cast<float4*>(out + base + i) = o;
        }
    } else {
        for (int j = i; j < min(i + 4, P); ++j) out[base + j] = a[base + j] * w0 + b[base + j] * w1;
    }
}

// grid B*C, block 256
__global__ __launch_bounds__(256) void sk_mix_bwd_w_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                           const float* __restrict__ b, float* __restrict__ gw, int P) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * P;
    float s0 = 0.0f, s1 = 0.0f;
    if ((P & 3) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(g + base);
        const float4* a4 = reinterpret_cast<const float4*>(a + base);
        const float4* b4 = reinterpret_cast<const float4*>(b + base);
        for (int i = threadIdx.x; i < (P >> 2); i += 256) {
            const float4 gg = g4[i], x = a4[i], y = b4[i];
            s0 += (gg.x * x.x + gg.y * x.y) + (gg.z * x.z + gg.w * x.w);
            s1 += (gg.x * y.x + gg.y * y.y) + (gg.z * y.z + gg.w * y.w);
        }
    } else {
        for (int i = threadIdx.x; i < P; i += 256) {
            s0 += g[base + i] * a[base + i];
            s1 += g[base + i] * b[base + i];
        }
    }
    const float t0 = block_sum_256(s0, red);
    const float t1 = block_sum_256(s1, red);
    if (threadIdx.x == 0) {
        gw[2 * blockIdx.x] = t0;
        gw[2 * blockIdx.x + 1] = t1;
    }
}

// grid (ceil(P/1024), B*C), block 256
__global__ __launch_bounds__(256) void sk_mix_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                           const float* __restrict__ gs, float* __restrict__ ga,
                                                           float* __restrict__ gb, int P) {
    const int row = blockIdx.y;
    const float w0 = w[2 * row], w1 = w[2 * row + 1];
    const float add = gs ? gs[row] / (float)P : 0.0f;
    const size_t base = (size_t)row * P;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if ((P & 3) == 0) {
        if (i < P) {
            const float4 gg = *reinterpret_cast<const float4*>(g + base + i);
            float4 x, y;
            x.x = gg.x * w0 + add; x.y = gg.y * w0 + add; x.z = gg.z * w0 + add; x.w = gg.w * w0 + add;
            y.x = gg.x * w1 + add; y.y = gg.y * w1 + add; y.z = gg.z * w1 + add; y.w = gg.w * w1 + add;
            *reinterpret_cast<float4*>(ga + base + i) = x;
            *reinterpret_cast<float4*>(gb + base + i) = y;
        }
    } else {
        for (int j = i; j < min(i + 4, P); ++j) {
            ga[base + j] = g[base + j] * w0 + add;
            gb[base + j] = g[base + j] * w1 + add;
        }
    }
}

bool sk_args_ok(const char* what, int B, int C, int P) {
    if (B < 0 || C < 1 || P < 1 || (long long)B * C > 0x7fffffffLL || (long long)B * C > 65535LL * 65535LL) {
        camli_set_error("%s: bad shape B=%d C=%d P=%d", what, B, C, P);
        return false;
    }
    return true;
}

}  // namespace

extern "C" int camli_sk_pool_fwd(const float* a, const float* b, float* s, int B, int C, int P, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!a || !b || !s) { camli_set_error("camli_sk_pool_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!sk_args_ok("camli_sk_pool_fwd", B, C, P)) return CAMLI_EINVAL;
    hipLaunchKernelGGL(sk_pool_fwd_kernel, dim3(B * C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, s, P);
    return camli_check_launch("camli_sk_pool_fwd");
}

extern "C" int camli_sk_mix_fwd(const float* a, const float* b, const float* w, float* out, int B, int C, int P,
                                void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!a || !b || !w || !out) { camli_set_error("camli_sk_mix_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!sk_args_ok("camli_sk_mix_fwd", B, C, P) || B * C > 65535) {
        camli_set_error("camli_sk_mix_fwd: bad shape B=%d C=%d P=%d", B, C, P);
        return CAMLI_EINVAL;
    }
    hipLaunchKernelGGL(sk_mix_fwd_kernel, dim3(camli_divup(P, 1024), B * C), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, b, w, out, P);
    return camli_check_launch("camli_sk_mix_fwd");
}

extern "C" int camli_sk_mix_bwd_w(const float* g, const float* a, const float* b, float* gw, int B, int C, int P,
                                  void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!g || !a || !b || !gw) { camli_set_error("camli_sk_mix_bwd_w: null pointer"); return CAMLI_EINVAL; }
    if (!sk_args_ok("camli_sk_mix_bwd_w", B, C, P)) return CAMLI_EINVAL;
    hipLaunchKernelGGL(sk_mix_bwd_w_kernel, dim3(B * C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, a, b, gw, P);
    return camli_check_launch("camli_sk_mix_bwd_w");
}

extern "C" int camli_sk_mix_bwd_x(const float* g, const float* w, const float* gs, float* ga, float* gb, int B, int C,
                                  int P, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!g || !w || !ga || !gb) { camli_set_error("camli_sk_mix_bwd_x: null pointer"); return CAMLI_EINVAL; }
    if (!sk_args_ok("camli_sk_mix_bwd_x", B, C, P) || B * C > 65535) {
        camli_set_error("camli_sk_mix_bwd_x: bad shape B=%d C=%d P=%d", B, C, P);
        return CAMLI_EINVAL;
    }
    hipLaunchKernelGGL(sk_mix_bwd_x_kernel, dim3(camli_divup(P, 1024), B * C), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), g, w, gs, ga, gb, P);
    return camli_check_launch("camli_sk_mix_bwd_x");
}

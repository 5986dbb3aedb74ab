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


// ---- the gate: w = softmax_pairs(sigmoid(relu(s Wmid^T) Wout^T)) on [B,C] vectors (clfm.py:183-184,
// 199-203: two bias-free Linear layers, ReLU, Sigmoid, reshape [B,C,2], softmax over the pair).  In torch
// this is ~5 launches forward and ~9 backward per SKFusion call, all on kilobyte-sized tensors. ----
constexpr int SKG_MAXC = 1024, SKG_MAXR = 512;   // CamLiPWC fuses a 627-channel correlation feature (clfm.py:171-214 at configs[1])

// The gate is two tiny matrix-vector products per batch row; round 2/3a ran them as one serial dot product per thread
// (64 of 256 threads busy, every step a dependent global load: 18 us forward, 32 us for the s-adjoint, 54 calls per step).
// Now a workgroup of 1024 threads per batch row and two access-shaped helpers:
//   rows form : y[i] = sum_j W[i*ld + j] x[j]   (the contraction runs along the contiguous dimension) -- a group of G
//               lanes per row, lanes stride over j (coalesced), butterfly sum inside the group
//   cols form : y[j] = sum_i W[i*ld + j] x[i]   (the OUTPUT index is the contiguous one) -- lanes along j, the i range cut
//               into NT / J parts whose partial sums meet in LDS, added in part order
// Both are fixed-order sums (bit-reproducible); x lives in LDS.
constexpr int SKG_NT = 1024;

template <typename F>
__device__ __forceinline__ void skg_rows(const float* __restrict__ W, int ld, const float* xs, int I, int J, F&& emit) {
    // short rows take small groups (more rows in flight, fewer butterfly steps: a step is an LDS-crossbar round trip)
    const int G = J > 128 ? 64 : (J > 64 ? 32 : 16);
    const int lane = threadIdx.x & (G - 1), group = threadIdx.x / G, ngroups = SKG_NT / G;
    for (int i = group; i < I; i += 2 * ngroups) {           // two rows per trip: their loads and butterflies overlap
        const int i2 = i + ngroups;
        const bool two = i2 < I;
        const float* __restrict__ row = W + (size_t)i * ld;
        const float* __restrict__ row2 = W + (size_t)(two ? i2 : i) * ld;
        float a = 0.0f, a2 = 0.0f;
        for (int j = lane; j < J; j += G) {
            const float x = xs[j];
            a = __builtin_fmaf(row[j], x, a);
            a2 = __builtin_fmaf(row2[j], x, a2);
        }
        for (int off = G >> 1; off >= 1; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            a2 += __shfl_xor(a2, off, 64);
        }
        if (lane == 0) {
            emit(i, a);
            if (two) emit(i2, a2);
        }
    }
}

// part_buf: SKG_NT floats of LDS.  Ends with the results handed to emit(j, sum) by the threads j < J (J <= SKG_NT) or
// by thread j % SKG_NT (J > SKG_NT: one part, no LDS); the caller synchronises afterwards.
template <typename F>
__device__ __forceinline__ void skg_cols(const float* __restrict__ W, int ld, const float* xs, int I, int J, float* part_buf,
                                         F&& emit) {
    const int tid = threadIdx.x;
    if (J >= SKG_NT) {
        for (int j = tid; j < J; j += SKG_NT) {
            float a = 0.0f;
            for (int i = 0; i < I; ++i) a = __builtin_fmaf(W[(size_t)i * ld + j], xs[i], a);
            emit(j, a);
        }
        return;
    }
    const int parts = SKG_NT / J, part = tid / J, j = tid - part * J;
    float a = 0.0f;
    if (part < parts) {
        const int per = (I + parts - 1) / parts, lo = part * per, hi = min(I, lo + per);
#pragma unroll 8
        for (int i = lo; i < hi; ++i) a = __builtin_fmaf(W[(size_t)i * ld + j], xs[i], a);
        part_buf[part * J + j] = a;
    }
    __syncthreads();
    if (tid < J) {
        float t = part_buf[tid];
        for (int q = 1; q < parts; ++q) t += part_buf[q * J + tid];
        emit(tid, t);
    }
}

// grid B, block 1024
__global__ __launch_bounds__(SKG_NT) void sk_gate_fwd_kernel(const float* __restrict__ s, const float* __restrict__ wmid,
                                                             const float* __restrict__ wout, float* __restrict__ m,
                                                             float* __restrict__ z, float* __restrict__ w, int C, int R) {
    __shared__ float ss[SKG_MAXC], sm[SKG_MAXR], sz[2 * SKG_MAXC];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += SKG_NT) ss[c] = s[(size_t)b * C + c];
    __syncthreads();
    skg_rows(wmid, C, ss, R, C, [&](int r, float a) {          // m = relu(Wmid s)
        a = fmaxf(a, 0.0f);
        sm[r] = a;
        m[(size_t)b * R + r] = a;
    });
    __syncthreads();
    skg_rows(wout, R, sm, 2 * C, R, [&](int o, float a) {      // z = sigmoid(Wout m)
        a = 1.0f / (1.0f + __expf(-a));
        sz[o] = a;
        z[(size_t)b * 2 * C + o] = a;
    });
    __syncthreads();
    for (int c = tid; c < C; c += SKG_NT) {
        const float z0 = sz[2 * c], z1 = sz[2 * c + 1];
        const float mx = fmaxf(z0, z1);
        const float e0 = __expf(z0 - mx), e1 = __expf(z1 - mx);
        const float inv = 1.0f / (e0 + e1);
        w[((size_t)b * C + c) * 2] = e0 * inv;
        w[((size_t)b * C + c) * 2 + 1] = e1 * inv;
    }
}

// Gate adjoint, two kernels, no atomics (round 3; round 2 ran ONE workgroup per batch row that pushed 2*C*R + R*C float
// atomics each: 8 workgroups on 256 CUs, 50 us per call, 54 calls per step, and a non-reproducible sum):
//   sk_gate_bwd_s_kernel  grid B: gpre (softmax pair + sigmoid adjoint), gm = relu'(m) * Wout^T gpre, gs = Wmid^T gm
//   sk_gate_bwd_w_kernel  grid R: hidden unit r recomputes gpre[b,:] and gm[b,r] for every b (a few hundred flops) and
//                         ADDS  gWout[:,r] += sum_b gpre[b,:] m[b,r],  gWmid[r,:] += sum_b gm[b,r] s[b,:]  in batch order
// (plain read-modify-write: the accumulators belong to one stream; the callers pass zero-filled or running buffers).
__device__ __forceinline__ float sk_gpre(const float* __restrict__ gw, const float* __restrict__ w,
                                         const float* __restrict__ z, int b, int o, int C) {
    const int c = o >> 1;
    const size_t e = ((size_t)b * C + c) * 2;
    const float w0 = w[e], w1 = w[e + 1], g0 = gw[e], g1 = gw[e + 1];
    const float dot = g0 * w0 + g1 * w1;                       // softmax over the pair
    const float zz = z[(size_t)b * 2 * C + o];
    const float wo = (o & 1) ? w1 : w0, go = (o & 1) ? g1 : g0;
    return wo * (go - dot) * zz * (1.0f - zz);                 // ... then the sigmoid
}

__global__ __launch_bounds__(SKG_NT) void sk_gate_bwd_s_kernel(const float* __restrict__ gw, const float* __restrict__ m,
                                                               const float* __restrict__ z, const float* __restrict__ w,
                                                               const float* __restrict__ wmid, const float* __restrict__ wout,
                                                               float* __restrict__ gs, int C, int R) {
    __shared__ float gp[2 * SKG_MAXC], gm[SKG_MAXR], part_buf[SKG_NT];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int o = tid; o < 2 * C; o += SKG_NT) gp[o] = sk_gpre(gw, w, z, b, o, C);
    __syncthreads();
    skg_cols(wout, R, gp, 2 * C, R, part_buf, [&](int r, float a) {       // gm = Wout^T gpre, through the ReLU
        gm[r] = m[(size_t)b * R + r] > 0.0f ? a : 0.0f;
    });
    __syncthreads();
    skg_cols(wmid, C, gm, R, C, part_buf, [&](int c, float a) {           // gs = Wmid^T gm
        gs[(size_t)b * C + c] = a;
    });
}

__global__ __launch_bounds__(256) void sk_gate_bwd_w_kernel(const float* __restrict__ gw, const float* __restrict__ s,
                                                            const float* __restrict__ m, const float* __restrict__ z,
                                                            const float* __restrict__ w, const float* __restrict__ wout,
                                                            float* __restrict__ gwmid, float* __restrict__ gwout, int B,
                                                            int C, int R) {
    __shared__ float red[4];
    __shared__ float gm_b;
    const int tid = threadIdx.x, r = blockIdx.x;
    constexpr int OPT = 2 * SKG_MAXC / 256;      // output units per thread
    float acc_out[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) acc_out[i] = 0.0f;
    constexpr int CPT = SKG_MAXC / 256;
    float acc_mid[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) acc_mid[i] = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float mb = m[(size_t)b * R + r];
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < OPT; ++i) {
            const int o = tid + 256 * i;
            if (o < 2 * C) {
                const float g = sk_gpre(gw, w, z, b, o, C);
                acc_out[i] = __builtin_fmaf(g, mb, acc_out[i]);
                part = __builtin_fmaf(wout[o * R + r], g, part);
            }
        }
        const float tot = block_sum_256(part, red);
        if (tid == 0) gm_b = mb > 0.0f ? tot : 0.0f;
        __syncthreads();
        const float gmv = gm_b;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            if (c < C) acc_mid[i] = __builtin_fmaf(gmv, s[(size_t)b * C + c], acc_mid[i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const int o = tid + 256 * i;
        if (o < 2 * C) gwout[o * R + r] += acc_out[i];
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + 256 * i;
        if (c < C) gwmid[r * C + c] += acc_mid[i];
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

extern "C" int camli_sk_gate_fwd(const float* s, const float* wmid, const float* wout, float* m, float* z, float* w,
                                 int B, int C, int R, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!s || !wmid || !wout || !m || !z || !w) { camli_set_error("camli_sk_gate_fwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || R < 1 || C > SKG_MAXC || R > SKG_MAXR) {
        camli_set_error("camli_sk_gate_fwd: bad shape B=%d C=%d (<= %d) R=%d (<= %d)", B, C, SKG_MAXC, R, SKG_MAXR);
        return C > SKG_MAXC || R > SKG_MAXR ? CAMLI_ENOTSUP : CAMLI_EINVAL;
    }
    hipLaunchKernelGGL(sk_gate_fwd_kernel, dim3(B), dim3(SKG_NT), 0, reinterpret_cast<hipStream_t>(stream), s, wmid, wout, m, z,
                       w, C, R);
    return camli_check_launch("camli_sk_gate_fwd");
}

extern "C" int camli_sk_gate_bwd(const float* gw, const float* s, const float* m, const float* z, const float* w,
                                 const float* wmid, const float* wout, float* gs, float* gwmid, float* gwout, int B,
                                 int C, int R, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gw || !s || !m || !z || !w || !wmid || !wout || !gs || !gwmid || !gwout) {
        camli_set_error("camli_sk_gate_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || R < 1 || C > SKG_MAXC || R > SKG_MAXR) {
        camli_set_error("camli_sk_gate_bwd: bad shape B=%d C=%d (<= %d) R=%d (<= %d)", B, C, SKG_MAXC, R, SKG_MAXR);
        return C > SKG_MAXC || R > SKG_MAXR ? CAMLI_ENOTSUP : CAMLI_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sk_gate_bwd_s_kernel, dim3(B), dim3(SKG_NT), 0, st, gw, m, z, w, wmid, wout, gs, C, R);
    hipLaunchKernelGGL(sk_gate_bwd_w_kernel, dim3(R), dim3(256), 0, st, gw, s, m, z, w, wout, gwmid, gwout, B, C, R);
    return camli_check_launch("camli_sk_gate_bwd");
}

// Bias + activation epilogue of the 1x1 / kxk convolutions and its adjoint, gfx950.
//
// The reference's Conv{1,2}dNormRelu blocks (models/mlp.py:41-128) and the plain nn.Conv2d + ReLU
// pairs of models/raft_core.py run, per layer, conv -> broadcast bias add -> activation forward and
// activation-backward -> bias-gradient reduction backward: four elementwise passes plus a reduction
// over an activation-sized tensor.  Here the convolution runs without bias and
//   fwd : y = act(x + bias[c])                     in place, one read + one write
//   bwd : gx = gy * act'(y);  gbias[c] += sum gx   one pass; the per-channel sum is reduced in the
//         same kernel (wave shuffle -> LDS -> one float atomic per workgroup)
// act: 0 identity, 1 relu, 2 leaky_relu(0.1), 3 sigmoid, 4 tanh, 5 relu followed by torch.nan_to_num (round 3: the motion
// encoder's last convolution, models/raft_core.py:163-164 -- NaN -> 0, +inf -> FLT_MAX, and a zero gradient wherever the
// pre-activation was not finite; sign-mask form only).  Layout [B, C, P] (P = spatial size).
#include "camli_common.h"

namespace {

template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
    if (ACT == 1) return v > 0.0f ? v : 0.0f;
    if (ACT == 2) return v > 0.0f ? v : 0.1f * v;
    if (ACT == 3) return 1.0f / (1.0f + __expf(-v));
    if (ACT == 4) return tanhf(v);
    if (ACT == 5) return v > 0.0f ? fminf(v, 3.402823466e+38f) : 0.0f;     // NaN fails v > 0: 0, as nan_to_num(relu(NaN))
    return v;
}

// derivative expressed through the OUTPUT y (what is kept for the backward)
template <int ACT>
__device__ __forceinline__ float act_grad(float y) {
    if (ACT == 1) return y > 0.0f ? 1.0f : 0.0f;
    if (ACT == 2) return y > 0.0f ? 1.0f : 0.1f;
    if (ACT == 3) return y * (1.0f - y);
    if (ACT == 4) return 1.0f - y * y;
    return 1.0f;
}

// Sign mask of the output (ACT 1 / 2 only): one bit per element, so that the backward reads 1/32 byte per
// element instead of y (4 bytes).  Word w of plane (b,c) holds, for the 64 float4 groups 64w .. 64w+63 of
// the plane, the ballots of their x, y, z, w components: mask[((b*C + c) * words(P) + w) * 4 + comp], bit =
// lane.  fwd and bwd walk the plane identically (chunks are multiples of 1024 elements), so a wave
// iteration always covers one whole word.
__host__ __device__ __forceinline__ int mask_words(int P) { return (P / 4 + 63) / 64; }

// grid (chunks, C, B), block 256; each block walks its chunk of one (b, c) plane
// RES (round 3): y = act(x + bias[c] + res) -- the closing statement of a residual block, relu(bn3(conv3(.)) + shortcut)
// (mmdet ResNet Bottleneck), in the same single pass; as bias pass + add + relu it was 7 tensor streams instead of 3.
template <int ACT, bool MASK, bool RES = false>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                            unsigned long long* __restrict__ mask, int C, int P,
                                                            int chunk, const float* __restrict__ res = nullptr,
                                                            float* out = nullptr, size_t out_bs = 0) {
    // out == nullptr: in place.  Otherwise the result goes to out[b * out_bs + c * P + .] -- a channel slice of a wider
    // tensor (the concatenation the next convolution reads: no cat pass), x is left untouched.
    const int c = blockIdx.y, b = blockIdx.z;
    const float bv = bias[c];
    float* plane = x + ((size_t)b * C + c) * P;
    float* oplane = out ? out + (size_t)b * out_bs + (size_t)c * P : plane;
    const float* __restrict__ rplane = RES ? res + ((size_t)b * C + c) * P : nullptr;
    const int beg = blockIdx.x * chunk, end = min(P, beg + chunk);
    if ((P & 3) == 0 && (chunk & 3) == 0) {
        float4* p4 = reinterpret_cast<float4*>(plane);
        float4* o4 = reinterpret_cast<float4*>(oplane);
        unsigned long long* __restrict__ mrow = MASK ? mask + ((size_t)b * C + c) * mask_words(P) * 4 : nullptr;
        const float4* __restrict__ r4 = reinterpret_cast<const float4*>(rplane);
        auto apply = [&](float4 v, int e) {
            float4 raw = make_float4(v.x + bv, v.y + bv, v.z + bv, v.w + bv);
            if (RES) {
                const float4 r = r4[e];
                raw.x += r.x; raw.y += r.y; raw.z += r.z; raw.w += r.w;
            }
            v.x = act_fwd<ACT>(raw.x); v.y = act_fwd<ACT>(raw.y); v.z = act_fwd<ACT>(raw.z); v.w = act_fwd<ACT>(raw.w);
            o4[e] = v;
            if (MASK) {
                // ACT 5: the gradient passes where the pre-activation is positive AND finite (relu, then nan_to_num)
                constexpr float top = 3.402823466e+38f;
                const unsigned long long mx = __ballot(ACT == 5 ? (raw.x > 0.0f && raw.x <= top) : v.x > 0.0f);
                const unsigned long long my = __ballot(ACT == 5 ? (raw.y > 0.0f && raw.y <= top) : v.y > 0.0f);
                const unsigned long long mz = __ballot(ACT == 5 ? (raw.z > 0.0f && raw.z <= top) : v.z > 0.0f);
                const unsigned long long mw = __ballot(ACT == 5 ? (raw.w > 0.0f && raw.w <= top) : v.w > 0.0f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* w = mrow + (size_t)(e >> 6) * 4;
                    w[0] = mx; w[1] = my; w[2] = mz; w[3] = mw;
                }
            }
        };
        // four 16-byte loads in flight per lane before the first store (the plane is updated in place, so the compiler
        // will not hoist a later load above an earlier store itself); the test is wave-uniform so that a mask word is
        // always produced by one full-wave ballot
        int e = beg / 4 + threadIdx.x;
        const int e4 = end / 4;
        for (; e - (int)(threadIdx.x & 63) + 63 + 768 < e4; e += 1024) {
            const float4 v0 = p4[e], v1 = p4[e + 256], v2 = p4[e + 512], v3 = p4[e + 768];
            apply(v0, e); apply(v1, e + 256); apply(v2, e + 512); apply(v3, e + 768);
        }
        for (; e < e4; e += 256) apply(p4[e], e);
    } else {
        for (int e = beg + threadIdx.x; e < end; e += 256) oplane[e] = act_fwd<ACT>(plane[e] + bv + (RES ? rplane[e] : 0.0f));
    }
}

// MASK: the activation derivative comes from the sign mask (ACT 1 / 2, P % 4 == 0) and y is not read
template <int ACT, bool MASK>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const unsigned long long* __restrict__ mask,
                                                            float* __restrict__ gx, float* __restrict__ gbias, int C,
                                                            int P, int chunk, size_t gy_bs) {
    // gy_bs: batch stride of gy in floats (C * P, or more when gy is a channel slice of a wider gradient -- the adjoint of a
    // cat hands those over, and copying them out first cost a pass per slice)
    __shared__ float partial[4];
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t base = ((size_t)b * C + c) * P;
    const size_t gbase = (size_t)b * gy_bs + (size_t)c * P;
    const int beg = blockIdx.x * chunk, end = min(P, beg + chunk);
    float acc = 0.0f;
    if ((P & 3) == 0 && (chunk & 3) == 0) {
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gy + gbase);
        float4* __restrict__ o4 = reinterpret_cast<float4*>(gx + base);
        if (MASK) {
            const unsigned long long* __restrict__ mrow = mask + ((size_t)b * C + c) * mask_words(P) * 4;
            const int lane = threadIdx.x & 63;
            constexpr float neg = ACT == 1 ? 0.0f : 0.1f;
            auto apply = [&](const float4 g, int e) {
                const unsigned long long* w = mrow + (size_t)(e >> 6) * 4;
                float4 o;
                o.x = g.x * (((w[0] >> lane) & 1ull) ? 1.0f : neg); o.y = g.y * (((w[1] >> lane) & 1ull) ? 1.0f : neg);
                o.z = g.z * (((w[2] >> lane) & 1ull) ? 1.0f : neg); o.w = g.w * (((w[3] >> lane) & 1ull) ? 1.0f : neg);
                o4[e] = o;
                acc += (o.x + o.y) + (o.z + o.w);
            };
            int e = beg / 4 + threadIdx.x;
            const int e4 = end / 4;
            for (; e + 768 < e4; e += 1024) {      // four loads in flight per lane
                const float4 g0 = g4[e], g1 = g4[e + 256], g2 = g4[e + 512], g3 = g4[e + 768];
                apply(g0, e); apply(g1, e + 256); apply(g2, e + 512); apply(g3, e + 768);
            }
            for (; e < e4; e += 256) apply(g4[e], e);
        } else {
            const float4* __restrict__ y4 = reinterpret_cast<const float4*>(y + base);
            auto apply = [&](const float4 g, const float4 yy, int e) {
                float4 o;
                o.x = g.x * act_grad<ACT>(yy.x); o.y = g.y * act_grad<ACT>(yy.y);
                o.z = g.z * act_grad<ACT>(yy.z); o.w = g.w * act_grad<ACT>(yy.w);
                o4[e] = o;
                acc += (o.x + o.y) + (o.z + o.w);
            };
            int e = beg / 4 + threadIdx.x;
            const int e4 = end / 4;
            for (; e + 256 < e4; e += 512) {       // two (gy, y) pairs = four loads in flight per lane
                const float4 g0 = g4[e], y0 = y4[e], g1 = g4[e + 256], y1 = y4[e + 256];
                apply(g0, y0, e); apply(g1, y1, e + 256);
            }
            for (; e < e4; e += 256) apply(g4[e], y4[e], e);
        }
    } else {
        for (int e = beg + threadIdx.x; e < end; e += 256) {
            const float o = gy[gbase + e] * act_grad<ACT>(y[base + e]);
            gx[base + e] = o;
            acc += o;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(gbias + c, (partial[0] + partial[1]) + (partial[2] + partial[3]));
}

// Identity activation: gx == gy, so the adjoint only needs the per-channel sum of gy (4 B/elem read, nothing
// written but the C bias sums).  Same grid / reduction as the general kernel.
__global__ __launch_bounds__(256) void bias_sum_kernel(const float* __restrict__ gy, float* __restrict__ gbias, int C,
                                                        int P, int chunk, size_t gy_bs) {
    __shared__ float partial[4];
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t base = (size_t)b * gy_bs + (size_t)c * P;
    const int beg = blockIdx.x * chunk, end = min(P, beg + chunk);
    float acc = 0.0f;
    if ((P & 3) == 0 && (chunk & 3) == 0) {
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gy + base);
        int e = beg / 4 + threadIdx.x;
        const int e4 = end / 4;
        for (; e + 768 < e4; e += 1024) {
            const float4 a = g4[e], b4 = g4[e + 256], c4 = g4[e + 512], d4 = g4[e + 768];
            acc += (a.x + a.y) + (a.z + a.w);
            acc += (b4.x + b4.y) + (b4.z + b4.w);
            acc += (c4.x + c4.y) + (c4.z + c4.w);
            acc += (d4.x + d4.y) + (d4.z + d4.w);
        }
        for (; e < e4; e += 256) {
            const float4 g = g4[e];
            acc += (g.x + g.y) + (g.z + g.w);
        }
    } else {
        for (int e = beg + threadIdx.x; e < end; e += 256) acc += gy[base + e];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(gbias + c, (partial[0] + partial[1]) + (partial[2] + partial[3]));
}

// ---------------------------------------------------------------------------------------------------------------------
// channels-last forms (round 3).  The ResNet trunk runs channels-last since MIOpen's implicit-GEMM kernels are NHWC
// kernels: fed NCHW tensors it wraps them in layout transposes (15 ms per step) and, for the trunk's shapes, picks slower
// solvers -- profiles/r03_trunk_conv_layout_microbench.txt: 22.1 -> 18.1 ms per forward+backward of the trunk at batch 16.
// Tensor = [n_pix, C] with C fastest, C a power of two in [4, 1024]: a thread's 16-byte group always holds the same
// four channels (the grid stride is a multiple of C), so its bias quad is loaded once and its bias-gradient partials
// stay in registers.  Sign mask: word i holds the ballots of float4 groups 64 i .. 64 i + 63 (components x, y, z, w in
// four consecutive words) of the FLAT tensor; forward and adjoint walk it identically.
// ---------------------------------------------------------------------------------------------------------------------
template <int ACT, bool MASK, bool RES>
__global__ __launch_bounds__(256) void bias_act_nhwc_fwd_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                                 const float* __restrict__ res,
                                                                 unsigned long long* __restrict__ mask, size_t n4, int C) {
    float4* __restrict__ p4 = reinterpret_cast<float4*>(x);
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(res);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 bv = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % (size_t)C));
    // four groups per trip, their loads requested together (one at a time the 535 MB activations of the first stage are a
    // chain of 32 dependent load -> store round trips per thread); whole waves take a trip together: ballots complete
    for (; i - (threadIdx.x & 63) < n4; i += 4 * stride) {
        float4 v[4], r[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = i + u * stride;
            in[u] = e < n4;
            v[u] = in[u] ? p4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (RES) r[u] = in[u] ? r4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = i + u * stride;
            if (e - (threadIdx.x & 63) >= n4) break;          // wave-uniform: this wave has no group in trip u
            float4 raw = make_float4(v[u].x + bv.x, v[u].y + bv.y, v[u].z + bv.z, v[u].w + bv.w);
            if (RES) { raw.x += r[u].x; raw.y += r[u].y; raw.z += r[u].z; raw.w += r[u].w; }
            float4 o;
            o.x = act_fwd<ACT>(raw.x); o.y = act_fwd<ACT>(raw.y); o.z = act_fwd<ACT>(raw.z); o.w = act_fwd<ACT>(raw.w);
            if (in[u]) p4[e] = o;
            if (MASK) {
                const unsigned long long mx = __ballot(in[u] && o.x > 0.0f), my = __ballot(in[u] && o.y > 0.0f);
                const unsigned long long mz = __ballot(in[u] && o.z > 0.0f), mw = __ballot(in[u] && o.w > 0.0f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* w = mask + (e >> 6) * 4;
                    w[0] = mx; w[1] = my; w[2] = mz; w[3] = mw;
                }
            }
        }
    }
}

// adjoint: gx = gy * [mask bit] (RELU) or gx aliases gy (identity: gx == nullptr, only the sums); gbias[c] += sum gx
template <bool RELU>
__global__ __launch_bounds__(256) void bias_act_nhwc_bwd_kernel(const float* __restrict__ gy,
                                                                 const unsigned long long* __restrict__ mask,
                                                                 float* __restrict__ gx, float* __restrict__ gbias, size_t n4,
                                                                 int C, float* __restrict__ partials) {
    __shared__ float4 part[256];
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gy);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(gx);
    const size_t stride = (size_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = i + u * stride < n4 ? g4[i + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = i + u * stride;
            if (e >= n4) break;
            if (RELU) {
                const unsigned long long* w = mask + (e >> 6) * 4;
                g[u].x = ((w[0] >> lane) & 1ull) ? g[u].x : 0.0f; g[u].y = ((w[1] >> lane) & 1ull) ? g[u].y : 0.0f;
                g[u].z = ((w[2] >> lane) & 1ull) ? g[u].z : 0.0f; g[u].w = ((w[3] >> lane) & 1ull) ? g[u].w : 0.0f;
                o4[e] = g[u];
            }
            acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w;
        }
    }
    // threads tid, tid + C/4, tid + 2 C/4 ... of the block hold the same four channels
    part[threadIdx.x] = acc;
    __syncthreads();
    const int q = C >> 2;                     // channel quads; 256 % q == 0
    if ((int)threadIdx.x < q) {
        float4 s = part[threadIdx.x];
        for (int t = threadIdx.x + q; t < 256; t += q) {
            const float4 o = part[t];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        // the block's first float4 index is a multiple of 256, hence of q: thread t holds quad t
        if (partials) {      // [gridDim.x][C]: summed in block order by bias_nhwc_reduce_kernel (no atomics, reproducible)
            *reinterpret_cast<float4*>(partials + (size_t)blockIdx.x * C + 4 * threadIdx.x) = s;
        } else {
            float* dst = gbias + 4 * threadIdx.x;
            unsafeAtomicAdd(dst + 0, s.x); unsafeAtomicAdd(dst + 1, s.y); unsafeAtomicAdd(dst + 2, s.z); unsafeAtomicAdd(dst + 3, s.w);
        }
    }
}

// gbias[c] += sum over the blocks' partial rows.  grid ceil(C/64), block (64, 16)
__global__ __launch_bounds__(1024) void bias_nhwc_reduce_kernel(const float* __restrict__ partials, int n_rows, int C,
                                                                 float* __restrict__ gbias) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    float acc = 0.0f;
    if (c < C) {
#pragma unroll 8
        for (int r = threadIdx.y; r < n_rows; r += 16) acc += partials[(size_t)r * C + c];
    }
    red[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && c < C) {
        float v = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[g][threadIdx.x];
        gbias[c] += v;
    }
}

bool nhwc_ok(const char* what, long long n_pix, int C, int act) {
    if (n_pix < 0 || C < 4 || C > 1024 || (C & (C - 1)) || (act != 0 && act != 1)) {
        camli_set_error("%s: channels-last form needs C a power of two in [4, 1024] and act 0 or 1 (n_pix=%lld C=%d act=%d)", what,
                        n_pix, C, act);
        return false;
    }
    return true;
}

int nhwc_blocks(size_t n4) {
    const size_t want = (n4 + 256 * 8 - 1) / (256 * 8);      // ~8 groups per thread, at most 16384 workgroups
    return (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
}

int pick_chunk(int P) {
    // ~8K elements per block, multiple of 1024 so float4 groups never straddle chunks
    return P <= 8192 ? ((P + 1023) / 1024) * 1024 : 8192;
}

bool shape_ok(const char* what, int B, int C, int P, int act) {
    if (B < 0 || C < 1 || P < 1 || act < 0 || act > 5 || B > 65535 || C > 65535) {
        camli_set_error("%s: bad arguments B=%d C=%d P=%d act=%d", what, B, C, P, act);
        return false;
    }
    return true;
}

}  // namespace

extern "C" int64_t camli_bias_act_mask_bytes(int B, int C, int P) {
    if (B < 0 || C < 1 || P < 4 || (P & 3)) return 0;     // 0: no mask form for this shape (scalar path)
    return (int64_t)B * C * mask_words(P) * 4 * (int64_t)sizeof(unsigned long long);
}

extern "C" int camli_bias_act_fwd(float* x_inout, const float* bias, void* sign_mask, int B, int C, int P, int act,
                                  void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!x_inout || !bias) { camli_set_error("camli_bias_act_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!shape_ok("camli_bias_act_fwd", B, C, P, act)) return CAMLI_EINVAL;
    if ((sign_mask && !((act == 1 || act == 2 || act == 5) && (P & 3) == 0)) || (act == 5 && !sign_mask)) {
        camli_set_error("camli_bias_act_fwd: a sign mask needs act 1, 2 or 5 and P %% 4 == 0, act 5 needs the mask (act=%d P=%d)", act, P);
        return CAMLI_EINVAL;
    }
    const int chunk = pick_chunk(P);
    dim3 grid(camli_divup(P, chunk), C, B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* m = static_cast<unsigned long long*>(sign_mask);
#define L(A, M) hipLaunchKernelGGL((bias_act_fwd_kernel<A, M>), grid, dim3(256), 0, s, x_inout, bias, m, C, P, chunk)
    switch (act) {
        case 0: L(0, false); break;
        case 1: if (m) L(1, true); else L(1, false); break;
        case 2: if (m) L(2, true); else L(2, false); break;
        case 3: L(3, false); break;
        case 5: L(5, true); break;
        default: L(4, false); break;
    }
#undef L
    return camli_check_launch("camli_bias_act_fwd");
}

// act(x + bias[c]) written to a channel slice of a wider tensor: out[b * out_batch_stride + c * P + p]; x [B,C,P] is only read.
// Sign mask as camli_bias_act_fwd (indexed by the dense (b, c, p)).  The adjoint is camli_bias_act_bwd_strided on the slice
// of the wider gradient.
extern "C" int camli_bias_act_into_fwd(const float* x, const float* bias, void* sign_mask, float* out, int64_t out_batch_stride,
                                       int B, int C, int P, int act, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!x || !bias || !out) { camli_set_error("camli_bias_act_into_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!shape_ok("camli_bias_act_into_fwd", B, C, P, act)) return CAMLI_EINVAL;
    if ((sign_mask && !((act == 1 || act == 2 || act == 5) && (P & 3) == 0)) || (act == 5 && !sign_mask)) {
        camli_set_error("camli_bias_act_into_fwd: a sign mask needs act 1, 2 or 5 and P %% 4 == 0, act 5 needs the mask (act=%d P=%d)", act, P);
        return CAMLI_EINVAL;
    }
    if (out_batch_stride < (int64_t)C * P || ((P & 3) == 0 && ((out_batch_stride & 3) || (reinterpret_cast<uintptr_t>(out) & 15)))) {
        camli_set_error("camli_bias_act_into_fwd: batch stride %lld (needs >= C*P, with P %% 4 == 0 a multiple of 4 and a 16-byte aligned out)",
                        (long long)out_batch_stride);
        return CAMLI_EINVAL;
    }
    const int chunk = pick_chunk(P);
    dim3 grid(camli_divup(P, chunk), C, B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* m = static_cast<unsigned long long*>(sign_mask);
    float* xin = const_cast<float*>(x);
#define L(A, M) hipLaunchKernelGGL((bias_act_fwd_kernel<A, M>), grid, dim3(256), 0, s, xin, bias, m, C, P, chunk, nullptr, out, (size_t)out_batch_stride)
    switch (act) {
        case 0: L(0, false); break;
        case 1: if (m) L(1, true); else L(1, false); break;
        case 2: if (m) L(2, true); else L(2, false); break;
        case 3: L(3, false); break;
        case 5: L(5, true); break;
        default: L(4, false); break;
    }
#undef L
    return camli_check_launch("camli_bias_act_into_fwd");
}

extern "C" int camli_bias_act_res_fwd(float* x_inout, const float* bias, const float* res, void* sign_mask, int B, int C,
                                      int P, int act, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!x_inout || !bias || !res) { camli_set_error("camli_bias_act_res_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!shape_ok("camli_bias_act_res_fwd", B, C, P, act)) return CAMLI_EINVAL;
    if (!(act == 0 || act == 1) || (sign_mask && !(act == 1 && (P & 3) == 0)) ||
        ((P & 3) == 0 && ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(x_inout)) & 15))) {
        camli_set_error("camli_bias_act_res_fwd: act must be 0 or 1 (mask: act 1, P %% 4 == 0), tensors 16-byte aligned (act=%d P=%d)", act, P);
        return CAMLI_EINVAL;
    }
    const int chunk = pick_chunk(P);
    dim3 grid(camli_divup(P, chunk), C, B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* m = static_cast<unsigned long long*>(sign_mask);
#define L(A, M) hipLaunchKernelGGL((bias_act_fwd_kernel<A, M, true>), grid, dim3(256), 0, s, x_inout, bias, m, C, P, chunk, res)
    if (act == 0) L(0, false);
    else if (m) L(1, true);
    else L(1, false);
#undef L
    return camli_check_launch("camli_bias_act_res_fwd");
}

extern "C" int64_t camli_bias_act_nhwc_mask_bytes(long long n_pix, int C) {
    const long long n4 = n_pix * C / 4;
    return ((n4 + 63) / 64) * 4 * (int64_t)sizeof(unsigned long long);
}

extern "C" int camli_bias_act_nhwc_fwd(float* x_inout, const float* bias, const float* res, void* sign_mask, long long n_pix,
                                       int C, int act, void* stream) {
    if (n_pix == 0) return CAMLI_OK;
    if (!x_inout || !bias) { camli_set_error("camli_bias_act_nhwc_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!nhwc_ok("camli_bias_act_nhwc_fwd", n_pix, C, act)) return CAMLI_EINVAL;
    if ((sign_mask && act != 1) || ((reinterpret_cast<uintptr_t>(x_inout) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(bias)) & 15)) {
        camli_set_error("camli_bias_act_nhwc_fwd: a sign mask needs act 1; pointers must be 16-byte aligned");
        return CAMLI_EINVAL;
    }
    const size_t n4 = (size_t)n_pix * C / 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* m = static_cast<unsigned long long*>(sign_mask);
    const dim3 grid(nhwc_blocks(n4));
#define L(A, M, R) hipLaunchKernelGGL((bias_act_nhwc_fwd_kernel<A, M, R>), grid, dim3(256), 0, s, x_inout, bias, res, m, n4, C)
    if (act == 0) { if (res) L(0, false, true); else L(0, false, false); }
    else if (m) { if (res) L(1, true, true); else L(1, true, false); }
    else { if (res) L(1, false, true); else L(1, false, false); }
#undef L
    return camli_check_launch("camli_bias_act_nhwc_fwd");
}

static int nhwc_bwd_blocks(size_t n4, bool partial_rows) {
    const int want = nhwc_blocks(n4), cap = partial_rows ? 4096 : 1024;
    return want > cap ? cap : want;
}

extern "C" int64_t camli_bias_act_nhwc_bwd_workspace_bytes(long long n_pix, int C) {
    if (n_pix < 0 || C < 4) return 0;
    return (int64_t)nhwc_bwd_blocks((size_t)n_pix * C / 4, true) * C * (int64_t)sizeof(float);
}

extern "C" int camli_bias_act_nhwc_bwd(const float* gy, const void* sign_mask, float* gx, float* gbias, float* workspace,
                                       long long n_pix, int C, int act, void* stream) {
    if (n_pix == 0) return CAMLI_OK;
    if (!gy || !gbias || (act == 1 && (!sign_mask || !gx))) { camli_set_error("camli_bias_act_nhwc_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!nhwc_ok("camli_bias_act_nhwc_bwd", n_pix, C, act)) return CAMLI_EINVAL;
    const size_t n4 = (size_t)n_pix * C / 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // with `workspace` (camli_bias_act_nhwc_bwd_workspace_bytes) every workgroup writes its C partial sums and a second
    // kernel adds them in block order; without it the workgroups end in C float atomics each on the same C addresses
    // (16,384 workgroups: 1.6 ms on a 535 MB activation, 7x its forward), so their number is capped at 1,024
    const int blocks = nhwc_bwd_blocks(n4, workspace != nullptr);
    const dim3 grid(blocks);
    if (act == 1)
        hipLaunchKernelGGL(bias_act_nhwc_bwd_kernel<true>, grid, dim3(256), 0, s, gy, static_cast<const unsigned long long*>(sign_mask),
                           gx, gbias, n4, C, workspace);
    else
        hipLaunchKernelGGL(bias_act_nhwc_bwd_kernel<false>, grid, dim3(256), 0, s, gy, nullptr, nullptr, gbias, n4, C, workspace);
    if (workspace)
        hipLaunchKernelGGL(bias_nhwc_reduce_kernel, dim3(camli_divup(C, 64)), dim3(64, 16), 0, s, workspace, blocks, C, gbias);
    return camli_check_launch("camli_bias_act_nhwc_bwd");
}

extern "C" int camli_bias_act_bwd_strided(const float* gy, int64_t gy_batch_stride, const float* y, const void* sign_mask,
                                          float* gx, float* gbias, int B, int C, int P, int act, void* stream);

extern "C" int camli_bias_act_bwd(const float* gy, const float* y, const void* sign_mask, float* gx, float* gbias,
                                  int B, int C, int P, int act, void* stream) {
    return camli_bias_act_bwd_strided(gy, (int64_t)C * P, y, sign_mask, gx, gbias, B, C, P, act, stream);
}

// gy [B,C,P] with batch stride gy_batch_stride floats (>= C*P): a channel slice of a wider gradient is read in place
extern "C" int camli_bias_act_bwd_strided(const float* gy, int64_t gy_batch_stride, const float* y, const void* sign_mask,
                                          float* gx, float* gbias, int B, int C, int P, int act, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!shape_ok("camli_bias_act_bwd", B, C, P, act)) return CAMLI_EINVAL;
    if (gy_batch_stride < (int64_t)C * P || ((P & 3) == 0 && ((gy_batch_stride & 3) || (reinterpret_cast<uintptr_t>(gy) & 15)))) {
        camli_set_error("camli_bias_act_bwd: batch stride %lld (needs >= C*P = %lld, and with P %% 4 == 0 a multiple of 4 and a 16-byte aligned pointer)",
                        (long long)gy_batch_stride, (long long)C * P);
        return CAMLI_EINVAL;
    }
    if (act == 0 && !gx) {   // identity: the caller aliases gx to gy, only the bias sums are produced
        if (!gy || !gbias) { camli_set_error("camli_bias_act_bwd: null pointer"); return CAMLI_EINVAL; }
        const int chunk0 = pick_chunk(P);
        hipLaunchKernelGGL(bias_sum_kernel, dim3(camli_divup(P, chunk0), C, B), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), gy, gbias, C, P, chunk0, (size_t)gy_batch_stride);
        return camli_check_launch("camli_bias_act_bwd(identity)");
    }
    if (!gy || (!y && !sign_mask && act != 0) || !gx || !gbias) { camli_set_error("camli_bias_act_bwd: null pointer"); return CAMLI_EINVAL; }
    if ((sign_mask && !((act == 1 || act == 2 || act == 5) && (P & 3) == 0)) || (act == 5 && !sign_mask)) {
        camli_set_error("camli_bias_act_bwd: a sign mask needs act 1, 2 or 5 and P %% 4 == 0, act 5 needs the mask (act=%d P=%d)", act, P);
        return CAMLI_EINVAL;
    }
    const int chunk = pick_chunk(P);
    dim3 grid(camli_divup(P, chunk), C, B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned long long* m = static_cast<const unsigned long long*>(sign_mask);
#define L(A, M) hipLaunchKernelGGL((bias_act_bwd_kernel<A, M>), grid, dim3(256), 0, s, gy, y, m, gx, gbias, C, P, chunk, (size_t)gy_batch_stride)
    switch (act) {
        case 0: L(0, false); break;
        case 1: if (m) L(1, true); else L(1, false); break;
        case 2: if (m) L(2, true); else L(2, false); break;
        case 3: L(3, false); break;
        case 5: L(1, true); break;       // the mask already holds "positive and finite": plain masked ReLU adjoint
        default: L(4, false); break;
    }
#undef L
    return camli_check_launch("camli_bias_act_bwd");
}

// Cost MLP of the RAFT-style point cost-volume lookup and the sum over the neighbours, gfx950.
//
// Replaces, in Correlation3D.forward of the reference (models/camliraft_l_core.py:62-101),
//     cost = self.cost_mlp(lookup).sum(dim=-1)         lookup [B,4,N,k] = (dxyz, cost-volume entry) per neighbour,
//                                                       cost_mlp = MLP2d(4 -> 32 -> 32, bias, ReLU), per level
// which on [B,4,N,4k] columns (the four levels side by side, k = 16) is 2 GEMMs + 2 bias/ReLU passes over two
// [B,32,N,64] tensors (134 MB each at batch 8) + a reduction pass forward, and about twice that backward.  Here the
// hidden activations never leave the registers:
//
//   fwd : out[b, l*32 + o, n] = sum_j relu(b2[o] + W2[o,:] . relu(b1 + W1 x[b,:,n,l*16+j]))
//         lane = column (level l = lane / 16, neighbour j = lane % 16) -> a wave is one point; the weights are
//         wave-uniform (scalar loads), the neighbour sum is a 16-lane DPP rotation sum, two points share a weight row
//   bwd : recomputes both layers from the 16-byte-per-column input, gives d/d(cost-volume entry) (the coordinates
//         are not differentiable on this path) and the parameter gradients.  dW2 = G2 H1^T and [dW1 | db1] =
//         G1 [X | 1]^T are contractions over the ~1 M columns of a call: they run on the matrix cores
//         (v_mfma_f32_32x32x2_f32, operands transposed through LDS, one 32x32 accumulator tile each per wave),
//         per-workgroup partial tiles are added in block order by a second kernel -- no atomics, bit-reproducible.
//
// Bounds: VALU (2 x 1152 fmaf per column forward, ~2300 backward) -- 33.5 M columns x ... at batch 8; HBM traffic is
// the 16.8 MB input + 8.4 MB output (forward), against ~0.8 GB for the unfused chain.
#include "camli_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CM_H = 32;                         // hidden width = output width of the cost MLP
constexpr int CM_COLS = 64;                      // columns per point: 4 levels x 16 neighbours = one wave
constexpr int CM_OUT = 4 * CM_H;                 // output channels (level-major)
constexpr int CM_PART = CM_H * CM_H + CM_H * 4 + CM_H + CM_H;   // dW2 | dW1 | db1 | db2 = 1216 floats per workgroup

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
// every lane of a 16-lane row receives the row's sum (row_ror:8, 4, 2, 1)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0x128>(v);
    v += dpp_f32<0x124>(v);
    v += dpp_f32<0x122>(v);
    v += dpp_f32<0x121>(v);
    return v;
}

__device__ __forceinline__ void layer1(const float* __restrict__ w1, const float* __restrict__ b1, const float (&x)[4],
                                       float (&h1)[CM_H]) {
#pragma unroll
    for (int i = 0; i < CM_H; ++i) {
        float a = b1[i];
        a = __builtin_fmaf(w1[i * 4 + 0], x[0], a);
        a = __builtin_fmaf(w1[i * 4 + 1], x[1], a);
        a = __builtin_fmaf(w1[i * 4 + 2], x[2], a);
        a = __builtin_fmaf(w1[i * 4 + 3], x[3], a);
        h1[i] = fmaxf(a, 0.0f);
    }
}

// grid ceil(B * N / (4 * CH)), block 256: wave = CH consecutive points of one batch element (N % CH == 0)
template <int CH>
__global__ __launch_bounds__(256) void corr3d_mlp_fwd_kernel(const float* __restrict__ lookup, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float* __restrict__ out, int B,
                                                             int N) {
    __shared__ float stage[4][CM_OUT][CH + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunks_per_b = N / CH;
    const int chunk = blockIdx.x * 4 + wv;
    const bool live = chunk < B * chunks_per_b;
    const int b = live ? chunk / chunks_per_b : 0, n0 = live ? (chunk % chunks_per_b) * CH : 0;
    const int l = lane >> 4, j = lane & 15;
    const float* __restrict__ xin = lookup + ((size_t)b * 4 * N + n0) * CM_COLS + lane;
    const size_t plane = (size_t)N * CM_COLS;

    for (int p = 0; p < CH; p += 2) {
        // the weights are re-fetched through the scalar cache per point pair: hoisted out of this loop the 1024 + 160
        // values would have to live in vector registers
        const float* __restrict__ w2p = w2;
        asm volatile("" : "+s"(w2p));
        float x0[4], x1[4], h0[CM_H], h1[CM_H];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x0[c] = xin[c * plane + (size_t)p * CM_COLS];
            x1[c] = xin[c * plane + (size_t)(p + 1) * CM_COLS];
        }
        layer1(w1, b1, x0, h0);
        layer1(w1, b1, x1, h1);
#pragma unroll
        for (int o = 0; o < CM_H; ++o) {
            float a0 = b2[o], a1 = a0;
#pragma unroll
            for (int i = 0; i < CM_H; ++i) {
                const float w = w2p[o * CM_H + i];
                a0 = __builtin_fmaf(w, h0[i], a0);
                a1 = __builtin_fmaf(w, h1[i], a1);
            }
            a0 = row16_sum(fmaxf(a0, 0.0f));
            a1 = row16_sum(fmaxf(a1, 0.0f));
            if (j == (o & 15)) {
                stage[wv][l * CM_H + o][p] = a0;
                stage[wv][l * CM_H + o][p + 1] = a1;
            }
        }
    }
    __syncthreads();
    if (!live) return;
    // out[b, ch, n0 .. n0+CH): CH consecutive floats per channel
    for (int e = lane; e < CM_OUT * (CH / 4); e += 64) {
        const int ch = e / (CH / 4), part = e - ch * (CH / 4);
        const float4 v = make_float4(stage[wv][ch][part * 4 + 0], stage[wv][ch][part * 4 + 1], stage[wv][ch][part * 4 + 2],
                                     stage[wv][ch][part * 4 + 3]);
        *reinterpret_cast<float4*>(out + ((size_t)b * CM_OUT + ch) * N + n0 + part * 4) = v;
    }
}

// grid ceil(B * N / (BW * CH)), block 64 * BW (BW = 2 waves: the three transposition buffers are 19 KB per wave).
// partials [gridDim.x][CM_PART]
constexpr int CM_BW = 2;
template <int CH>
__global__ __launch_bounds__(64 * CM_BW) void corr3d_mlp_bwd_kernel(const float* __restrict__ lookup, const float* __restrict__ gout,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ glookup, float* __restrict__ partials, int B,
                                                             int N) {
    constexpr int LD = CM_H + 1;
    __shared__ float bufA[CM_BW][CM_COLS][LD];     // G2^T, then G1^T: [column][channel]
    __shared__ float bufB[CM_BW][CM_COLS][LD];     // H1^T
    __shared__ float bufX[CM_BW][CM_COLS][8];      // x0..x3, 1, 0, 0, 0
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunks_per_b = N / CH;
    const int chunk = blockIdx.x * CM_BW + wv;
    const bool live = chunk < B * chunks_per_b;
    const int b = live ? chunk / chunks_per_b : 0, n0 = live ? (chunk % chunks_per_b) * CH : 0;
    const int l = lane >> 4;
    const int half = lane >> 5, cl = lane & 31;
    const size_t plane = (size_t)N * CM_COLS;
    const float* __restrict__ xin = lookup + ((size_t)b * 4 * N + n0) * CM_COLS + lane;
    const float* __restrict__ gin = gout + ((size_t)b * CM_OUT + l * CM_H) * N + n0;

    f32x16 acc_w2, acc_w1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_w2[r] = acc_w1[r] = 0.0f;
    float gb2 = 0.0f;       // lane (half, cl): sum of g2[o = cl] over the columns [32 half, 32 half + 32) of every point

    for (int p = 0; p < CH; ++p) {
        const float* __restrict__ w2p = w2;       // see the forward kernel
        asm volatile("" : "+s"(w2p));
        float x[4], h1[CM_H], gh1[CM_H];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = xin[c * plane + (size_t)p * CM_COLS];
        layer1(w1, b1, x, h1);
#pragma unroll
        for (int i = 0; i < CM_H; ++i) gh1[i] = 0.0f;
#pragma unroll
        for (int o = 0; o < CM_H; ++o) {
            float a = b2[o];
#pragma unroll
            for (int i = 0; i < CM_H; ++i) a = __builtin_fmaf(w2p[o * CM_H + i], h1[i], a);
            const float go = gin[(size_t)o * N + p];
            const float g = (live && a > 0.0f) ? go : 0.0f;
#pragma unroll
            for (int i = 0; i < CM_H; ++i) gh1[i] = __builtin_fmaf(w2p[o * CM_H + i], g, gh1[i]);
            bufA[wv][lane][o] = g;
        }
#pragma unroll
        for (int i = 0; i < CM_H; ++i) bufB[wv][lane][i] = h1[i];
        __syncthreads();
        // dW2[o][i] += sum over the 64 columns of g2[o] * h1[i]
#pragma unroll 8
        for (int s = 0; s < CM_COLS / 2; ++s)
            acc_w2 = __builtin_amdgcn_mfma_f32_32x32x2f32(bufA[wv][2 * s + half][cl], bufB[wv][2 * s + half][cl], acc_w2, 0, 0, 0);
#pragma unroll 8
        for (int s = 0; s < CM_COLS / 2; ++s) gb2 += bufA[wv][32 * half + s][cl];
        __syncthreads();
        float gx3 = 0.0f;
#pragma unroll
        for (int i = 0; i < CM_H; ++i) {
            const float g1 = h1[i] > 0.0f ? gh1[i] : 0.0f;
            bufA[wv][lane][i] = g1;
            gx3 = __builtin_fmaf(w1[i * 4 + 3], g1, gx3);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) bufX[wv][lane][c] = x[c];
        bufX[wv][lane][4] = 1.0f;
        bufX[wv][lane][5] = bufX[wv][lane][6] = bufX[wv][lane][7] = 0.0f;
        if (live) glookup[((size_t)b * 4 + 3) * plane + (size_t)(n0 + p) * CM_COLS + lane] = gx3;
        __syncthreads();
        // [dW1 | db1][i][c] += sum over the columns of g1[i] * [x | 1][c]
#pragma unroll 8
        for (int s = 0; s < CM_COLS / 2; ++s) {
            const float xb = cl < 8 ? bufX[wv][2 * s + half][cl & 7] : 0.0f;
            acc_w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bufA[wv][2 * s + half][cl], xb, acc_w1, 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- the waves' tiles -> one partial row per workgroup, added in wave order ----
    float* red = &bufA[0][0][0];                 // CM_PART floats (bufA holds CM_BW * 64 * 33)
    static_assert(CM_BW * CM_COLS * LD >= CM_PART, "reduction row does not fit");
    for (int w = 0; w < CM_BW; ++w) {
        if (wv == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;       // D[row][cl]
                float* d2 = red + row * CM_H + cl;                        // dW2[o = row][i = cl]
                *d2 = (w == 0 ? 0.0f : *d2) + acc_w2[r];
                if (cl < 4) {
                    float* d1 = red + CM_H * CM_H + row * 4 + cl;         // dW1[i = row][c = cl]
                    *d1 = (w == 0 ? 0.0f : *d1) + acc_w1[r];
                } else if (cl == 4) {
                    float* d1 = red + CM_H * CM_H + CM_H * 4 + row;       // db1[i = row]
                    *d1 = (w == 0 ? 0.0f : *d1) + acc_w1[r];
                }
            }
            const float both = gb2 + __shfl_xor(gb2, 32, 64);          // the two column halves
            if (half == 0) {
                float* d = red + CM_H * CM_H + CM_H * 4 + CM_H + cl;      // db2[o = cl]
                *d = (w == 0 ? 0.0f : *d) + both;
            }
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < CM_PART; t += 64 * CM_BW) partials[(size_t)blockIdx.x * CM_PART + t] = red[t];
}

// destination += sum over the workgroups' partial rows, in block order.  grid ceil(CM_PART / 64), block (64, 16)
__global__ __launch_bounds__(1024) void corr3d_mlp_reduce_kernel(const float* __restrict__ partials, int n_rows,
                                                                  float* __restrict__ gw1, float* __restrict__ gb1,
                                                                  float* __restrict__ gw2, float* __restrict__ gb2) {
    __shared__ float red[16][64];
    const int t = blockIdx.x * 64 + threadIdx.x;
    float acc = 0.0f;
    if (t < CM_PART) {
#pragma unroll 8
        for (int r = threadIdx.y; r < n_rows; r += 16) acc += partials[(size_t)r * CM_PART + t];
    }
    red[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && t < CM_PART) {
        float v = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[g][threadIdx.x];
        float* dst = t < CM_H * CM_H ? gw2 + t
                     : t < CM_H * CM_H + CM_H * 4 ? gw1 + (t - CM_H * CM_H)
                     : t < CM_H * CM_H + CM_H * 4 + CM_H ? gb1 + (t - CM_H * CM_H - CM_H * 4)
                                                          : gb2 + (t - CM_H * CM_H - CM_H * 4 - CM_H);
        *dst += v;
    }
}

constexpr int CM_CH = 8;      // points per wave

bool mlp_shape_ok(const char* what, int B, int N, int levels, int k, int hidden) {
    if (B < 0 || N < 1 || levels != 4 || k != 16 || hidden != CM_H || N % CM_CH != 0) {
        camli_set_error("%s: B=%d N=%d levels=%d k=%d hidden=%d (kernels cover 4 levels x 16 neighbours, width 32, N %% %d == 0)",
                        what, B, N, levels, k, hidden, CM_CH);
        return false;
    }
    return true;
}

}  // namespace

extern "C" int camli_corr3d_mlp_supported(int levels, int k, int hidden, int N) {
    return levels == 4 && k == 16 && hidden == CM_H && N > 0 && N % CM_CH == 0;
}

extern "C" int camli_corr3d_mlp_fwd(const float* lookup, const float* w1, const float* b1, const float* w2, const float* b2,
                                    float* out, int B, int N, int levels, int k, int hidden, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!lookup || !w1 || !b1 || !w2 || !b2 || !out) { camli_set_error("camli_corr3d_mlp_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!mlp_shape_ok("camli_corr3d_mlp_fwd", B, N, levels, k, hidden)) return CAMLI_EINVAL;
    const int blocks = camli_divup(B * (N / CM_CH), 4);
    hipLaunchKernelGGL(corr3d_mlp_fwd_kernel<CM_CH>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), lookup, w1,
                       b1, w2, b2, out, B, N);
    return camli_check_launch("camli_corr3d_mlp_fwd");
}

extern "C" int64_t camli_corr3d_mlp_bwd_workspace_bytes(int B, int N) {
    if (B < 0 || N < CM_CH) return 0;
    return (int64_t)camli_divup(B * (N / CM_CH), CM_BW) * CM_PART * (int64_t)sizeof(float);
}

extern "C" int camli_corr3d_mlp_bwd(const float* lookup, const float* gout, const float* w1, const float* b1, const float* w2,
                                    const float* b2, float* glookup, float* gw1, float* gb1, float* gw2, float* gb2,
                                    float* workspace, int B, int N, int levels, int k, int hidden, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!lookup || !gout || !w1 || !b1 || !w2 || !b2 || !glookup || !gw1 || !gb1 || !gw2 || !gb2 || !workspace) {
        camli_set_error("camli_corr3d_mlp_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (!mlp_shape_ok("camli_corr3d_mlp_bwd", B, N, levels, k, hidden)) return CAMLI_EINVAL;
    const int blocks = camli_divup(B * (N / CM_CH), CM_BW);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(corr3d_mlp_bwd_kernel<CM_CH>, dim3(blocks), dim3(64 * CM_BW), 0, s, lookup, gout, w1, b1, w2, b2, glookup, workspace,
                       B, N);
    hipLaunchKernelGGL(corr3d_mlp_reduce_kernel, dim3(camli_divup(CM_PART, 64)), dim3(64, 16), 0, s, workspace, blocks, gw1, gb1, gw2,
                       gb2);
    return camli_check_launch("camli_corr3d_mlp_bwd");
}

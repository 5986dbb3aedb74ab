// 1x1 convolution 128 -> 128 on channel-first tensors with bias and activation in the epilogue, gfx950 (round 6).
//
//   y[b][o][p] = act(sum_c w[o][c] * x[b][c][p] + bias[o])          x, y: [B][128][P], P contiguous; w [128][128]
//
// The aligners of SKFusion / FusionAwareInterp (models/clfm.py:11-25,170-181 of the reference) and the point lane's 128-wide
// Conv1d layers are this shape: 2.1 GFLOP over 67 MB at batch 8, 8,160 positions -- the matrix pipe and HBM need the same 13 us,
// the library's strided-batched GEMM takes 33 and the bias / activation pass behind it another 9.
//
// One workgroup = 128 positions of one sample, all 128 outputs.  The weights never touch LDS: wave w keeps the A fragments of its
// 32 output channels for all 32 K steps in 64 VGPRs (lane (m, g) loads float4 w[o][16 j + 4 g ..+3] and feeds component i to K
// step 4 j + i, so K step s = 4 j + i contracts the channels c = 16 j + 4 g + i, g = 0..3).  The x tile [128 c][128 p] is staged
// once (row stride 132 dwords: the four lane groups of a ds_read_b32 fall on banks 0 / 16 / 32 / 48) and every wave reads all of
// it; two workgroups per CU hide each other's staging.  v_mfma_f32_16x16x4_f32: D[o = 4 g + r][p = lane % 16].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pwc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 128;            // input = output channels
constexpr int TP = 128;           // positions per workgroup
constexpr int LDX = TP + 4;       // LDS row stride in dwords
constexpr size_t LDS_BYTES = (size_t)C * LDX * sizeof(float);

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

template <int ACT>
__device__ __forceinline__ float act(float v) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    return v;
}

// grid B * ceil(P / 128), block 256, dynamic LDS LDS_BYTES.  P % 4 == 0; x_bs / y_bs = batch strides in floats (a channel slice
// of a wider tensor is read / written in place); bias may be null.
template <int ACT>
__global__ __launch_bounds__(256, 2) void pwconv128_fwd_kernel(const float* __restrict__ x, int64_t x_bs, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int64_t y_bs, int P,
                                                                int tiles) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int o0 = 32 * wave;
    f32x4 wa[2][8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int j = 0; j < 8; ++j) wa[mf][j] = *reinterpret_cast<const f32x4*>(w + (size_t)(o0 + 16 * mf + l16) * C + 16 * j + 4 * g);
    const float* __restrict__ xb = x + (size_t)b * x_bs;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int e = tid; e < C * (TP / 4); e += 256) {
        const int c = e >> 5, q = (e & 31) << 2;
        const f32x4 v = p0 + q < P ? *reinterpret_cast<const f32x4*>(xb + (size_t)c * P + p0 + q) : z;
        *reinterpret_cast<f32x4*>(xs + c * LDX + q) = v;
    }
    __syncthreads();
    f32x4 acc[2][8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[mf][f] = z;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* row = xs + (16 * j + 4 * g + i) * LDX + l16;
            float bf[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) bf[f] = row[16 * f];
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][j][i], bf[f], acc[0][f], 0, 0, 0);
                acc[1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][j][i], bf[f], acc[1][f], 0, 0, 0);
            }
        }
    }
    float* __restrict__ yb = y + (size_t)b * y_bs;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = bias ? bias[o0 + 16 * mf + 4 * g + r] : 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int p = p0 + 16 * f + l16;
            if (p < P) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yb[(size_t)(o0 + 16 * mf + 4 * g + r) * P + p] = act<ACT>(acc[mf][f][r] + bv[r]);
            }
        }
    }
}


// ---- pipelined form: persistent workgroups walk sub-tiles of 64 positions, the next sub-tile's loads in flight (registers) while
// the matrix pipe works on the current one (LDS double buffer, one barrier per sub-tile) ----
constexpr int SP = 64;             // positions per sub-tile
constexpr int LDS2 = SP + 4;       // row stride in dwords (rows 4 apart land 16 banks apart)
constexpr size_t LDS2_BYTES = (size_t)2 * C * LDS2 * sizeof(float);

// grid <= B * ceil(P / 64) workgroups of 256 threads, dynamic LDS LDS2_BYTES.  P % 4 == 0.
template <int ACT>
__global__ __launch_bounds__(256, 2) void pwconv128_pipe_kernel(const float* __restrict__ x, int64_t x_bs, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int64_t y_bs, int P,
                                                                 int tiles, int total) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int o0 = 32 * wave;
    f32x4 wa[2][8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int j = 0; j < 8; ++j) wa[mf][j] = *reinterpret_cast<const f32x4*>(w + (size_t)(o0 + 16 * mf + l16) * C + 16 * j + 4 * g);
    float bv[2][4];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[mf][r] = bias ? bias[o0 + 16 * mf + 4 * g + r] : 0.f;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // thread's share of a sub-tile: 8 float4, element e = tid + 256 u -> row c = e / 16, float4 column q = e % 16
    f32x4 pre[8];
    auto fetch = [&](int t) {
        const int b = t / tiles, p0 = (t - b * tiles) * SP;
        const float* __restrict__ xb = x + (size_t)b * x_bs;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u, c = e >> 4, q = (e & 15) << 2;
            pre[u] = p0 + q < P ? *reinterpret_cast<const f32x4*>(xb + (size_t)c * P + p0 + q) : z;
        }
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u, c = e >> 4, q = (e & 15) << 2;
            *reinterpret_cast<f32x4*>(buf + c * LDS2 + q) = pre[u];
        }
    };
    int t = blockIdx.x;
    if (t >= total) return;
    fetch(t);
    stash(xs);
    __syncthreads();
    int cur = 0;
    for (; t < total; t += gridDim.x) {
        const int nxt = t + gridDim.x;
        if (nxt < total) fetch(nxt);
        const float* buf = xs + cur * (C * LDS2);
        f32x4 acc[2][4];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[mf][f] = z;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* row = buf + (16 * j + 4 * g + i) * LDS2 + l16;
                float bf[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) bf[f] = row[16 * f];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][j][i], bf[f], acc[0][f], 0, 0, 0);
                    acc[1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][j][i], bf[f], acc[1][f], 0, 0, 0);
                }
            }
        }
        // (tried: x as the A operand with fragment row m = position 4 m + f -- one ds_read_b128 per K step and float4 stores along
        // p, unpadded rows: 32.6 us against this form's 29.5)
        const int b = t / tiles, p0 = (t - b * tiles) * SP;
        float* __restrict__ yb = y + (size_t)b * y_bs;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int p = p0 + 16 * f + l16;
                if (p < P) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) yb[(size_t)(o0 + 16 * mf + 4 * g + r) * P + p] = act<ACT>(acc[mf][f][r] + bv[mf][r]);
                }
            }
        if (nxt < total) stash(xs + (cur ^ 1) * (C * LDS2));
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace pwc

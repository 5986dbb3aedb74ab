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
#include <stdlib.h>
#include <string.h>

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

// ---------------------------------------------------------------------------------------------------------
// Wave-per-point forms for the shapes every model uses (k = 16 neighbours, Wn = 16 mixing weights).
//
// The workgroup-per-point kernels above are latency-bound: a runtime-k loop issues one dependent
// (index -> row) load at a time and 256 scalar weight loads per point, so a point costs tens of microseconds of
// exposed latency (0.11 of the HBM roofline at the Encoder3D shapes).  Here ONE WAVE owns a point, lanes run along
// the channel-contiguous rows, everything is unrolled at compile time:
//   fwd : the 16 neighbour rows of a channel pass are 16 independent coalesced loads in flight at once; the 16x16
//         weights of the point are wave-uniform (scalar loads, four 64-byte rows at a time); 256 FMAs per lane
//   bwd : two kernels, no atomics, bit-reproducible.
//         (1) per point: T[n,j,:] = sum_w wgt[w,n,j] * gout[n,w,:] (the per-neighbour row gradient, written once,
//             coalesced) and gwgt[w,n,j] = <gout[n,w,:], feat[idx_j,:]> from an LDS-staged copy of the 32 rows
//             (each lane owns 4 of the 256 dot products, ds_read_b128 along the channels)
//         (2) per source point m: gfeat[m,:] = sum of the T rows that reference m, walked through the inverse
//             neighbour map (CSR: positions sorted by source index, built once per neighbour table by the host).
//             The reference's index_put_(accumulate=True) does the same with a device-wide sort PER CALL; the
//             previous kernel used 52 M float atomics per Encoder3D level.
constexpr int PCW_K = 16, PCW_WN = 16;

// grid ceil(B*N / 4), block 256 (one point per wave).  Channels are walked in blocks of 128 (two per lane); the
// weight loop is NOT unrolled across its four groups so that only 64 of the 256 wave-uniform weights occupy
// SGPRs at a time (all 256 would spill).
__global__ __launch_bounds__(256) void pointconv_mix_fwd_wave_kernel(const float* __restrict__ feat,
                                                                      const float* __restrict__ wgt,
                                                                      const int64_t* __restrict__ idx, int idx_stride,
                                                                      float* __restrict__ out, int B, int M, int N,
                                                                      int CH) {
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (p >= B * N) return;
    const int b = p / N, n = p - b * N;
    const int64_t* __restrict__ irow = idx + (size_t)p * idx_stride;
    const float* __restrict__ wbase = wgt + (size_t)b * PCW_WN * N * PCW_K + (size_t)n * PCW_K;
    const float* __restrict__ fb = feat + (size_t)b * M * CH;
    int m[PCW_K];
#pragma unroll
    for (int j = 0; j < PCW_K; ++j) m[j] = (int)irow[j];
    float* __restrict__ o = out + (size_t)p * PCW_WN * CH;
    for (int c0 = 0; c0 < CH; c0 += 128) {
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        const bool oka = ca < CH, okb = cb < CH;
        float fa[PCW_K], fbv[PCW_K];
#pragma unroll
        for (int j = 0; j < PCW_K; ++j) {
            const float* __restrict__ row = fb + (size_t)m[j] * CH;
            fa[j] = oka ? row[ca] : 0.0f;
            fbv[j] = okb ? row[cb] : 0.0f;
        }
#pragma unroll 1
        for (int wg = 0; wg < PCW_WN; wg += 4) {
#pragma unroll
            for (int w = wg; w < wg + 4; ++w) {
                const float* __restrict__ wr = wbase + (size_t)w * N * PCW_K;
                float acc_a = 0.0f, acc_b = 0.0f;
#pragma unroll
                for (int j = 0; j < PCW_K; ++j) {
                    const float wv = wr[j];
                    acc_a = __builtin_fmaf(wv, fa[j], acc_a);
                    acc_b = __builtin_fmaf(wv, fbv[j], acc_b);
                }
                if (oka) o[(size_t)w * CH + ca] = acc_a;
                if (okb) o[(size_t)w * CH + cb] = acc_b;
            }
        }
    }
}

// (1) grid ceil(B*N / 4), block 256; dynamic LDS 4 * 32 * CHP floats (CHP = 4 * (ceil(CH/4) | 1): an odd number of
// 16-byte slots per row keeps the 16 row-strided ds_read_b128 of a lane group on distinct banks)
__global__ __launch_bounds__(256) void pointconv_mix_bwd_point_kernel(const float* __restrict__ gout,
                                                                       const float* __restrict__ feat,
                                                                       const float* __restrict__ wgt,
                                                                       const int64_t* __restrict__ idx, int idx_stride,
                                                                       float* __restrict__ trows /*[B,N,16,CH] or null*/,
                                                                       float* __restrict__ gwgt /*[B,16,N,16] or null*/,
                                                                       int B, int M, int N, int CH, int CHP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (p >= B * N) return;          // whole waves leave; no block-wide barrier below
    const int b = p / N, n = p - b * N;
    float* sg = lds + (size_t)wave * 32 * CHP;     // [16][CHP] gout rows
    float* sf = sg + 16 * CHP;                     // [16][CHP] gathered feature rows
    const int64_t* __restrict__ irow = idx + (size_t)p * idx_stride;
    const float* __restrict__ wbase = wgt + (size_t)b * PCW_WN * N * PCW_K + (size_t)n * PCW_K;
    const float* __restrict__ fb = feat + (size_t)b * M * CH;
    const float* __restrict__ g = gout + (size_t)p * PCW_WN * CH;
    int m[PCW_K];
#pragma unroll
    for (int j = 0; j < PCW_K; ++j) m[j] = (int)irow[j];
    // channels in blocks of 128 (two per lane): the weights are walked row by row (four 64-byte rows = 64 scalars in
    // SGPRs at a time, as in the forward kernel) and each one feeds both channel halves
    for (int c0 = 0; c0 < CHP; c0 += 128) {
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        const bool oka = ca < CH, okb = cb < CH;
        float ga[PCW_WN], gb[PCW_WN];
#pragma unroll
        for (int w = 0; w < PCW_WN; ++w) {
            ga[w] = oka ? g[(size_t)w * CH + ca] : 0.0f;
            gb[w] = okb ? g[(size_t)w * CH + cb] : 0.0f;
        }
        if (gwgt) {
#pragma unroll
            for (int w = 0; w < PCW_WN; ++w) {
                if (ca < CHP) sg[w * CHP + ca] = ga[w];
                if (cb < CHP) sg[w * CHP + cb] = gb[w];
            }
#pragma unroll
            for (int j = 0; j < PCW_K; ++j) {
                const float* __restrict__ row = fb + (size_t)m[j] * CH;
                if (ca < CHP) sf[j * CHP + ca] = oka ? row[ca] : 0.0f;
                if (cb < CHP) sf[j * CHP + cb] = okb ? row[cb] : 0.0f;
            }
        }
        if (trows) {
            float ta[PCW_K], tb[PCW_K];
#pragma unroll
            for (int j = 0; j < PCW_K; ++j) ta[j] = tb[j] = 0.0f;
#pragma unroll 1
            for (int wg = 0; wg < PCW_WN; wg += 4) {
#pragma unroll
                for (int w = wg; w < wg + 4; ++w) {
                    const float* __restrict__ wr = wbase + (size_t)w * N * PCW_K;
                    const float gwa = ga[w], gwb = gb[w];
#pragma unroll
                    for (int j = 0; j < PCW_K; ++j) {
                        const float wv = wr[j];
                        ta[j] = __builtin_fmaf(wv, gwa, ta[j]);
                        tb[j] = __builtin_fmaf(wv, gwb, tb[j]);
                    }
                }
            }
            float* __restrict__ t = trows + (size_t)p * PCW_K * CH;
#pragma unroll
            for (int j = 0; j < PCW_K; ++j) {
                if (oka) t[(size_t)j * CH + ca] = ta[j];
                if (okb) t[(size_t)j * CH + cb] = tb[j];
            }
        }
    }
    if (gwgt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();       // the wave reads back only what its own lanes wrote above
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int w = lane & 15, j0 = 4 * (lane >> 4);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        const float4* __restrict__ gq = reinterpret_cast<const float4*>(sg + w * CHP);
        const float4* __restrict__ f0 = reinterpret_cast<const float4*>(sf + (j0 + 0) * CHP);
        const float4* __restrict__ f1 = reinterpret_cast<const float4*>(sf + (j0 + 1) * CHP);
        const float4* __restrict__ f2 = reinterpret_cast<const float4*>(sf + (j0 + 2) * CHP);
        const float4* __restrict__ f3 = reinterpret_cast<const float4*>(sf + (j0 + 3) * CHP);
        for (int t = 0; t < CHP / 4; ++t) {
            const float4 gg = gq[t];
            float4 ff = f0[t];
            a0 = __builtin_fmaf(gg.x, ff.x, a0); a0 = __builtin_fmaf(gg.y, ff.y, a0);
            a0 = __builtin_fmaf(gg.z, ff.z, a0); a0 = __builtin_fmaf(gg.w, ff.w, a0);
            ff = f1[t];
            a1 = __builtin_fmaf(gg.x, ff.x, a1); a1 = __builtin_fmaf(gg.y, ff.y, a1);
            a1 = __builtin_fmaf(gg.z, ff.z, a1); a1 = __builtin_fmaf(gg.w, ff.w, a1);
            ff = f2[t];
            a2 = __builtin_fmaf(gg.x, ff.x, a2); a2 = __builtin_fmaf(gg.y, ff.y, a2);
            a2 = __builtin_fmaf(gg.z, ff.z, a2); a2 = __builtin_fmaf(gg.w, ff.w, a2);
            ff = f3[t];
            a3 = __builtin_fmaf(gg.x, ff.x, a3); a3 = __builtin_fmaf(gg.y, ff.y, a3);
            a3 = __builtin_fmaf(gg.z, ff.z, a3); a3 = __builtin_fmaf(gg.w, ff.w, a3);
        }
        *reinterpret_cast<float4*>(gwgt + ((size_t)b * PCW_WN + w) * N * PCW_K + (size_t)n * PCW_K + j0) =
            make_float4(a0, a1, a2, a3);
    }
}

// (1, round 3 product path) the same two results on the matrix cores, no LDS: one wave per point, and both
// contractions are 16x16 tiles of v_mfma_f32_16x16x4_f32 whose operands are exactly what a lane loads from memory:
//   gwgt[w][j] = sum_c gout[w][c] * feat[idx_j][c]   A[i = w][k] and B[k][j]: lane (row = lane % 16, quad = lane / 16)
//                reads 16 bytes (4 consecutive channels) of gout row w = row and of feature row idx_{j = row}; the four
//                components are four K steps (the K index of an MFMA is arbitrary as long as A and B agree), so a load
//                instruction fetches 64 contiguous bytes of each of the 16 rows and nothing is transposed
//   T[j][c]    = sum_w wgt[w][j] * gout[w][c]        A[i = j][k = w] = the point's 16x16 weights (4 registers, loaded once),
//                B[k = w][n = c]: lane (c = c0 + lane % 16, w = 4 s + lane / 16) reads gout again (L1 / L2 hit), D gives
//                T[j = 4 (lane / 16) + r][c] -> 64-byte row segments
// The LDS form above staged 32 rows per wave (68 KB per workgroup at CH = 131: 2 waves per SIMD) and spent its time in
// ds_read_b128 + 512 fmaf per lane; here the arithmetic is 8 * ceil(CH / 16) matrix instructions per point and the
// kernel is bound by its ~27 KB of memory traffic per point.  Rows are only 4-byte aligned (CH = 99, 131 ...): the
// 16-byte loads are declared with 4-byte alignment (gfx950 global loads take any dword-aligned address).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ f32x4_t load_quad(const float* __restrict__ row, int q, int CH) {
    const int c = 4 * q;
    if (c + 3 < CH) return *reinterpret_cast<const f32x4_u*>(row + c);
    f32x4_t v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c < CH) v[0] = row[c];
    if (c + 1 < CH) v[1] = row[c + 1];
    if (c + 2 < CH) v[2] = row[c + 2];
    return v;
}

__global__ __launch_bounds__(256) void pointconv_mix_bwd_point_mfma_kernel(const float* __restrict__ gout,
                                                                            const float* __restrict__ feat,
                                                                            const float* __restrict__ wgt,
                                                                            const int64_t* __restrict__ idx, int idx_stride,
                                                                            float* __restrict__ trows /*[B,N,16,CH] or null*/,
                                                                            float* __restrict__ gwgt /*[B,16,N,16] or null*/,
                                                                            int B, int M, int N, int CH) {
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (p >= B * N) return;
    const int b = p / N, n = p - b * N;
    const int r16 = lane & 15, g4 = lane >> 4;
    const float* __restrict__ g = gout + (size_t)p * PCW_WN * CH;
    const float* __restrict__ wbase = wgt + (size_t)b * PCW_WN * N * PCW_K + (size_t)n * PCW_K;
    if (gwgt) {
        const int mrow = (int)idx[(size_t)p * idx_stride + r16];
        const float* __restrict__ grow = g + (size_t)r16 * CH;
        const float* __restrict__ frow = feat + ((size_t)b * M + mrow) * CH;
        const int nq = (CH + 3) >> 2;
        f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int q0 = 0; q0 < nq; q0 += 16) {       // four quads per lane and trip: eight 16-byte loads in flight
            f32x4_t a[4], f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + 4 * u + g4;       // quads >= nq load zeros: they add nothing
                a[u] = load_quad(grow, q, CH);
                f[u] = load_quad(frow, q, CH);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], f[u][e], acc, 0, 0, 0);
            }
        }
        // D[i = w = 4 g4 + r][j = r16]
#pragma unroll
        for (int r = 0; r < 4; ++r)
            gwgt[(((size_t)b * PCW_WN + 4 * g4 + r) * N + n) * PCW_K + r16] = acc[r];
    }
    if (trows) {
        float wa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) wa[s] = wbase[(size_t)(4 * s + g4) * N * PCW_K + r16];    // wgt[w = 4 s + g4][j = r16]
        float* __restrict__ t = trows + (size_t)p * PCW_K * CH;
        for (int c0 = 0; c0 < CH; c0 += 64) {      // four 16-channel tiles per trip: sixteen loads in flight
            float bv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 16 * u + r16;
#pragma unroll
                for (int s = 0; s < 4; ++s) bv[u][s] = c < CH ? g[(size_t)(4 * s + g4) * CH + c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 16 * u + r16;
                f32x4_t d = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < 4; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], bv[u][s], d, 0, 0, 0);
                if (c < CH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[(size_t)(4 * g4 + r) * CH + c] = d[r];    // T[j = 4 g4 + r][c]
                }
            }
        }
    }
}

// (2) grid ceil(B*M / 4), block 256 (one source point per wave): rows[pos] summed over the CSR segment of (b, m).
// Used for the PointConv feature gradient (rows = T, RL = CH) -- positions index rows of length RL.
__global__ __launch_bounds__(256) void segment_row_sum_kernel(const float* __restrict__ rows,
                                                               const int32_t* __restrict__ order,
                                                               const int32_t* __restrict__ offsets,
                                                               float* __restrict__ dst, int n_segments, int RL) {
    const int lane = threadIdx.x & 63;
    const int s = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (s >= n_segments) return;
    const int beg = offsets[s], end = offsets[s + 1];
    for (int c0 = 0; c0 < RL; c0 += 128) {
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        float acc_a = 0.0f, acc_b = 0.0f;
        int e = beg;
        for (; e + 4 <= end; e += 4) {
            const float* r0 = rows + (size_t)order[e] * RL;
            const float* r1 = rows + (size_t)order[e + 1] * RL;
            const float* r2 = rows + (size_t)order[e + 2] * RL;
            const float* r3 = rows + (size_t)order[e + 3] * RL;
            const float v0 = ca < RL ? r0[ca] : 0.0f, v1 = ca < RL ? r1[ca] : 0.0f;
            const float v2 = ca < RL ? r2[ca] : 0.0f, v3 = ca < RL ? r3[ca] : 0.0f;
            const float u0 = cb < RL ? r0[cb] : 0.0f, u1 = cb < RL ? r1[cb] : 0.0f;
            const float u2 = cb < RL ? r2[cb] : 0.0f, u3 = cb < RL ? r3[cb] : 0.0f;
            acc_a = (((acc_a + v0) + v1) + v2) + v3;
            acc_b = (((acc_b + u0) + u1) + u2) + u3;
        }
        for (; e < end; ++e) {
            const float* r0 = rows + (size_t)order[e] * RL;
            if (ca < RL) acc_a += r0[ca];
            if (cb < RL) acc_b += r0[cb];
        }
        if (ca < RL) dst[(size_t)s * RL + ca] = acc_a;
        if (cb < RL) dst[(size_t)s * RL + cb] = acc_b;
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
    if (k == PCW_K && Wn == PCW_WN) {
        hipLaunchKernelGGL(pointconv_mix_fwd_wave_kernel, dim3(camli_divup(B * N, 4)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), feat_cl, wgt, idx, idx_stride, out, B, M, N, CH);
        return camli_check_launch("camli_pointconv_mix_fwd(wave)");
    }
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

extern "C" int64_t camli_pointconv_mix_bwd_scratch_bytes(int B, int N, int CH, int k) {
    if (B < 0 || N < 1 || CH < 1 || k < 1) return 0;
    return (int64_t)B * N * k * CH * (int64_t)sizeof(float);
}

extern "C" int camli_pointconv_mix_bwd_sorted(const float* gout, const float* feat_cl, const float* wgt,
                                              const int64_t* idx, int idx_stride, const int32_t* inv_order,
                                              const int32_t* inv_offsets, float* scratch, float* gfeat_cl, float* gwgt,
                                              int B, int M, int N, int CH, int Wn, int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gout || !feat_cl || !wgt || !idx || (!gfeat_cl && !gwgt)) {
        camli_set_error("camli_pointconv_mix_bwd_sorted: null pointer");
        return CAMLI_EINVAL;
    }
    if (!mix_args_ok("camli_pointconv_mix_bwd_sorted", B, M, N, CH, Wn, k, idx_stride)) return CAMLI_EINVAL;
    if (k != PCW_K || Wn != PCW_WN) {
        camli_set_error("camli_pointconv_mix_bwd_sorted: needs k = 16 and Wn = 16 (got k=%d Wn=%d); use camli_pointconv_mix_bwd", k, Wn);
        return CAMLI_ENOTSUP;
    }
    if (gfeat_cl && (!inv_order || !inv_offsets || !scratch)) {
        camli_set_error("camli_pointconv_mix_bwd_sorted: the feature gradient needs the inverse map and the scratch rows");
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const bool lds_form = []() { const char* e = getenv("CAMLI_MIX_BWD"); return e && !strcmp(e, "lds"); }();
    if (!lds_form) {
        hipLaunchKernelGGL(pointconv_mix_bwd_point_mfma_kernel, dim3(camli_divup(B * N, 4)), dim3(256), 0, s, gout, feat_cl,
                           wgt, idx, idx_stride, gfeat_cl ? scratch : nullptr, gwgt, B, M, N, CH);
    } else {
    const int CHP = 4 * (camli_divup(CH, 4) | 1);
    const size_t lds = gwgt ? (size_t)4 * 32 * CHP * sizeof(float) : 0;
    if (lds > 150 * 1024) {
        camli_set_error("camli_pointconv_mix_bwd_sorted: CH = %d exceeds the LDS tile", CH);
        return CAMLI_ENOTSUP;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pointconv_mix_bwd_point_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(pointconv_mix_bwd_point_kernel, dim3(camli_divup(B * N, 4)), dim3(256), lds, s, gout, feat_cl, wgt,
                       idx, idx_stride, gfeat_cl ? scratch : nullptr, gwgt, B, M, N, CH, CHP);
    }
    if (gfeat_cl)
        hipLaunchKernelGGL(segment_row_sum_kernel, dim3(camli_divup(B * M, 4)), dim3(256), 0, s, scratch, inv_order,
                           inv_offsets, gfeat_cl, B * M, CH);
    return camli_check_launch("camli_pointconv_mix_bwd_sorted");
}

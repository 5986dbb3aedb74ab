// Set-conv neighbour-weight network  W = MLP2d(3 -> 8 -> 32 -> C, relu)(xyz[knn] - centre), gfx950.
//
// Replaces, for one PointConvDW instance, the composed path of models/point_conv.py:110-121 +
// models/mlp.py:100-162 (gather the neighbours, subtract the centre, three 1x1 convolutions each
// followed by a bias add and a ReLU pass: 8 launches and three [B,*,N,k] intermediates).
//
//   out[b, c, n, j] = relu(b3[c] + sum_q W3[c,q] * h2[q]),   h2 = relu(b2 + W2 h1),  h1 = relu(b1 + W1 d)
//   d = xyz[b, :, idx[b,n,j]] - centres[b, :, n]
//
// Every layer is "bias first, then a q-ordered fmaf chain" -- exactly what v_mfma_f32_32x32x2_f32
// computes when the accumulator is initialised with the bias -- so the kernel is bit-exact against
// the scalar restatement oracle_weightnet_fwd.
//
// The last layer carries 2*32*C flop per neighbour (93 % of the work) and runs on the matrix cores:
// per batch it is D[C x N*k] = W3[C x 32] * H2[32 x N*k].  One wave owns a tile of 32 consecutive
// columns (n,j) and all C rows:
//   * lane l computes the offset and h1 of column (l & 31), and the 16 h2 rows q = 2s + (l >> 5):
//     these ARE the B fragments (B[k = l>>5][j = l&31]) of the 16 K-steps -- no LDS round trip
//   * the A fragments W3[32t + (l&31)][2s + (l>>5)] stay in registers for the whole kernel
//   * C/D layout: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5): each store instruction writes two
//     128-byte row segments; the [B,C,N,k] output is written exactly once and nothing else touches HBM
// Roofline: HBM write of 4*B*C*N*k bytes vs 2*B*N*k*(24 + 256 + 32*C) flop at the fp32 MFMA rate
// (157 TFLOP/s): for C = 128 the write takes ~2x the MFMA time, so the kernel is HBM(write)-bound.
#include "camli_common.h"

#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WN_MAXC = 128;

__device__ __forceinline__ int wn_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// offset + layer 1 of one column; all lanes of a half-wave pair (l, l+32) compute the same column
__device__ __forceinline__ void wn_hidden1(const float* __restrict__ xyz, const float* __restrict__ centres,
                                           const int64_t* __restrict__ idx, int idx_stride,
                                           const float* __restrict__ w1, const float* __restrict__ b1, int b, int M,
                                           int N, int k, int col, float (&h1)[8], float (&off)[3], int kmajor = 0) {
    // column enumeration of the [.., N*k] output: (n, j) with j fastest ([B,C,N,k]), or -- k-major, [B,C,k,N] -- with n
    // fastest: 32 consecutive columns are then 32 consecutive points of ONE neighbour slot, which is the order the
    // set-conv forward streams them in (setconv.hip, k-major kernel)
    const int n = kmajor ? col % N : col / k;
    const int j = kmajor ? col / N : col - n * k;
    const int id = (int)idx[((size_t)b * N + n) * idx_stride + j];
#pragma unroll
    for (int d = 0; d < 3; ++d) off[d] = xyz[((size_t)b * 3 + d) * M + id] - centres[((size_t)b * 3 + d) * N + n];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = b1[i];
        a = __builtin_fmaf(w1[i * 3 + 0], off[0], a);
        a = __builtin_fmaf(w1[i * 3 + 1], off[1], a);
        a = __builtin_fmaf(w1[i * 3 + 2], off[2], a);
        h1[i] = fmaxf(a, 0.0f);
    }
}

// grid-stride over (batch, 32-column tile); block 256 = 4 independent waves
template <int MT>
__global__ __launch_bounds__(256) void weightnet_fwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ centres, const int64_t* __restrict__ idx, int idx_stride,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    float* __restrict__ out, int B, int C, int M, int N, int k, int kmajor) {
    __shared__ __attribute__((aligned(16))) float s_w2[32 * 8];
    __shared__ float s_b2[32];
    __shared__ __attribute__((aligned(16))) float s_b3[WN_MAXC];
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, cl = lane & 31;
    s_w2[tid] = w2[tid];
    if (tid < 32) s_b2[tid] = b2[tid];
    if (tid < WN_MAXC) s_b3[tid] = tid < C ? b3[tid] : 0.0f;
    // W3 -> LDS with coalesced loads (row stride 33), then each lane picks its A fragments
    __shared__ float s_w3f[32 * MT * 33];
    for (int e = tid; e < 32 * MT * 32; e += 256) {
        const int c = e >> 5, q = e & 31;
        s_w3f[c * 33 + q] = c < C ? w3[c * 32 + q] : 0.0f;
    }
    __syncthreads();
    float a3[MT][16];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s) a3[t][s] = s_w3f[(32 * t + cl) * 33 + 2 * s + half];

    const int NK = N * k;
    const int tiles_per_batch = (NK + 31) / 32;
    const int total = B * tiles_per_batch;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform tile / batch / row offsets
    for (int tile = blockIdx.x * 4 + wave; tile < total; tile += gridDim.x * 4) {
        // keep the (loop-invariant) LDS reads of W2 / b2 / b3 inside the loop: hoisted they cost 200+ VGPRs
        asm volatile("" ::: "memory");
        const int b = tile / tiles_per_batch;
        const int col = (tile - b * tiles_per_batch) * 32 + cl;
        const bool valid = col < NK;
        float h1[8], off[3];
        wn_hidden1(xyz, centres, idx, idx_stride, w1, b1, b, M, N, k, valid ? col : NK - 1, h1, off, kmajor);
        float h2[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int q = 2 * s + half;
            float a = s_b2[q];
#pragma unroll
            for (int i = 0; i < 8; ++i) a = __builtin_fmaf(s_w2[q * 8 + i], h1[i], a);
            h2[s] = fmaxf(a, 0.0f);
        }
        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = s_b3[32 * t + wn_row(r, half)];
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[t][s], h2[s], acc[t], 0, 0, 0);
        if (valid) {
            // lane part of the address: its column and its half's +4 rows; the rest is wave-uniform
            float* __restrict__ dst = out + (size_t)b * C * NK;
            const int lane_off = 4 * half * NK + col;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int urow = 32 * t + wn_row(r, 0);
                    if (urow + 4 * half < C) dst[(size_t)urow * NK + lane_off] = fmaxf(acc[t][r], 0.0f);
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Backward: gradients of all six parameters from gout [B,C,N,k] in one launch (the coordinates are
// constants).  Same tiling (one wave = 32 columns x all C rows); every product runs on the matrix
// cores.  With the column index on the lanes (C/D layout of the forward):
//   pre3 = b3 + W3 h2              as the forward, same fmaf order -> identical ReLU mask
//   g3   = gout * (pre3 > 0)       gout read in the C/D layout (two 128-byte row segments per load)
//   gh2[q,col] = sum_c W3[c,q] g3[c,col]        A = W3^T from LDS, B = g3 straight from the C/D
//                                               registers (contraction index c <-> (t, r, half))
//   g2 = gh2 * (h2 > 0);  gh1 = W2^T g2 (VALU + one cross-half add);  g1 = gh1 * (h1 > 0)
// The parameter gradients contract over the columns, i.e. over the lanes' index, so their operands
// are transposed through wave-private LDS tiles [row][33]:
//   gw3[c,q] += g3 h2^T          (MT x 16 MFMA)         gb3[c] += sum of the A fragments
//   [gw2 | gb2][q, 0..8] += g2 [h1 ; 1]^T   (16 MFMA)   [gw1 | gb1][i, 0..3] += g1 [d ; 1]^T (16 MFMA)
// Partial sums stay in registers across all tiles of a wave; gw3/gb3 are combined per workgroup in
// LDS and flushed with one float atomic per element per workgroup, the small ones per wave.
// ---------------------------------------------------------------------------------------------
constexpr int WN_LD = 33;   // padded LDS row stride (conflict-free row-major <-> column access)

template <int MT, int OCC>
__global__ __launch_bounds__(256, OCC) void weightnet_bwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ centres, const int64_t* __restrict__ idx, int idx_stride,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    const float* __restrict__ gout, float* __restrict__ partials, int B, int C, int M, int N, int k, int kmajor) {
    __shared__ __attribute__((aligned(16))) float s_w2[32 * 8];
    __shared__ float s_b2[32];
    __shared__ __attribute__((aligned(16))) float s_b3[WN_MAXC];
    __shared__ float s_w3[32 * MT * WN_LD];          // W3[c][q], zero rows beyond C
    __shared__ float s_t[4][2][32 * WN_LD];          // per wave: two transpose tiles [row][col]
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, cl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_w2[tid] = w2[tid];
    if (tid < 32) s_b2[tid] = b2[tid];
    if (tid < WN_MAXC) s_b3[tid] = tid < C ? b3[tid] : 0.0f;
    for (int e = tid; e < 32 * MT * 32; e += 256) {
        const int c = e >> 5, q = e & 31;
        s_w3[c * WN_LD + q] = c < C ? w3[c * 32 + q] : 0.0f;
    }
    __syncthreads();
    float* __restrict__ s_g = s_t[wave][0];
    float* __restrict__ s_h = s_t[wave][1];

    f32x16 gw[MT], gsm;     // gsm: [gw2 | gb2] in columns 0..8, [gw1 | gb1] (rows 0..7) in columns 16..19
    float gb[MT];
#pragma unroll
    for (int r = 0; r < 16; ++r) gsm[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        gb[t] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) gw[t][r] = 0.0f;
    }

    const int NK = N * k;
    const int tiles_per_batch = (NK + 31) / 32;
    const int total = B * tiles_per_batch;
    for (int tile = blockIdx.x * 4 + wave; tile < total; tile += gridDim.x * 4) {
        asm volatile("" ::: "memory");
        const int b = tile / tiles_per_batch;
        const int col = (tile - b * tiles_per_batch) * 32 + cl;
        const bool valid = col < NK;
        float h1[8], off[3];
        wn_hidden1(xyz, centres, idx, idx_stride, w1, b1, b, M, N, k, valid ? col : NK - 1, h1, off, kmajor);

        // ---- hidden layer 2 in C/D row order: transpose tile for gw3, ReLU mask for g2 ----
        unsigned h2_mask = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = wn_row(r, half);
            float a = s_b2[q];
#pragma unroll
            for (int i = 0; i < 8; ++i) a = __builtin_fmaf(s_w2[q * 8 + i], h1[i], a);
            s_h[q * WN_LD + cl] = fmaxf(a, 0.0f);
            h2_mask |= (a > 0.0f ? 1u : 0u) << r;
            if (r & 1) __builtin_amdgcn_sched_barrier(0);      // 16 weights in flight, not 128 (register budget)
        }
        __builtin_amdgcn_wave_barrier();
        float h2f[16];      // the column's h2 in the forward's K order, read back from the tile (other half's rows too)
#pragma unroll
        for (int s = 0; s < 16; ++s) h2f[s] = s_h[(2 * s + half) * WN_LD + cl];
        // ---- per 32-row tile of C: pre3 (forward operand order -> identical mask), g3 = gout * (pre3 > 0),
        // gh2 += W3^T g3 (K = (r, half) <-> c = 32t + wn_row(r, half)), gw3 += g3 h2^T via the transpose tile.
        // gout of a tile is requested before its pre3 MFMAs and consumed after them. ----
        const float* __restrict__ src = gout + (size_t)b * C * NK;
        const int lane_off = 4 * half * NK + col;
        f32x16 gd;
#pragma unroll
        for (int r = 0; r < 16; ++r) gd[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            __builtin_amdgcn_sched_barrier(0);      // keep the compiler from hoisting later tiles' loads (register budget)
            float g[16];                            // requested before this tile's pre3 MFMAs, consumed after them
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int urow = 32 * t + wn_row(r, 0);
                g[r] = (valid && urow + 4 * half < C) ? src[(size_t)urow * NK + lane_off] : 0.0f;
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = s_b3[32 * t + wn_row(r, half)];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w3[(32 * t + cl) * WN_LD + 2 * s + half], h2f[s], acc, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // operands of four steps in flight, not of sixteen
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.0f ? g[r] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gd = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w3[(32 * t + wn_row(r, half)) * WN_LD + cl], acc[r], gd, 0, 0, 0);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) s_g[wn_row(r, half) * WN_LD + cl] = acc[r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float ga = s_g[cl * WN_LD + 2 * s + half];   // A[i = c_local = cl][k = col = 2s + half]
                gb[t] += ga;
                gw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, s_h[cl * WN_LD + 2 * s + half], gw[t], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- g2 = gh2 * (h2 > 0) stays in gd ----
#pragma unroll
        for (int r = 0; r < 16; ++r) gd[r] = (h2_mask >> r) & 1u ? gd[r] : 0.0f;
        // ---- [gw2 | gb2] += g2 [h1 ; 1]^T ----
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) s_g[wn_row(r, half) * WN_LD + cl] = gd[r];
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_h[i * WN_LD + cl] = h1[i];
            s_h[8 * WN_LD + cl] = 1.0f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a2 = s_g[cl * WN_LD + 2 * s + half];                         // g2[q = cl][col]
            const float bv = s_h[(cl < 9 ? cl : 0) * WN_LD + 2 * s + half];          // [h1 ; 1][j = cl][col]
            gsm = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, cl < 9 ? bv : 0.0f, gsm, 0, 0, 0);
        }
        // ---- gh1 = W2^T g2 (each half holds 16 of the 32 q), g1 = gh1 * (h1 > 0) ----
        float g1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) g1[i] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = wn_row(r, half);
#pragma unroll
            for (int i = 0; i < 8; ++i) g1[i] = __builtin_fmaf(s_w2[q * 8 + i], gd[r], g1[i]);
            if (r & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            g1[i] += __shfl_xor(g1[i], 32, 64);
            g1[i] = h1[i] > 0.0f ? g1[i] : 0.0f;
        }
        // ---- [gw1 | gb1] += g1 [d ; 1]^T ----
        __builtin_amdgcn_wave_barrier();
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_g[i * WN_LD + cl] = g1[i];
#pragma unroll
            for (int d = 0; d < 3; ++d) s_h[d * WN_LD + cl] = off[d];
            s_h[3 * WN_LD + cl] = 1.0f;
        }
        __builtin_amdgcn_wave_barrier();
        const bool bsel = cl >= 16 && cl < 20;      // output columns 16..19 <- [d ; 1] rows 0..3
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a1 = s_g[(cl < 8 ? cl : 0) * WN_LD + 2 * s + half];
            const float bv = s_h[(bsel ? cl - 16 : 0) * WN_LD + 2 * s + half];
            gsm = __builtin_amdgcn_mfma_f32_32x32x2f32(cl < 8 ? a1 : 0.0f, bsel ? bv : 0.0f, gsm, 0, 0, 0);
        }
    }

    // ---- combine the four waves in LDS in a FIXED order (wave 0, 1, 2, 3 take turns; within a turn every lane owns
    // its addresses; the two half-wave partials of the bias sums are merged in registers first), then the workgroup's
    // partial sums go to its slice of the workspace and a second kernel adds the slices in a fixed order: no atomics
    // anywhere, bit-reproducible (round 2 merged the waves with LDS float atomics, whose order is not) ----
    __syncthreads();
    constexpr int RED_W3 = 32 * MT * WN_LD;      // gw3 rows padded to 33: column 32 = gb3
    constexpr int RED_W2 = RED_W3 + 32 * 9;      // [gw2 | gb2] as [32][9]
    constexpr int RED_ALL = RED_W2 + 8 * 4;      // [gw1 | gb1] as [8][4]
    float* red = &s_t[0][0][0];                  // 4*2*32*33 floats >= RED_ALL
    for (int e = tid; e < RED_ALL; e += 256) red[e] = 0.0f;
#pragma unroll
    for (int t = 0; t < MT; ++t) gb[t] += __shfl_xor(gb[t], 32, 64);
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(32 * t + wn_row(r, half)) * WN_LD + cl] += gw[t][r];
                if (half == 0) red[(32 * t + cl) * WN_LD + 32] += gb[t];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = wn_row(r, half);
                if (cl < 9) red[RED_W3 + q * 9 + cl] += gsm[r];
                if (r < 4 && cl >= 16 && cl < 20) red[RED_W2 + q * 4 + cl - 16] += gsm[r];   // rows 0..7: r in 0..3
            }
        }
        __syncthreads();
    }
    float* __restrict__ mine = partials + (size_t)blockIdx.x * RED_ALL;
    for (int e = tid; e < RED_ALL; e += 256) mine[e] = red[e];
}

// Second stage: out element e = sum over the workgroups' slices, added in a fixed order.
// block (64 elements x 16 slice groups), grid ceil(red_all / 64).
__global__ __launch_bounds__(1024) void weightnet_reduce_kernel(const float* __restrict__ partials, int n_slices,
                                                                 int red_all, int red_w3, int red_w2, int C,
                                                                 float* __restrict__ gw1, float* __restrict__ gb1,
                                                                 float* __restrict__ gw2, float* __restrict__ gb2,
                                                                 float* __restrict__ gw3, float* __restrict__ gb3) {
    __shared__ float part[16][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float acc = 0.0f;
    if (e < red_all)
        for (int sl = grp; sl < n_slices; sl += 16) acc += partials[(size_t)sl * red_all + e];
    part[grp][el] = acc;
    __syncthreads();
    if (grp != 0 || e >= red_all) return;
    float v = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += part[g][el];
    if (e < red_w3) {
        const int c = e / WN_LD, q = e - c * WN_LD;
        if (c < C) (q < 32 ? gw3[c * 32 + q] : gb3[c]) = v;
    } else if (e < red_w2) {
        const int q = (e - red_w3) / 9, j = (e - red_w3) - q * 9;
        (j < 8 ? gw2[q * 8 + j] : gb2[q]) = v;
    } else {
        const int i = (e - red_w2) >> 2, j = (e - red_w2) & 3;
        (j < 3 ? gw1[i * 3 + j] : gb1[i]) = v;
    }
}

constexpr int WN_BWD_BLOCKS = 512;
inline int wn_red_all(int C) { return 32 * ((C + 31) / 32) * WN_LD + 32 * 9 + 8 * 4; }

}  // namespace

extern "C" int camli_weightnet_fwd(const float* xyz, const float* centres, const int64_t* idx, int idx_stride,
                                   const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                   const float* b3, float* out, int B, int C, int M, int N, int k, int k_major,
                                   void* stream) {
    if (B == 0 || N == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!xyz || !centres || !idx || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out) {
        camli_set_error("camli_weightnet_fwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 0 || k < 1 || idx_stride < k || (long long)N * k > 0x7fffffffLL / 4) {
        camli_set_error("camli_weightnet_fwd: bad shape B=%d C=%d M=%d N=%d k=%d idx_stride=%d", B, C, M, N, k,
                        idx_stride);
        return CAMLI_EINVAL;
    }
    if (C > WN_MAXC) {
        camli_set_error("camli_weightnet_fwd: C=%d > %d output channels not supported", C, WN_MAXC);
        return CAMLI_ENOTSUP;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long tiles = (long long)B * (((long long)N * k + 31) / 32);
    // two workgroups per CU (the register budget allows no more): each wave amortises its W3 fragments over
    // several tiles
    const int blocks = (int)(tiles / 4 + 1 < 512 ? tiles / 4 + 1 : 512);
#define CAMLI_WN_LAUNCH(MT)                                                                                       \
    hipLaunchKernelGGL((weightnet_fwd_kernel<MT>), dim3(blocks), dim3(256), 0, s, xyz, centres, idx, idx_stride, \
                       w1, b1, w2, b2, w3, b3, out, B, C, M, N, k, k_major)
    if (C <= 32) CAMLI_WN_LAUNCH(1);
    else if (C <= 64) CAMLI_WN_LAUNCH(2);
    else if (C <= 96) CAMLI_WN_LAUNCH(3);
    else CAMLI_WN_LAUNCH(4);
#undef CAMLI_WN_LAUNCH
    return camli_check_launch("camli_weightnet_fwd");
}

extern "C" int64_t camli_weightnet_bwd_workspace_bytes(int C) {
    if (C < 1 || C > WN_MAXC) return 0;
    return (int64_t)WN_BWD_BLOCKS * wn_red_all(C) * (int64_t)sizeof(float);
}

extern "C" int camli_weightnet_bwd(const float* xyz, const float* centres, const int64_t* idx, int idx_stride,
                                   const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                   const float* b3, const float* gout, float* gw1, float* gb1, float* gw2, float* gb2,
                                   float* gw3, float* gb3, float* workspace, int64_t workspace_bytes, int B, int C,
                                   int M, int N, int k, int k_major, void* stream) {
    if (!xyz || !centres || !idx || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !gw1 || !gb1 || !gw2 || !gb2 || !gw3 ||
        !gb3 || !workspace || (!gout && B > 0 && N > 0)) {
        camli_set_error("camli_weightnet_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 0 || k < 1 || idx_stride < k || (long long)N * k > 0x7fffffffLL / 4) {
        camli_set_error("camli_weightnet_bwd: bad shape B=%d C=%d M=%d N=%d k=%d idx_stride=%d", B, C, M, N, k,
                        idx_stride);
        return CAMLI_EINVAL;
    }
    if (C > WN_MAXC) {
        camli_set_error("camli_weightnet_bwd: C=%d > %d output channels not supported", C, WN_MAXC);
        return CAMLI_ENOTSUP;
    }
    if (workspace_bytes < camli_weightnet_bwd_workspace_bytes(C)) {
        camli_set_error("camli_weightnet_bwd: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
                        (long long)camli_weightnet_bwd_workspace_bytes(C));
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long tiles = (long long)B * (((long long)N * k + 31) / 32);   // 0 tiles: every slice is zero
    const int blocks = (int)(tiles / 4 + 1 < WN_BWD_BLOCKS ? tiles / 4 + 1 : WN_BWD_BLOCKS);
#define CAMLI_WN_LAUNCH(MT, OCC)                                                                                      \
    hipLaunchKernelGGL((weightnet_bwd_kernel<MT, OCC>), dim3(blocks), dim3(256), 0, s, xyz, centres, idx, idx_stride, \
                       w1, b1, w2, b2, w3, b3, gout, workspace, B, C, M, N, k, k_major)
    // C > 64: 48-64 accumulator registers for dW3 on top of the rest do not fit 256 VGPRs (round 3 shipped the two-waves-
    // per-SIMD build with 87 scratch instructions in its loop); one wave per SIMD, no spills.  CAMLI_WN_BWD_OCC=2: the old build
    const char* occ_env = getenv("CAMLI_WN_BWD_OCC");
    const int occ = occ_env ? atoi(occ_env) : 1;
    if (C <= 32) CAMLI_WN_LAUNCH(1, 2);
    else if (C <= 64) CAMLI_WN_LAUNCH(2, 2);
    else if (C <= 96) { if (occ == 2) CAMLI_WN_LAUNCH(3, 2); else CAMLI_WN_LAUNCH(3, 1); }
    else { if (occ == 2) CAMLI_WN_LAUNCH(4, 2); else CAMLI_WN_LAUNCH(4, 1); }
#undef CAMLI_WN_LAUNCH
    int rc = camli_check_launch("camli_weightnet_bwd");
    if (rc != CAMLI_OK) return rc;
    const int red_all = wn_red_all(C), red_w3 = 32 * ((C + 31) / 32) * WN_LD;
    hipLaunchKernelGGL(weightnet_reduce_kernel, dim3(camli_divup(red_all, 64)), dim3(1024), 0, s, workspace, blocks,
                       red_all, red_w3, red_w3 + 32 * 9, C, gw1, gb1, gw2, gb2, gw3, gb3);
    return camli_check_launch("camli_weightnet_bwd(reduce)");
}

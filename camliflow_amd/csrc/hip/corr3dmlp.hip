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
//         lane = column (level l = lane / 16, neighbour j = lane % 16) -> a wave is one point; layer 1 on the vector ALU
//         with wave-uniform weights (scalar loads), layer 2 on the matrix cores (operands straight from registers,
//         one v_permlane32_swap per K step), the neighbour sum is a 16-lane DPP rotation sum
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

// Orders a wave's own LDS writes before its own later LDS reads of other lanes' slots.  The LDS executes one wave's
// instructions in issue order, so no wait is needed -- only the compiler must not move the reads above the writes.
// (The transposition buffers of the adjoint are private to a wave: a workgroup barrier there would only couple the waves.)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The 1024 + 160 weights are wave-uniform and read-only: loads through the CONSTANT address space are scalar loads
// (s_load_dwordx16, one per half row).  `opaque_zero()` is an offset the optimiser cannot see through: added to the
// pointer inside the per-point loop it keeps the loads inside the loop -- hoisted out of it the 1184 values would have to
// live in vector registers.
typedef const float __attribute__((address_space(4))) * cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_constant(const float* p) { return (cfloat_ptr)(p); }
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

__device__ __forceinline__ void layer1(const float* w1_global, const float* b1_global, const float (&x)[4],
                                       float (&h1)[CM_H]) {
    const int z = opaque_zero();
    const cfloat_ptr w1 = as_constant(w1_global) + z, b1 = as_constant(b1_global) + z;
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

// The lookup's gather folded into the MLP kernels (GATHER = true): the column of lane L = (level L / 16, neighbour L % 16)
// of point n is built in place -- x[0..2] = xyz2[:, m] - xyz1[:, n], x[3] = volume_l[n, m], m = knn_l[n, L % 16] -- instead of
// being read from a materialised [B,4,N,64] tensor (camli_corr3d_gather_levels_fwd: one launch, 16.8 MB written and read
// back per GRU iteration at batch 8); the adjoint adds d/d(volume entry) straight into the persistent gradient volumes
// (a point's 16 neighbours of a level are distinct and launches on one stream are ordered: plain read-modify-write).
// Nested target levels: level l is the first size[l] points of xyz2 [B,3,M0].
struct CmGather {
    const float* xyz1;         // [B,3,N]
    const float* xyz2;         // [B,3,M0]
    const float* vol[4];       // [B,N,size[l]] cost volumes
    const int64_t* knn[4];     // [B,N,16]
    int size[4];
    int m0;
};
struct CmGradVols {
    float* vol[4];             // [B,N,size[l]] gradient volumes, accumulated
};

// One column = (level lane / 16, neighbour lane % 16) of one point.  The gathers of a point form a two-deep dependent chain
// (neighbour index -> coordinates / volume entry) on top of a loop that works through CH points one after the other, so both
// kernels run them as a software pipeline: the index of point p + 2 and the gathers of point p + 1 are in flight while point
// p is on the matrix cores.
struct CmColumn {
    float x[4];
    size_t entry;      // offset of the volume entry inside its level: where the adjoint adds
};
// what a lane needs of its level (lane / 16), picked once per kernel: inside the per-point loop the selects would be
// re-evaluated as vector loads of the kernel-argument fields -- a third link in front of the index -> gather chain
struct CmLane {
    const int64_t* knn;
    const float* vol;
    int size;
};
__device__ __forceinline__ CmLane lane_level(const CmGather& ga, int lane) {
    const int l = lane >> 4;
    const int64_t* k0 = ga.knn[0]; const int64_t* k1 = ga.knn[1]; const int64_t* k2 = ga.knn[2]; const int64_t* k3 = ga.knn[3];
    const float* v0 = ga.vol[0]; const float* v1 = ga.vol[1]; const float* v2 = ga.vol[2]; const float* v3 = ga.vol[3];
    const int s0 = ga.size[0], s1 = ga.size[1], s2 = ga.size[2], s3 = ga.size[3];
    CmLane ll;
    ll.knn = l == 0 ? k0 : l == 1 ? k1 : l == 2 ? k2 : k3;
    ll.vol = l == 0 ? v0 : l == 1 ? v1 : l == 2 ? v2 : v3;
    ll.size = l == 0 ? s0 : l == 1 ? s1 : l == 2 ? s2 : s3;
    return ll;
}
__device__ __forceinline__ int column_index(const CmLane& ll, int b, int n, int N, int lane) {
    return (int)ll.knn[((size_t)b * N + n) * 16 + (lane & 15)];
}
__device__ __forceinline__ void column_gather(const CmGather& ga, const CmLane& ll, int b, int n, int N, int m, CmColumn& col) {
    col.entry = ((size_t)b * N + n) * ll.size + m;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        col.x[a] = ga.xyz2[((size_t)b * 3 + a) * ga.m0 + m] - ga.xyz1[((size_t)b * 3 + a) * N + n];
    col.x[3] = ll.vol[col.entry];
}

// grid ceil(B * N / (4 * CH)), block 256: wave = CH consecutive points of one batch element (N % CH == 0).
// Layer 1 on the vector ALU in the column layout (lane = column, 32 units in registers); layer 2 on the matrix cores:
// per 32-column tile D[o][col] = b2[o] + sum_i W2[o][i] H1[i][col] is 16 v_mfma_f32_32x32x2_f32 (bias in the
// accumulator, then the i-ordered fmaf chain).  The A fragments (W2[o = lane % 32][i = 2s + lane / 32]) are 16 registers
// loaded once per wave; the B fragment of step s wants H1[2s + lane / 32][col = 32 t + lane % 32], and ONE
// v_permlane32_swap of (h1[2s], h1[2s + 1]) yields it for both tiles: [X.lo | Y.lo] and [X.hi | Y.hi].
// D layout: lane holds column 32 t + lane % 32, units o = (r & 3) + 8 (r >> 2) + 4 (lane / 32), r = 0..15.
template <int CH, bool GATHER>
__global__ __launch_bounds__(256) void corr3d_mlp_fwd_kernel(const float* __restrict__ lookup, CmGather ga,
                                                             const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float* __restrict__ out, int B,
                                                             int N) {
    __shared__ float stage[4][CM_OUT][CH + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunks_per_b = N / CH;
    const int chunk = blockIdx.x * 4 + wv;
    const bool live = chunk < B * chunks_per_b;
    const int b = live ? chunk / chunks_per_b : 0, n0 = live ? (chunk % chunks_per_b) * CH : 0;
    const int half = lane >> 5, cl = lane & 31, q = cl >> 4, j = lane & 15;
    const float* __restrict__ xin = GATHER ? nullptr : lookup + ((size_t)b * 4 * N + n0) * CM_COLS + lane;
    const size_t plane = (size_t)N * CM_COLS;

    float wa[CM_H / 2];
    f32x16 bias;
#pragma unroll
    for (int s = 0; s < CM_H / 2; ++s) wa[s] = w2[cl * CM_H + 2 * s + half];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = b2[(r & 3) + 8 * (r >> 2) + 4 * half];

    CmColumn nxt;
    int m_nxt = 0;
    CmLane ll = {};
    if (GATHER) {
        ll = lane_level(ga, lane);
        column_gather(ga, ll, b, n0, N, column_index(ll, b, n0, N, lane), nxt);
        m_nxt = column_index(ll, b, n0 + (CH > 1 ? 1 : 0), N, lane);
    }
    for (int p = 0; p < CH; ++p) {
        float x[4], h1[CM_H];
        if (GATHER) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = nxt.x[c];
            if (p + 1 < CH) column_gather(ga, ll, b, n0 + p + 1, N, m_nxt, nxt);
            m_nxt = column_index(ll, b, n0 + min(p + 2, CH - 1), N, lane);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = xin[c * plane + (size_t)p * CM_COLS];
        }
        layer1(w1, b1, x, h1);
        f32x16 d0 = bias, d1 = bias;
#pragma unroll
        for (int s = 0; s < CM_H / 2; ++s) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(h1[2 * s]), __float_as_uint(h1[2 * s + 1]), false, false);
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], __uint_as_float(sw[0]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], __uint_as_float(sw[1]), d1, 0, 0, 0);
        }
        // ReLU, sum over the 16 neighbours of a level (= one 16-lane row), lane j keeps value r = j
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v0 = row16_sum(fmaxf(d0[r], 0.0f));
            const float v1 = row16_sum(fmaxf(d1[r], 0.0f));
            if (j == r) {
                const int o = (r & 3) + 8 * (r >> 2) + 4 * half;
                stage[wv][q * CM_H + o][p] = v0;             // tile 0: levels 0, 1
                stage[wv][(2 + q) * CM_H + o][p] = v1;       // tile 1: levels 2, 3
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
// partials [gridDim.x][CM_PART].  Per point (= 64 columns) 128 v_mfma_f32_32x32x2_f32:
//   pre2 = b2 + W2 H1            32  (as in the forward; D layout: lane = column 32 t + lane % 32, 16 of the 32 units)
//   G2   = gout where pre2 > 0       (D layout, in place; gout rows read in that layout)
//   gH1  = W2^T G2               32  (B fragments from the D-layout registers: one v_permlane32_swap of registers
//                                     (4m, 4m+1) gives the K steps o = 8m, 8m+1 and o = 8m+4, 8m+5; (4m+2, 4m+3) likewise)
//   G1   = gH1 where h1 > 0          (the layer-1 sign bits travel as one 32-bit mask per column, swapped across the halves)
//   dW2 += G2 H1^T, [dW1|db1] += G1 [X|1]^T   32 + 32  (contractions over the columns: operands transposed through LDS)
constexpr int CM_BW = 2;
__device__ __forceinline__ int unit_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int CH, bool GATHER>
__global__ __launch_bounds__(64 * CM_BW) __attribute__((amdgpu_waves_per_eu(2))) void corr3d_mlp_bwd_kernel(const float* __restrict__ lookup, CmGather ga, CmGradVols gvols, const float* __restrict__ gout,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ glookup, float* __restrict__ partials, int B,
                                                             int N) {
    constexpr int LD = CM_H + 1;
    __shared__ float bufA[CM_BW][CM_COLS][LD];     // G2^T, then G1^T: [column][unit]
    __shared__ float bufB[CM_BW][CM_COLS][LD];     // H1^T
    __shared__ float bufX[CM_BW][CM_COLS][8];      // x0..x3, 1, 0, 0, 0
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunks_per_b = N / CH;
    const int chunk = blockIdx.x * CM_BW + wv;
    const bool live = chunk < B * chunks_per_b;
    const float livef = live ? 1.0f : 0.0f;
    const int b = live ? chunk / chunks_per_b : 0, n0 = live ? (chunk % chunks_per_b) * CH : 0;
    const int half = lane >> 5, cl = lane & 31, q = cl >> 4;
    const size_t plane = (size_t)N * CM_COLS;
    const float* __restrict__ xin = GATHER ? nullptr : lookup + ((size_t)b * 4 * N + n0) * CM_COLS + lane;
    // gout rows in the D layout: tile t -> level 2 t + q, unit unit_of(r, half)
    const float* __restrict__ gin = gout + ((size_t)b * CM_OUT + q * CM_H) * N + n0;

    // The A fragments (W2[o = cl][i = 2s + half] for pre2, W2^T[i = cl][o = 2s + half] for gH1), the bias and W1[:, 3] in
    // the D layout are re-read per point (L1 hits): kept in registers they are 80 of them, and with the five accumulator
    // tiles the kernel then holds one wave per SIMD -- nothing to overlap the matrix cores with.
    f32x16 acc_w2, acc_w1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_w2[r] = acc_w1[r] = 0.0f;
    float gb2 = 0.0f;       // lane (half, cl): sum of g2[o = cl] over the columns [32 half, 32 half + 32) of every point

    CmColumn nxt;
    int m_nxt = 0;
    float gold_nxt = 0.0f;      // the gradient-volume entry the lane adds into travels with the column (read early, written late)
    float* const gbase = !GATHER ? nullptr
                         : (lane >> 4) == 0 ? gvols.vol[0] : (lane >> 4) == 1 ? gvols.vol[1] : (lane >> 4) == 2 ? gvols.vol[2] : gvols.vol[3];
    CmLane ll = {};
    if (GATHER) {
        ll = lane_level(ga, lane);
        column_gather(ga, ll, b, n0, N, column_index(ll, b, n0, N, lane), nxt);
        gold_nxt = live ? gbase[nxt.entry] : 0.0f;
        m_nxt = column_index(ll, b, n0 + (CH > 1 ? 1 : 0), N, lane);
    }
    for (int p = 0; p < CH; ++p) {
        const int z = opaque_zero();
        const float* __restrict__ w2v = w2 + z;
        const float* __restrict__ b2v = b2 + z;
        const float* __restrict__ w1v = w1 + z;
        float x[4], h1[CM_H];
        float* gslot = nullptr;
        float gold = 0.0f;
        if (GATHER) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = nxt.x[c];
            gslot = gbase + nxt.entry;
            gold = gold_nxt;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = xin[c * plane + (size_t)p * CM_COLS];
        }
        layer1(w1, b1, x, h1);
        unsigned mask = 0u;                               // bit i: h1[i] > 0 for this lane's column
#pragma unroll
        for (int i = 0; i < CM_H; ++i) {
            mask |= h1[i] > 0.0f ? (1u << i) : 0u;
            bufB[wv][lane][i] = h1[i];
        }
        const auto msw = __builtin_amdgcn_permlane32_swap(mask, mask, false, false);
        const unsigned m0 = (unsigned)msw[0] >> (4 * half), m1 = (unsigned)msw[1] >> (4 * half);     // columns cl and 32 + cl

        // ---- pre2, then G2 in place ----
        f32x16 d0, d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) d0[r] = d1[r] = b2v[unit_of(r, half)];
#pragma unroll
        for (int s = 0; s < CM_H / 2; ++s) {
            const float wa = w2v[cl * CM_H + 2 * s + half];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(h1[2 * s]), __float_as_uint(h1[2 * s + 1]), false, false);
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, __uint_as_float(sw[0]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, __uint_as_float(sw[1]), d1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = unit_of(r, half);
            const float go0 = gin[(size_t)u * N + p] * livef;
            const float go1 = gin[(size_t)(2 * CM_H + u) * N + p] * livef;
            d0[r] = d0[r] > 0.0f ? go0 : 0.0f;
            d1[r] = d1[r] > 0.0f ? go1 : 0.0f;
            bufA[wv][cl][u] = d0[r];
            bufA[wv][32 + cl][u] = d1[r];
        }
        // ---- gH1 = W2^T G2 ----
        f32x16 e0, e1;
#pragma unroll
        for (int r = 0; r < 16; ++r) e0[r] = e1[r] = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int r0 = 4 * m + 2 * pr, sa = 4 * m + pr, sb = 4 * m + 2 + pr;
                const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0[r0]), __float_as_uint(d0[r0 + 1]), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d1[r0]), __float_as_uint(d1[r0 + 1]), false, false);
                const float wta = w2v[(2 * sa + half) * CM_H + cl], wtb = w2v[(2 * sb + half) * CM_H + cl];
                e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wta, __uint_as_float(s0[0]), e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wta, __uint_as_float(s1[0]), e1, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wtb, __uint_as_float(s0[1]), e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wtb, __uint_as_float(s1[1]), e1, 0, 0, 0);
            }
        }
        wave_lds_fence();
        // dW2[o][i] += sum over the 64 columns of g2[o] * h1[i];  db2[o] += sum of g2[o]
#pragma unroll 8
        for (int s = 0; s < CM_COLS / 2; ++s)
            acc_w2 = __builtin_amdgcn_mfma_f32_32x32x2f32(bufA[wv][2 * s + half][cl], bufB[wv][2 * s + half][cl], acc_w2, 0, 0, 0);
#pragma unroll 8
        for (int s = 0; s < CM_COLS / 2; ++s) gb2 += bufA[wv][32 * half + s][cl];
        wave_lds_fence();
        // ---- layer-1 adjoint (D layout): G1, d/d(cost entry) ----
        float part0 = 0.0f, part1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bit = (r & 3) + 8 * (r >> 2), u = unit_of(r, half);
            const float g10 = (m0 >> bit) & 1u ? e0[r] : 0.0f;
            const float g11 = (m1 >> bit) & 1u ? e1[r] : 0.0f;
            bufA[wv][cl][u] = g10;
            bufA[wv][32 + cl][u] = g11;
            const float w13 = w1v[u * 4 + 3];
            part0 = __builtin_fmaf(w13, g10, part0);
            part1 = __builtin_fmaf(w13, g11, part1);
        }
        {   // a column's 32 units are split over the two lane halves: add the halves, lane L keeps column L
            const auto t0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part0), __float_as_uint(part0), false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part1), __float_as_uint(part1), false, false);
            const float tot0 = __uint_as_float(t0[0]) + __uint_as_float(t0[1]);
            const float tot1 = __uint_as_float(t1[0]) + __uint_as_float(t1[1]);
            const float mine = half == 0 ? tot0 : tot1;
            if (GATHER) {
                if (live) *gslot = gold + mine;
            } else if (live) {
                glookup[((size_t)b * 4 + 3) * plane + (size_t)(n0 + p) * CM_COLS + lane] = mine;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) bufX[wv][lane][c] = x[c];
        bufX[wv][lane][4] = 1.0f;
        bufX[wv][lane][5] = bufX[wv][lane][6] = bufX[wv][lane][7] = 0.0f;
        wave_lds_fence();
        if (GATHER) {
            // the next point's gathers go out HERE: every other global load of the iteration (the per-point re-reads of the
            // weights, L1 hits) is behind us, so no in-order vmcnt wait stands between these and the 32 LDS-fed MFMAs below.
            // Measured (tools/ab_corr3d.py, batch 8, 2048 points): 152 us against 114 us for the kernel fed from a
            // materialised lookup tensor + 19 us gather adjoint + 5 us zero fill: the 15 extra live registers cost the
            // weight re-reads their batching (20 full vmcnt waits per point instead of 5) -- the forward gains 17 us
            if (p + 1 < CH) {
                column_gather(ga, ll, b, n0 + p + 1, N, m_nxt, nxt);
                gold_nxt = live ? gbase[nxt.entry] : 0.0f;
            }
            m_nxt = column_index(ll, b, n0 + min(p + 2, CH - 1), N, lane);
        }
        // [dW1 | db1][i][c] += sum over the columns of g1[i] * [x | 1][c]
#pragma unroll 8
        for (int s = 0; s < CM_COLS / 2; ++s) {
            const float xb = cl < 8 ? bufX[wv][2 * s + half][cl & 7] : 0.0f;
            acc_w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bufA[wv][2 * s + half][cl], xb, acc_w1, 0, 0, 0);
        }
        wave_lds_fence();
    }

    // ---- the waves' tiles -> one partial row per workgroup, added in wave order ----
    __syncthreads();
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
    hipLaunchKernelGGL((corr3d_mlp_fwd_kernel<CM_CH, false>), dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       lookup, CmGather{}, w1, b1, w2, b2, out, B, N);
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
    hipLaunchKernelGGL((corr3d_mlp_bwd_kernel<CM_CH, false>), dim3(blocks), dim3(64 * CM_BW), 0, s, lookup, CmGather{}, CmGradVols{}, gout, w1, b1,
                       w2, b2, glookup, workspace, B, N);
    hipLaunchKernelGGL(corr3d_mlp_reduce_kernel, dim3(camli_divup(CM_PART, 64)), dim3(64, 16), 0, s, workspace, blocks, gw1, gb1, gw2,
                       gb2);
    return camli_check_launch("camli_corr3d_mlp_bwd");
}


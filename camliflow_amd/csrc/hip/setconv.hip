// Depth-wise point set-convolution core ("PointConvDW" gather * weight -> max over neighbours)
// and its adjoint, gfx950.
//
// Replaces the composed tail of models/point_conv.py:122-128 of the reference:
//     features = batch_indexing(features, knn_indices)        # materialises [B,C,N,k]
//     features = features * self.weight_net(knn_offset)       # another [B,C,N,k]
//     features = torch.max(features, dim=-1)[0]
// whose backward scatter-adds a [B,C,N,k] tensor back into [B,C,M] with one atomic per element.
//
//   out[b,c,n] = max_j feat[b,c,idx[b,n,j]] * weight[b,c,n,j]      (first maximum wins ties)
//
// HBM-bound on streaming `weight` once (4*B*C*N*k bytes); the gathered feature rows (M floats per
// (b,c)) stay in L1/L2.  Lane = query point n (coalesced out/arg stores, each lane streams its own
// contiguous k-float weight row with 16-byte loads), 4 channels per thread so the index row is
// read once per 4 channels.  The adjoint touches ONE neighbour per (b,c,n): one float atomic into
// the feature gradient and one plain read-modify-write into the (persistent, caller-zeroed)
// weight-gradient buffer -- k times fewer atomics than the composed path.
#include "camli_common.h"

#include <stdlib.h>
#include <string.h>

namespace {

constexpr int SC_CPT = 4;    // channels per thread
constexpr int SC_CGRP = 4;   // channel groups per block (waves)

// grid (ceil(N/64), ceil(C/16), B), block 256
template <int K>
__global__ __launch_bounds__(256) void pointconv_dw_fwd_kernel(const float* __restrict__ feat,
                                                                const float* __restrict__ weight,
                                                                const int64_t* __restrict__ idx, int idx_stride,
                                                                float* __restrict__ out,
                                                                unsigned char* __restrict__ arg,
                                                                float* __restrict__ wsel, int* __restrict__ msel,
                                                                int C, int M, int N, int k_rt) {
    const int k = K > 0 ? K : k_rt;
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = (blockIdx.y * SC_CGRP + (threadIdx.x >> 6)) * SC_CPT;
    const int b = blockIdx.z;
    if (n >= N || c0 >= C) return;
    const int64_t* __restrict__ irow = idx + ((size_t)b * N + n) * idx_stride;

    float best[SC_CPT];
    int barg[SC_CPT];
    const float* frow[SC_CPT];
    const float* wrow[SC_CPT];
#pragma unroll
    for (int u = 0; u < SC_CPT; ++u) {
        const int c = min(c0 + u, C - 1);
        best[u] = -INFINITY;
        barg[u] = 0;
        frow[u] = feat + ((size_t)b * C + c) * M;
        wrow[u] = weight + (((size_t)b * C + c) * N + n) * (size_t)k;
    }
    if (K > 0 && (K % 4) == 0) {
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += 4) {
            int m[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) m[t] = (int)irow[j0 + t];
#pragma unroll
            for (int u = 0; u < SC_CPT; ++u) {
                const float4 w4 = *reinterpret_cast<const float4*>(wrow[u] + j0);
                const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float p = frow[u][m[t]] * wv[t];
                    const bool gt = p > best[u];
                    best[u] = gt ? p : best[u];
                    barg[u] = gt ? j0 + t : barg[u];
                }
            }
        }
    } else {
        for (int j = 0; j < k; ++j) {
            const int m = (int)irow[j];
#pragma unroll
            for (int u = 0; u < SC_CPT; ++u) {
                const float p = frow[u][m] * wrow[u][j];
                const bool gt = p > best[u];
                best[u] = gt ? p : best[u];
                barg[u] = gt ? j : barg[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SC_CPT; ++u) {
        if (c0 + u < C) {
            const size_t o = ((size_t)b * C + c0 + u) * N + n;
            out[o] = best[u];
            arg[o] = (unsigned char)barg[u];
            if (wsel) {   // compact record for the adjoint: the winning weight and neighbour
                wsel[o] = wrow[u][barg[u]];
                msel[o] = (int)irow[barg[u]];
            }
        }
    }
}

// Forward, tiled form (k in {4,8,16,32}).  grid (ceil(N/64), B), block 256 = 4 waves; every wave
// owns the same 64 points and a quarter of the channels.  The wave streams its [64 x k] weight
// chunk of one channel with lane-CONTIGUOUS 16-byte loads (each cache line is touched once, by one
// instruction), parks it in LDS with a padded row stride (k+4 floats: conflict-free ds_read_b128),
// and every lane then reads back its own k-float row.  The next channel's chunk is already in
// flight in registers while the current one is multiplied / maxed.  The neighbour index row is
// converted to int32 once and lives in VGPRs for all channels.
// ROWLDS (M <= 4096, M % 4 == 0): the feature row of the wave's current channel is staged in wave-private LDS too
// (coalesced 16-byte loads, prefetched with the weight chunk) and the K gathers per lane read LDS instead of L1.
// Measured on the fp32 path: 32 scattered 4-byte global gathers per lane and channel touch ~40 cache lines per wave
// instruction and keep the CU's texture-address path busier than the HBM stream of the weights (2.3-2.7 TB/s cap);
// an LDS gather costs a few bank-conflict cycles.
template <int K, bool ROWLDS>
__global__ __launch_bounds__(256) void pointconv_dw_fwd_tiled_kernel(const float* __restrict__ feat,
                                                                      const float* __restrict__ weight,
                                                                      const int64_t* __restrict__ idx, int idx_stride,
                                                                      float* __restrict__ out,
                                                                      unsigned char* __restrict__ arg,
                                                                      float* __restrict__ wsel, int* __restrict__ msel,
                                                                      int C, int M, int N) {
    constexpr int LD = K + 4;                 // padded LDS row (floats)
    constexpr int V = K / 4;                  // float4 per row == staging loads per lane per channel
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * 64;
    const int n = n0 + lane;
    const bool valid = n < N;
    constexpr int RV = ROWLDS ? 16 : 1;                     // float4 row loads per lane: M <= 4096
    float* tile = lds + w * (64 * LD + (ROWLDS ? M : 0));
    float* rowbuf = tile + 64 * LD;

    int m[K];
    {
        const int64_t* __restrict__ irow = idx + ((size_t)b * N + (valid ? n : N - 1)) * idx_stride;
#pragma unroll
        for (int j = 0; j < K; ++j) m[j] = (int)irow[j];
    }
    const int rows_here = min(64, N - n0);
    const int chunk = rows_here * K;          // floats of this tile's weight chunk per channel

    auto issue = [&](int c, float4 (&st)[V]) {
        const float* __restrict__ src = weight + (((size_t)b * C + c) * N + n0) * K;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int e = (v * 64 + lane) * 4;
            st[v] = e < chunk ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](const float4 (&st)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int e = (v * 64 + lane) * 4;
            const int r = e / K, col = e - r * K;
            *reinterpret_cast<float4*>(tile + r * LD + col) = st[v];
        }
    };

    // blockIdx.z splits the channels: 256 (tile, batch) workgroups alone put ONE wave on each SIMD, which leaves the
    // gather / LDS latency of every channel exposed; CS channel slices per tile give CS waves per SIMD
    const int cstep = 4 * gridDim.z;
    float4 rstage[RV];
    auto issue_row = [&](int c) {
        if (ROWLDS) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(feat + ((size_t)b * C + c) * M);
#pragma unroll
            for (int v = 0; v < RV; ++v) {
                const int e = v * 64 + lane;
                rstage[v] = e * 4 < M ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto park_row = [&]() {
        if (ROWLDS) {
#pragma unroll
            for (int v = 0; v < RV; ++v) {
                const int e = v * 64 + lane;
                if (e * 4 < M) *reinterpret_cast<float4*>(rowbuf + e * 4) = rstage[v];
            }
        }
    };
    float4 stage[V];
    int c = blockIdx.z * 4 + w;
    if (c < C) { issue(c, stage); issue_row(c); }
    for (; c < C; c += cstep) {
        park(stage);                                   // wave-private LDS region: no barrier needed
        park_row();
        if (c + cstep < C) { issue(c + cstep, stage); issue_row(c + cstep); }   // next channel in flight during the compute below
        const float* __restrict__ frow = ROWLDS ? rowbuf : feat + ((size_t)b * C + c) * M;
        float best = -INFINITY;
        int barg = 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 w4 = *reinterpret_cast<const float4*>(tile + lane * LD + v * 4);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float p = frow[m[v * 4 + t]] * wv[t];
                const bool gt = p > best;
                best = gt ? p : best;
                barg = gt ? v * 4 + t : barg;
            }
        }
        if (valid) {
            const size_t o = ((size_t)b * C + c) * N + n;
            out[o] = best;
            arg[o] = (unsigned char)barg;
            if (wsel) {
                wsel[o] = tile[lane * LD + barg];
                int ms = m[0];
#pragma unroll
                for (int j = 1; j < K; ++j) ms = (barg == j) ? m[j] : ms;
                msel[o] = ms;
            }
        }
    }
}

// Forward, shared-row form (round 3).  The tiled kernel above gives every 64-point tile its own workgroup, so the
// feature row of a channel ([M] floats) is staged once per TILE: at N = M = 2048 that is 32 L2 reads of every row,
// 268 MB of L2 -> LDS traffic next to the 134 MB weight stream the kernel exists for (k = 16; PMC: 1.37x the
// algorithmic HBM bytes, and the L2 path busier than HBM).  Here a workgroup is NWV waves = NWV consecutive tiles of
// the SAME batch element walking the SAME channel slice in step: the row of the current channel is staged ONCE per
// workgroup into a double-buffered LDS row (one barrier per channel), every wave gathers from it, and each wave still
// streams its own [64 x K] weight chunk through a wave-private LDS transposition with the next channel's chunk and
// row already in flight in registers.  Row traffic drops by NWV (8x), the index rows are read once per channel slice.
// grid (ceil(N / (64*NWV)), B, CS), block 64*NWV.  dynamic LDS: NWV * 64 * (K+4) floats + 2 * M floats.
template <int K, int NWV, int RVT>
__global__ __launch_bounds__(64 * NWV) void pointconv_dw_fwd_shared_kernel(const float* __restrict__ feat,
                                                                           const float* __restrict__ weight,
                                                                           const int64_t* __restrict__ idx, int idx_stride,
                                                                           float* __restrict__ out,
                                                                           unsigned char* __restrict__ arg,
                                                                           float* __restrict__ wsel, int* __restrict__ msel,
                                                                           int C, int M, int N) {
    constexpr int T = 64 * NWV;
    constexpr int LD = K + 4;
    constexpr int V = K / 4;                  // float4 per weight row; RVT = float4 row loads per thread: M <= 4 * RVT * T
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int b = blockIdx.y;
    const int n0 = (blockIdx.x * NWV + w) * 64;
    const int n = n0 + lane;
    const bool valid = n < N;
    float* tile = lds + w * (64 * LD);
    float* rows = lds + NWV * (64 * LD);      // [2][M]
    const int m4 = M >> 2;

    int m[K];
    {
        const int64_t* __restrict__ irow = idx + ((size_t)b * N + (valid ? n : N - 1)) * idx_stride;
#pragma unroll
        for (int j = 0; j < K; ++j) m[j] = (int)irow[j];
    }
    const int rows_here = n0 < N ? min(64, N - n0) : 0;
    const int chunk = rows_here * K;

    float4 stage[V];
    float4 rstage[RVT];
    auto issue = [&](int c, float4 (&stage)[V], float4 (&rstage)[RVT]) {
        const float* __restrict__ src = weight + (((size_t)b * C + c) * N + n0) * K;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int e = (v * 64 + lane) * 4;
            stage[v] = e < chunk ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4* __restrict__ rsrc = reinterpret_cast<const float4*>(feat + ((size_t)b * C + c) * M);
#pragma unroll
        for (int v = 0; v < RVT; ++v) {
            const int e = v * T + tid;
            rstage[v] = e < m4 ? rsrc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](float* rowbuf, const float4 (&stage)[V], const float4 (&rstage)[RVT]) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int e = (v * 64 + lane) * 4;
            const int r = e / K, col = e - r * K;
            *reinterpret_cast<float4*>(tile + r * LD + col) = stage[v];
        }
#pragma unroll
        for (int v = 0; v < RVT; ++v) {
            const int e = v * T + tid;
            if (e < m4) *reinterpret_cast<float4*>(rowbuf + e * 4) = rstage[v];
        }
    };

    // One register stage.  Measured (tools/kernel_bench.py, C128 k16): a second stage (two chunks in flight per wave) was
    // SLOWER, 54 vs 48 us -- its VGPRs halve the resident workgroups, the bytes in flight per CU stay the same; and with
    // the gathers and all stores compiled out the pipeline still takes 43 us, i.e. what bounds the kernel is the
    // load -> LDS transposition -> compute chain of the <= 16 waves a CU holds, not the gathers or the stores.
    const int cstep = gridDim.z;
    int cur = 0;
    int c = blockIdx.z;
    if (c < C) issue(c, stage, rstage);
    for (; c < C; c += cstep) {
        float* rowbuf = rows + cur * M;
        park(rowbuf, stage, rstage);
        if (c + cstep < C) issue(c + cstep, stage, rstage);
        __syncthreads();      // the row of channel c is complete; the other buffer is free again after this barrier
        float best = -INFINITY;
        int barg = 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 w4 = *reinterpret_cast<const float4*>(tile + lane * LD + v * 4);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float p = rowbuf[m[v * 4 + t]] * wv[t];
                const bool gt = p > best;
                best = gt ? p : best;
                barg = gt ? v * 4 + t : barg;
            }
        }
        if (valid) {
            const size_t o = ((size_t)b * C + c) * N + n;
            out[o] = best;
            arg[o] = (unsigned char)barg;
            if (wsel) {
                wsel[o] = tile[lane * LD + barg];
                int ms = m[0];
#pragma unroll
                for (int j = 1; j < K; ++j) ms = (barg == j) ? m[j] : ms;
                msel[o] = ms;
            }
        }
        cur ^= 1;
    }
}

// Forward, k-major weights [B,C,k,N] (round 3).  With the neighbour slot j as the slow axis a lane's K weights of a
// channel are K perfectly coalesced dword rows: they go from HBM straight into the registers that multiply them -- no
// LDS transposition, no staging registers, and all K loads of the NEXT channel are in flight while the current one is
// reduced (the [B,C,N,k] kernels push every weight through global -> VGPR -> LDS -> VGPR; with gathers and stores
// compiled out that chain alone took 43 of their 48 us).  The weight network writes this layout directly (column
// enumeration with the point index fastest, weightnet.hip), the dense weight gradient is expanded into it.
// Workgroup = NWV waves = NWV consecutive 64-point tiles of one batch element walking one channel slice in step; the
// feature row of the current channel is staged once per workgroup in a double-buffered LDS row (one barrier per
// channel).  grid (ceil(N / (64*NWV)), B, CS), block 64*NWV.  dynamic LDS: 2 * M floats.
template <int K, int NWV, int RVT>
__global__ __launch_bounds__(64 * NWV) void pointconv_dw_fwd_kmajor_kernel(const float* __restrict__ feat,
                                                                           const float* __restrict__ weight,
                                                                           const int* __restrict__ idx_kn,
                                                                           float* __restrict__ out,
                                                                           unsigned char* __restrict__ arg,
                                                                           float* __restrict__ wsel, int* __restrict__ msel,
                                                                           int C, int M, int N) {
    constexpr int T = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [2][M]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int b = blockIdx.y;
    const int n = (blockIdx.x * NWV + w) * 64 + lane;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const int m4 = M >> 2;

    // neighbour table k-major too, int32 [B,K,N]: K coalesced rows.  (Read as int64 rows [B,N,stride] every workgroup
    // pulled 64 cache lines per load instruction -- with the channels split over 16-32 workgroups per tile that was
    // more line traffic than the weights, and it made every variant of this kernel stall at the same 3.5 TB/s.)
    int m[K];
    {
        const int* __restrict__ icol = idx_kn + (size_t)b * K * N + nc;
#pragma unroll
        for (int j = 0; j < K; ++j) m[j] = icol[(size_t)j * N];
    }
    float cur_w[K], nxt_w[K];
    float4 rstage[RVT];
    auto issue = [&](int c, float (&wv)[K], float4 (&rst)[RVT]) {
        const float* __restrict__ src = weight + ((size_t)b * C + c) * K * (size_t)N + nc;
#pragma unroll
        for (int j = 0; j < K; ++j) wv[j] = src[(size_t)j * N];
        const float4* __restrict__ rsrc = reinterpret_cast<const float4*>(feat + ((size_t)b * C + c) * M);
#pragma unroll
        for (int v = 0; v < RVT; ++v) {
            const int e = v * T + tid;
            rst[v] = e < m4 ? rsrc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int cstep = gridDim.z;
    int buf = 0;
    int c = blockIdx.z;
    if (c < C) issue(c, cur_w, rstage);
    for (; c < C; c += cstep) {
        float* rowbuf = rows + buf * M;
#pragma unroll
        for (int v = 0; v < RVT; ++v) {
            const int e = v * T + tid;
            if (e < m4) *reinterpret_cast<float4*>(rowbuf + e * 4) = rstage[v];
        }
        if (c + cstep < C) issue(c + cstep, nxt_w, rstage);     // next channel's K weight rows + feature row in flight
        __syncthreads();      // the row of channel c is complete; the other buffer is free again after this barrier
        float best = -INFINITY;
        int barg = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float p = rowbuf[m[j]] * cur_w[j];
            const bool gt = p > best;
            best = gt ? p : best;
            barg = gt ? j : barg;
        }
        if (valid) {
            const size_t o = ((size_t)b * C + c) * N + n;
            out[o] = best;
            arg[o] = (unsigned char)barg;
            if (wsel) {
                float ws = cur_w[0];
                int ms = m[0];
#pragma unroll
                for (int j = 1; j < K; ++j) {
                    ws = (barg == j) ? cur_w[j] : ws;
                    ms = (barg == j) ? m[j] : ms;
                }
                wsel[o] = ws;
                msel[o] = ms;
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) cur_w[j] = nxt_w[j];
        buf ^= 1;
    }
}

// Adjoint, row form.  One workgroup owns one (b, c) row: the feature row and the row of its
// gradient live in LDS (2*M floats), the scatter is an LDS float atomic, and the finished gradient
// row is written once -- no global atomics, no zero-fill of gfeat.
__global__ __launch_bounds__(256) void pointconv_dw_bwd_row_kernel(const float* __restrict__ gout,
                                                                    const float* __restrict__ feat,
                                                                    const float* __restrict__ wsel,
                                                                    const int* __restrict__ msel,
                                                                    float* __restrict__ gfeat, float* __restrict__ gwsel,
                                                                    int M, int N, int vec, int C, size_t gout_bs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sf = lds;        // feat row
    float* sg = lds + M;    // gradient row
    const size_t row = blockIdx.x;
    // gout may be a channel slice of a wider gradient (batch stride gout_bs floats >= C * N): row (b, c) starts here
    const float* __restrict__ grow = gout + (row / C) * gout_bs + (row % C) * (size_t)N;
    // 16-byte accesses whenever the rows allow it (r3: the scalar form kept one 4-byte load per lane in flight and ran
    // at 2.6 TB/s); `vec`: M and N multiples of 4 and every base pointer 16-byte aligned (checked by the launcher)
    if (vec) {
        const float4* __restrict__ f4 = reinterpret_cast<const float4*>(feat + row * M);
        for (int i = threadIdx.x; i < M / 4; i += 256) {
            reinterpret_cast<float4*>(sf)[i] = f4[i];
            reinterpret_cast<float4*>(sg)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
        for (int i = threadIdx.x; i < M; i += 256) {
            sf[i] = feat[row * M + i];
            sg[i] = 0.0f;
        }
    }
    __syncthreads();
    if (vec) {
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(grow);
        const int4* __restrict__ m4 = reinterpret_cast<const int4*>(msel + row * N);
        const float4* __restrict__ w4 = reinterpret_cast<const float4*>(wsel + row * N);
        float4* __restrict__ o4 = reinterpret_cast<float4*>(gwsel + row * N);
        for (int n = threadIdx.x; n < N / 4; n += 256) {
            const float4 g = g4[n];
            const int4 mm = m4[n];
            if (gwsel) o4[n] = make_float4(g.x * sf[mm.x], g.y * sf[mm.y], g.z * sf[mm.z], g.w * sf[mm.w]);
            if (gfeat) {
                const float4 w = w4[n];
                unsafeAtomicAdd(sg + mm.x, g.x * w.x);
                unsafeAtomicAdd(sg + mm.y, g.y * w.y);
                unsafeAtomicAdd(sg + mm.z, g.z * w.z);
                unsafeAtomicAdd(sg + mm.w, g.w * w.w);
            }
        }
    } else {
        for (int n = threadIdx.x; n < N; n += 256) {
            const size_t e = row * N + n;
            const float g = grow[n];
            const int mm = msel[e];
            if (gwsel) gwsel[e] = g * sf[mm];
            if (gfeat) unsafeAtomicAdd(sg + mm, g * wsel[e]);
        }
    }
    if (gfeat) {
        __syncthreads();
        if (vec) {
            float4* __restrict__ o4 = reinterpret_cast<float4*>(gfeat + row * M);
            for (int i = threadIdx.x; i < M / 4; i += 256) o4[i] = reinterpret_cast<const float4*>(sg)[i];
        } else {
            for (int i = threadIdx.x; i < M; i += 256) gfeat[row * M + i] = sg[i];
        }
    }
}

// Adjoint, row form WITHOUT float atomics (round 4): same traffic, fixed summation order.  The scatter
// gfeat[msel[n]] += gout[n] * wsel[n] has on average one contribution per target and a few targets with several; the LDS
// float atomic above adds those in whatever order the waves arrive (last-bit differences from run to run).  Here the
// entries of a 1024-entry chunk compete for their target with an INTEGER LDS atomic (min over a key that carries the
// entry's position: order-independent), the winner of a target adds its value with a plain read-modify-write, the losers
// try again in the next round -- a target with c contributions takes c rounds and receives them in ascending n, chunk
// after chunk.  Keys carry a round counter in their high bits (later rounds win over stale keys: no reset pass).
__global__ __launch_bounds__(256) void pointconv_dw_bwd_row_ordered_kernel(const float* __restrict__ gout,
                                                                            const float* __restrict__ feat,
                                                                            const float* __restrict__ wsel,
                                                                            const int* __restrict__ msel,
                                                                            float* __restrict__ gfeat, float* __restrict__ gwsel,
                                                                            int M, int N, int vec) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sf = lds;                                              // feat row
    float* sg = lds + M;                                          // gradient row
    unsigned* owner = reinterpret_cast<unsigned*>(lds + 2 * M);   // winning key per target
    const size_t row = blockIdx.x;
    if (vec) {
        const float4* __restrict__ f4 = reinterpret_cast<const float4*>(feat + row * M);
        for (int i = threadIdx.x; i < M / 4; i += 256) {
            reinterpret_cast<float4*>(sf)[i] = f4[i];
            reinterpret_cast<float4*>(sg)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            reinterpret_cast<uint4*>(owner)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        }
    } else {
        for (int i = threadIdx.x; i < M; i += 256) {
            sf[i] = feat[row * M + i];
            sg[i] = 0.0f;
            owner[i] = ~0u;
        }
    }
    __syncthreads();
    unsigned rnd = 0;
    for (int n0 = 0; n0 < N; n0 += 1024) {            // same trip count in every thread (barriers inside)
        float p[4];
        int m[4];
        bool alive[4];
        const int n = n0 + 4 * threadIdx.x;
        if (vec && n + 3 < N) {
            const float4 g = *reinterpret_cast<const float4*>(gout + row * N + n);
            const int4 mm = *reinterpret_cast<const int4*>(msel + row * N + n);
            m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
            if (gwsel) *reinterpret_cast<float4*>(gwsel + row * N + n) = make_float4(g.x * sf[mm.x], g.y * sf[mm.y], g.z * sf[mm.z], g.w * sf[mm.w]);
            float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (gfeat) w = *reinterpret_cast<const float4*>(wsel + row * N + n);
            p[0] = g.x * w.x; p[1] = g.y * w.y; p[2] = g.z * w.z; p[3] = g.w * w.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) alive[e] = gfeat != nullptr;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                alive[e] = false;
                p[e] = 0.0f;
                m[e] = 0;
                if (n + e < N) {
                    const size_t i = row * N + n + e;
                    const float g = gout[i];
                    m[e] = msel[i];
                    if (gwsel) gwsel[i] = g * sf[m[e]];
                    if (gfeat) {
                        p[e] = g * wsel[i];
                        alive[e] = true;
                    }
                }
            }
        }
        if (!gfeat) continue;                          // uniform
        bool any = true;
        while (any) {
            const unsigned hi = (0x3fffffu - rnd) << 10;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (alive[e]) atomicMin(&owner[m[e]], hi | (unsigned)(4 * threadIdx.x + e));
            __syncthreads();
            bool left = false;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (alive[e]) {
                    if (owner[m[e]] == (hi | (unsigned)(4 * threadIdx.x + e))) {
                        sg[m[e]] += p[e];             // the one winner of this target in this round
                        alive[e] = false;
                    } else {
                        left = true;
                    }
                }
            any = __syncthreads_or(left) != 0;
            ++rnd;
        }
    }
    if (gfeat) {
        __syncthreads();
        if (vec) {
            float4* __restrict__ o4 = reinterpret_cast<float4*>(gfeat + row * M);
            for (int i = threadIdx.x; i < M / 4; i += 256) o4[i] = reinterpret_cast<const float4*>(sg)[i];
        } else {
            for (int i = threadIdx.x; i < M; i += 256) gfeat[row * M + i] = sg[i];
        }
    }
}

// Adjoint, compact form.  thread = (b, c, n), n fastest: everything it touches except the feature
// row is a coalesced [B,C,N] stream.
//   gfeat[b,c,msel] += gout * wsel          (float atomic; rows of M floats stay in L2)
//   gwsel[b,c,n]     = gout * feat[b,c,msel] (the ONE non-zero of d/dweight[b,c,n,:], at slot arg)
__global__ __launch_bounds__(256) void pointconv_dw_bwd_kernel(const float* __restrict__ gout,
                                                                const float* __restrict__ feat,
                                                                const float* __restrict__ wsel,
                                                                const int* __restrict__ msel,
                                                                float* __restrict__ gfeat, float* __restrict__ gwsel,
                                                                size_t total, int M, int N) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = e / N;
        const float g = gout[e];
        const int m = msel[e];
        if (gwsel) gwsel[e] = g * feat[bc * M + m];
        if (gfeat) unsafeAtomicAdd(gfeat + bc * M + m, g * wsel[e]);
    }
}

// Dense weight gradient of one pass from the compact records of all its calls:
//   gweight[b,c,n,j] = sum_i [arg_i[b,c,n] == j] * gwsel_i[b,c,n]
// block = 256 consecutive (b,c,n) rows = one contiguous 256*k tile of gweight; rows are accumulated
// in LDS (stride k+1: conflict-free) and the tile is written once, fully coalesced.
constexpr int EX_MAX_CALLS = 64;
struct ExpandCalls {
    const float* gwsel[EX_MAX_CALLS];
    const unsigned char* arg[EX_MAX_CALLS];
};

__global__ __launch_bounds__(256) void pointconv_dw_expand_kernel(ExpandCalls calls, int n_calls,
                                                                   float* __restrict__ gweight, size_t rows, int k) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [256][k+1]
    const int ld = k + 1;
    const size_t row0 = (size_t)blockIdx.x * 256;
    const size_t row = row0 + threadIdx.x;
    float* mine = tile + threadIdx.x * ld;
    for (int j = 0; j < k; ++j) mine[j] = 0.0f;
    if (row < rows) {
        for (int i = 0; i < n_calls; ++i) mine[calls.arg[i][row]] += calls.gwsel[i][row];
    }
    __syncthreads();
    const size_t nrows = rows - row0 < 256 ? rows - row0 : 256;
    const size_t count = nrows * (size_t)k;
    float* __restrict__ dst = gweight + row0 * (size_t)k;
    for (size_t e = threadIdx.x; e < count; e += 256) {
        const int r = (int)(e / k), j = (int)(e - (size_t)r * k);
        dst[e] = tile[r * ld + j];
    }
}

// k-major form of the expansion: gweight [B,C,k,N].  thread = one (b, c, n): it sums its records into k registers and
// writes k coalesced rows (lanes along n); no LDS.
template <int K>
__global__ __launch_bounds__(256) void pointconv_dw_expand_kmajor_kernel(ExpandCalls calls, int n_calls,
                                                                          float* __restrict__ gweight, size_t rows, int N) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0.0f;
    for (int i = 0; i < n_calls; ++i) {
        const int a = calls.arg[i][row];
        const float g = calls.gwsel[i][row];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] += (a == j) ? g : 0.0f;
    }
    const size_t bc = row / N, n = row - bc * N;
    float* __restrict__ dst = gweight + bc * K * (size_t)N + n;
#pragma unroll
    for (int j = 0; j < K; ++j) dst[(size_t)j * N] = acc[j];
}

}  // namespace

extern "C" int camli_pointconv_dw_fwd(const float* feat, const float* weight, const int64_t* idx, int idx_stride,
                                      float* out, unsigned char* arg, float* wsel, int* msel, int B, int C, int M,
                                      int N, int k, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!feat || !weight || !idx || !out || !arg || ((wsel == nullptr) != (msel == nullptr))) {
        camli_set_error("camli_pointconv_dw_fwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 1 || k < 1 || k > 255 || idx_stride < k || B > 65535) {
        camli_set_error("camli_pointconv_dw_fwd: bad shape B=%d C=%d M=%d N=%d k=%d idx_stride=%d", B, C, M, N, k,
                        idx_stride);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CAMLI_DW_TILED(KK)                                                                                         \
    if (rowlds) {                                                                                                   \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pointconv_dw_fwd_tiled_kernel<KK, true>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (64 * (KK + 4) + 4096) * 4);  \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((pointconv_dw_fwd_tiled_kernel<KK, true>), dim3(camli_divup(N, 64), B, cs), dim3(256),   \
                           (size_t)4 * (64 * (KK + 4) + M) * sizeof(float), s, feat, weight, idx, idx_stride, out,   \
                           arg, wsel, msel, C, M, N);                                                               \
    } else {                                                                                                        \
        hipLaunchKernelGGL((pointconv_dw_fwd_tiled_kernel<KK, false>), dim3(camli_divup(N, 64), B, cs), dim3(256),  \
                           (size_t)4 * 64 * (KK + 4) * sizeof(float), s, feat, weight, idx, idx_stride, out, arg,    \
                           wsel, msel, C, M, N);                                                                    \
    }                                                                                                               \
    return camli_check_launch("camli_pointconv_dw_fwd")
    // Measured at batch 8, C = 128, 2048 points (tools/kernel_bench.py): staging the feature row in LDS pays from
    // k = 16 up (k = 32: 116 -> 88 us; k = 4: the row costs more than its four gathers), two channel slices per tile
    // pay for k <= 16 (k = 16: 64 -> 55 us, k = 4: 37 -> 31 us) and cost at k = 32 (LDS-limited residency).
    static const int cs_env = [] { const char* e = getenv("CAMLI_DW_CS"); return e ? atoi(e) : 0; }();
    static const int rowlds_env = [] { const char* e = getenv("CAMLI_DW_ROWLDS"); return e ? atoi(e) : -1; }();
    const bool row_ok = M <= 4096 && (M % 4) == 0 && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0);
    const bool rowlds = row_ok && (rowlds_env >= 0 ? rowlds_env != 0 : k >= 16);
    const int cs = cs_env > 0 ? cs_env : ((k <= 16 && C >= 32) ? 2 : 1);
    // shared-row form: 8 tiles per workgroup share the staged feature row (see the kernel).  Channel slices so that the
    // launch has >= 512 workgroups; CAMLI_DW_FWD=tiled keeps the per-tile kernel (A/B runs).
    static const bool shared_off = [] { const char* e = getenv("CAMLI_DW_FWD"); return e && !strcmp(e, "tiled"); }();
    if (!shared_off && (M % 4) == 0 && M <= 16 * 512 && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0) &&
        (k == 4 || k == 8 || k == 16 || k == 32)) {
        constexpr int NWV = 8;
        const int tiles = camli_divup(N, 64 * NWV);
        int slices = cs_env > 0 ? cs_env : camli_divup(512, tiles * B);
        slices = slices < 1 ? 1 : (slices > C ? C : slices);
        if (slices > 65535) slices = 65535;
        const size_t bytes = ((size_t)NWV * 64 * (k + 4) + 2 * (size_t)M) * sizeof(float);
#define CAMLI_DW_SHARED(KK, RV)                                                                                       \
    {                                                                                                                 \
        static size_t attr_bytes = 0;                                                                                 \
        if (bytes > attr_bytes) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pointconv_dw_fwd_shared_kernel<KK, NWV, RV>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                        \
            attr_bytes = bytes;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((pointconv_dw_fwd_shared_kernel<KK, NWV, RV>), dim3(tiles, B, slices), dim3(64 * NWV),     \
                           bytes, s, feat, weight, idx, idx_stride, out, arg, wsel, msel, C, M, N);                   \
        return camli_check_launch("camli_pointconv_dw_fwd");                                                          \
    }
#define CAMLI_DW_SHARED_K(KK)                   \
    if (M <= 2048) CAMLI_DW_SHARED(KK, 1)       \
    if (M <= 4096) CAMLI_DW_SHARED(KK, 2)       \
    CAMLI_DW_SHARED(KK, 4)
        switch (k) {
            case 4: CAMLI_DW_SHARED_K(4);
            case 8: CAMLI_DW_SHARED_K(8);
            case 16: CAMLI_DW_SHARED_K(16);
            default: CAMLI_DW_SHARED_K(32);
        }
#undef CAMLI_DW_SHARED_K
#undef CAMLI_DW_SHARED
    }
    switch (k) {
        case 4: CAMLI_DW_TILED(4);
        case 8: CAMLI_DW_TILED(8);
        case 16: CAMLI_DW_TILED(16);
        case 32: CAMLI_DW_TILED(32);
        default: break;
    }
#undef CAMLI_DW_TILED
    dim3 grid(camli_divup(N, 64), camli_divup(C, SC_CPT * SC_CGRP), B);
#define CAMLI_DW_LAUNCH(KK)                                                                                    \
    hipLaunchKernelGGL((pointconv_dw_fwd_kernel<KK>), grid, dim3(256), 0, s, feat, weight, idx, idx_stride, out, \
                       arg, wsel, msel, C, M, N, k)
    switch (k) {
        case 4: CAMLI_DW_LAUNCH(4); break;
        case 8: CAMLI_DW_LAUNCH(8); break;
        case 16: CAMLI_DW_LAUNCH(16); break;
        case 32: CAMLI_DW_LAUNCH(32); break;
        default: CAMLI_DW_LAUNCH(0); break;
    }
#undef CAMLI_DW_LAUNCH
    return camli_check_launch("camli_pointconv_dw_fwd");
}

extern "C" int camli_pointconv_dw_fwd_kmajor(const float* feat, const float* weight_kn, const int* idx_kn, float* out,
                                             unsigned char* arg, float* wsel, int* msel, int B, int C, int M, int N, int k,
                                             void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!feat || !weight_kn || !idx_kn || !out || !arg || ((wsel == nullptr) != (msel == nullptr))) {
        camli_set_error("camli_pointconv_dw_fwd_kmajor: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 1 || B > 65535) {
        camli_set_error("camli_pointconv_dw_fwd_kmajor: bad shape B=%d C=%d M=%d N=%d k=%d", B, C, M, N, k);
        return CAMLI_EINVAL;
    }
    if (!((M % 4) == 0 && M <= 16 * 512 && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0) && (k == 4 || k == 8 || k == 16 || k == 32))) {
        camli_set_error("camli_pointconv_dw_fwd_kmajor: needs k in {4,8,16,32}, M %% 4 == 0, M <= 8192, 16-byte aligned feat (M=%d k=%d)", M, k);
        return CAMLI_ENOTSUP;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {
        constexpr int NWV = 8;
        const int tiles = camli_divup(N, 64 * NWV);
        static const int cs_kmajor = [] { const char* e = getenv("CAMLI_DW_CS"); return e ? atoi(e) : 0; }();
        int slices = cs_kmajor > 0 ? cs_kmajor : camli_divup(1024, tiles * B);
        slices = slices < 1 ? 1 : (slices > C ? C : slices);
        if (slices > 65535) slices = 65535;
        const size_t bytes = 2 * (size_t)M * sizeof(float);
#define CAMLI_DW_KMAJOR(KK, RV)                                                                                          \
    {                                                                                                                    \
        hipLaunchKernelGGL((pointconv_dw_fwd_kmajor_kernel<KK, NWV, RV>), dim3(tiles, B, slices), dim3(64 * NWV), bytes, s, \
                           feat, weight_kn, idx_kn, out, arg, wsel, msel, C, M, N);                                      \
        return camli_check_launch("camli_pointconv_dw_fwd_kmajor");                                                      \
    }
#define CAMLI_DW_KMAJOR_K(KK)                   \
    if (M <= 2048) CAMLI_DW_KMAJOR(KK, 1)       \
    if (M <= 4096) CAMLI_DW_KMAJOR(KK, 2)       \
    CAMLI_DW_KMAJOR(KK, 4)
        switch (k) {
            case 4: CAMLI_DW_KMAJOR_K(4);
            case 8: CAMLI_DW_KMAJOR_K(8);
            case 16: CAMLI_DW_KMAJOR_K(16);
            default: CAMLI_DW_KMAJOR_K(32);
        }
#undef CAMLI_DW_KMAJOR_K
#undef CAMLI_DW_KMAJOR
    }
}

static int dw_bwd_impl(const float* gout, const float* feat, const float* wsel, const int* msel, float* gfeat, float* gwsel,
                       int B, int C, int M, int N, bool ordered, void* stream, int64_t gout_bs = 0) {
    if (gout_bs == 0) gout_bs = (int64_t)C * N;
    const bool dense = gout_bs == (int64_t)C * N;
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gout || !feat || !wsel || !msel || (!gfeat && !gwsel)) {
        camli_set_error("camli_pointconv_dw_bwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 1 || M < 1 || N < 1) {
        camli_set_error("camli_pointconv_dw_bwd: bad shape B=%d C=%d M=%d N=%d", B, C, M, N);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t row_lds = (size_t)2 * M * sizeof(float);
    const uintptr_t bits = (uintptr_t)gout | (uintptr_t)feat | (uintptr_t)wsel | (uintptr_t)msel | (uintptr_t)gfeat |
                           (uintptr_t)gwsel;      // a null pointer contributes no bits
    const int vec = ((M | N) & 3) == 0 && (bits & 15) == 0;
    if (!dense && (ordered || row_lds > 64 * 1024 || gout_bs < (int64_t)C * N || (vec && (gout_bs & 3)))) {
        camli_set_error("camli_pointconv_dw_bwd_strided: batch stride %lld is served by the LDS row kernel only (rows of <= 8192 floats, "
                        "stride >= C*N and a multiple of 4)", (long long)gout_bs);
        return CAMLI_ENOTSUP;
    }
    if (ordered) {
        if ((size_t)3 * M * sizeof(float) > 64 * 1024) {
            camli_set_error("camli_pointconv_dw_bwd_ordered: a row of M=%d floats and its two side arrays exceed 64 KB of LDS", M);
            return CAMLI_ENOTSUP;
        }
        hipLaunchKernelGGL(pointconv_dw_bwd_row_ordered_kernel, dim3((unsigned)((size_t)B * C)), dim3(256), (size_t)3 * M * sizeof(float),
                           s, gout, feat, wsel, msel, gfeat, gwsel, M, N, vec);
        return camli_check_launch("camli_pointconv_dw_bwd_ordered");
    }
    if (row_lds <= 64 * 1024) {
        hipLaunchKernelGGL(pointconv_dw_bwd_row_kernel, dim3((unsigned)((size_t)B * C)), dim3(256), row_lds, s, gout, feat,
                           wsel, msel, gfeat, gwsel, M, N, vec, C, (size_t)gout_bs);
        return camli_check_launch("camli_pointconv_dw_bwd");
    }
    // rows too long for LDS: global float atomics into a zero-filled gradient
    if (gfeat && hipMemsetAsync(gfeat, 0, (size_t)B * C * M * sizeof(float), s) != hipSuccess) {
        camli_set_error("camli_pointconv_dw_bwd: memset failed");
        return CAMLI_ELAUNCH;
    }
    const size_t total = (size_t)B * C * N;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(pointconv_dw_bwd_kernel, dim3(blocks), dim3(256), 0, s, gout, feat, wsel, msel, gfeat, gwsel, total,
                       M, N);
    return camli_check_launch("camli_pointconv_dw_bwd");
}

extern "C" int camli_pointconv_dw_bwd(const float* gout, const float* feat, const float* wsel, const int* msel,
                                      float* gfeat, float* gwsel, int B, int C, int M, int N, void* stream) {
    return dw_bwd_impl(gout, feat, wsel, msel, gfeat, gwsel, B, C, M, N, false, stream);
}

// The same adjoint without float atomics: fixed summation order, bit-reproducible (1.7x the time of the LDS-atomic form at
// C128 k16: 37.6 vs 21.7 us); what torch.use_deterministic_algorithms(True) selects.  M <= 5461.
// gout read in place from a channel slice of a wider gradient: batch stride in floats (>= C*N, a multiple of 4 when M and N
// are; rows of M <= 8192 floats -- the LDS row kernel); CAMLI_ENOTSUP otherwise (copy the slice and call camli_pointconv_dw_bwd)
extern "C" int camli_pointconv_dw_bwd_strided(const float* gout, int64_t gout_batch_stride, const float* feat, const float* wsel,
                                              const int* msel, float* gfeat, float* gwsel, int B, int C, int M, int N,
                                              void* stream) {
    if (gout_batch_stride < 1) { camli_set_error("camli_pointconv_dw_bwd_strided: bad stride"); return CAMLI_EINVAL; }
    return dw_bwd_impl(gout, feat, wsel, msel, gfeat, gwsel, B, C, M, N, false, stream, gout_batch_stride);
}

extern "C" int camli_pointconv_dw_bwd_ordered(const float* gout, const float* feat, const float* wsel, const int* msel,
                                              float* gfeat, float* gwsel, int B, int C, int M, int N, void* stream) {
    return dw_bwd_impl(gout, feat, wsel, msel, gfeat, gwsel, B, C, M, N, true, stream);
}

extern "C" int camli_pointconv_dw_expand(const float* const* gwsel_list, const unsigned char* const* arg_list,
                                         int n_calls, float* gweight, int B, int C, int N, int k, int k_major,
                                         void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gwsel_list || !arg_list || !gweight) {
        camli_set_error("camli_pointconv_dw_expand: null pointer");
        return CAMLI_EINVAL;
    }
    if (n_calls < 0 || n_calls > EX_MAX_CALLS || B < 0 || C < 1 || N < 1 || k < 1 || k > 255) {
        camli_set_error("camli_pointconv_dw_expand: bad arguments n_calls=%d (max %d) B=%d C=%d N=%d k=%d", n_calls,
                        EX_MAX_CALLS, B, C, N, k);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    ExpandCalls calls;
    for (int i = 0; i < n_calls; ++i) {
        if (!gwsel_list[i] || !arg_list[i]) {
            camli_set_error("camli_pointconv_dw_expand: call %d has a null record", i);
            return CAMLI_EINVAL;
        }
        calls.gwsel[i] = gwsel_list[i];
        calls.arg[i] = arg_list[i];
    }
    const size_t rows = (size_t)B * C * N;
    if (k_major) {
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const unsigned blocks = (unsigned)((rows + 255) / 256);
        switch (k) {
            case 4: hipLaunchKernelGGL(pointconv_dw_expand_kmajor_kernel<4>, dim3(blocks), dim3(256), 0, st, calls, n_calls, gweight, rows, N); break;
            case 8: hipLaunchKernelGGL(pointconv_dw_expand_kmajor_kernel<8>, dim3(blocks), dim3(256), 0, st, calls, n_calls, gweight, rows, N); break;
            case 16: hipLaunchKernelGGL(pointconv_dw_expand_kmajor_kernel<16>, dim3(blocks), dim3(256), 0, st, calls, n_calls, gweight, rows, N); break;
            case 32: hipLaunchKernelGGL(pointconv_dw_expand_kmajor_kernel<32>, dim3(blocks), dim3(256), 0, st, calls, n_calls, gweight, rows, N); break;
            default: camli_set_error("camli_pointconv_dw_expand: k-major needs k in {4,8,16,32}, got %d", k); return CAMLI_ENOTSUP;
        }
        return camli_check_launch("camli_pointconv_dw_expand");
    }
    const size_t lds = (size_t)256 * (k + 1) * sizeof(float);
    hipLaunchKernelGGL(pointconv_dw_expand_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), calls, n_calls, gweight, rows, k);
    return camli_check_launch("camli_pointconv_dw_expand");
}

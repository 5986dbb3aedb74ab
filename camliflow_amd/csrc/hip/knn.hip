// Brute-force k-nearest-neighbour for 2-D / 3-D points, gfx950.
//
// Replaces models/csrc/k_nearest_neighbor/k_nearest_neighbor_kernel.cu:9-113 of the reference
// (one thread per query, 64-slot local arrays that spill).  Results are bit-identical to the
// reference kernel's sequential insertion semantics (oracle/camli_oracle.c: oracle_knn), ties
// included.
//
// Design (CDNA4):
//   * one lane per query, 64 queries per wave; the candidate coordinates are wave-uniform, so
//     they arrive through the scalar cache (s_load) and cost no VALU/LDS traffic
//   * the k-entry sorted list of every lane lives in VGPRs (K is a template parameter, the
//     insertion network is fully unrolled; no scratch)
//   * a candidate that passes the (stale) k-th-distance test is appended to a small per-lane
//     LDS queue; the 5*K-op insertion network only runs when some lane's queue is full, and
//     then drains every lane's queue in candidate order -> exact sequential semantics while the
//     wave-level "some lane accepts" branch stops dominating (64 independent queries per wave
//     make that branch nearly always taken otherwise)
//   * for small batches the candidate range is split over the NW waves of a workgroup; the NW
//     partial lists are merged through LDS.  A merge is only order-exact when no candidate that
//     ties the final k-th distance was dropped anywhere; that is detected from the minimum
//     evicted distance and those (rare: exact duplicates only) queries are recomputed by a
//     single-wave in-order scan.
//   * splitting multiplies the insertions: a wave that sees M/NW candidates accepts ~k(1 + ln(M/(NW k))) of them, so 8
//     waves insert 8 x 82 candidates per query where one in-order scan inserts 116 (k = 16, M = 8192) -- the insertion
//     network, not the distance arithmetic, then dominates.  Cure: every wave publishes (per lane, in LDS, at each
//     drain) the distance at position k/G - 1 of its list, G = the number of waves whose union the result is taken
//     over.  G waves x k/G entries are k distinct candidates, so the MAXIMUM of the published values bounds the final
//     k-th distance of the union from above -- and it is ~G x tighter than the wave's own k-th distance.  A candidate
//     is queued only if it passes min(own k-th, that bound).  Published values only decrease, so a stale read is still
//     a valid bound: no barrier is needed, and exactness is untouched (only candidates strictly farther than the final
//     k-th distance are skipped; evictions are still tracked for the tie test).
#include "camli_common.h"

#include <stdlib.h>

namespace {

#ifndef KNN_UNROLL
#define KNN_UNROLL 8
#endif
constexpr int QBUF = KNN_UNROLL + 8;          // per-lane queue depth of accepted-but-not-inserted candidates
constexpr float KNN_INIT = 1e9f;  // k_nearest_neighbor_kernel.cu:27-30,70-73

template <int D>
__device__ __forceinline__ float sqdist(float ux, float uy, float uz, const float* __restrict__ p) {
    float x = p[0], y = p[1];
    float d = (ux - x) * (ux - x) + (uy - y) * (uy - y);
    if (D == 3) {
        float z = p[2];
        d = d + (uz - z) * (uz - z);
    }
    return d;
}

// Insert (d, i) into the ascending list; precondition !(d > dist[K-1]).  Lands after every
// entry with dist <= d, drops the old last entry (k_nearest_neighbor_kernel.cu:82-90).
template <int K>
__device__ __forceinline__ void list_insert(float (&dist)[K], int (&idx)[K], float d, int i, float& ev_min) {
    ev_min = fminf(ev_min, dist[K - 1]);
    bool prev = true;  // "the slot to the right shifted" (true for the virtual slot K)
#pragma unroll
    for (int j = K - 1; j >= 1; --j) {
        const bool sh = dist[j - 1] > d;
        // distances: clamp(d, dist[j-1], dist[j]) == the shifted / placed / kept value (one v_med3)
        const float nd = (j == K - 1) ? fmaxf(dist[j - 1], d) : __builtin_amdgcn_fmed3f(dist[j - 1], dist[j], d);
        // indices: two plain selects (kept un-nested so they lower to v_cndmask, not branches)
        const int keep = prev ? i : idx[j];
        const int ni = sh ? idx[j - 1] : keep;
        dist[j] = nd;
        idx[j] = ni;
        prev = sh;
    }
    dist[0] = fminf(dist[0], d);
    idx[0] = prev ? i : idx[0];
}

// In-order scan of candidates [lo, hi) for the lane's query.  lo/hi are wave-uniform, so the
// candidate coordinates come in through scalar loads; four candidates are fetched per trip to
// keep a batch of s_loads in flight.
// Walk candidates [lo, hi) in trips of U: the U*D coordinates of a trip are consecutive floats fetched as ONE run of
// scalar loads from a single base address (per-candidate address arithmetic keeps the compiler from merging them),
// and the next trip's run is requested before the current one is consumed, so the scalar-cache latency overlaps the
// distance arithmetic.  visit(d, c) per candidate in index order, after_trip() once per trip.
template <int D, typename Visit, typename AfterTrip>
__device__ __forceinline__ void for_each_candidate(const float* __restrict__ in_b, int lo, int hi, float ux, float uy,
                                                   float uz, Visit&& visit, AfterTrip&& after_trip) {
    constexpr int U = KNN_UNROLL;
    int c = lo;
    const int trips = (hi - lo) / U;
    float cur[U * D], nxt[U * D];
    if (trips > 0) {
        const float* __restrict__ p = in_b + (size_t)lo * D;
#pragma unroll
        for (int i = 0; i < U * D; ++i) cur[i] = p[i];
    }
    for (int t = 0; t < trips; ++t, c += U) {
        if (t + 1 < trips) {
            const float* __restrict__ p = in_b + (size_t)(c + U) * D;
#pragma unroll
            for (int i = 0; i < U * D; ++i) nxt[i] = p[i];
        }
        float dd[U];       // plain fp32 ops: pairing candidates on v_pk_add/mul_f32 measured 8-11 % SLOWER (dependent
                           // packed ops need extra wait states) although it issues 5 instead of 8 instructions
#pragma unroll
        for (int u = 0; u < U; ++u) dd[u] = sqdist<D>(ux, uy, uz, cur + u * D);
#pragma unroll
        for (int u = 0; u < U; ++u) visit(dd[u], c + u);
        after_trip();
#pragma unroll
        for (int i = 0; i < U * D; ++i) cur[i] = nxt[i];
    }
    for (; c < hi; ++c) {
        visit(sqdist<D>(ux, uy, uz, in_b + (size_t)c * D), c);
        after_trip();
    }
}

// `slots` (may be null): the published bounds, [4][nw][64] floats, row g holds position (K >> (g+1)) - 1 of every wave's
// list; `group` = G above (1: this wave takes no bound from the others, it still publishes).
constexpr int KNN_SLOT_ROWS = 4;
template <int K>
__device__ __forceinline__ float list_entry_for_group(const float (&dist)[K], int g) {
    // position K/2 - 1, K/4 - 1, K/8 - 1, K/16 - 1 (clamped to 0 when K is smaller than the divisor)
    constexpr int P0 = K / 2 >= 1 ? K / 2 - 1 : 0, P1 = K / 4 >= 1 ? K / 4 - 1 : 0, P2 = K / 8 >= 1 ? K / 8 - 1 : 0,
                  P3 = K / 16 >= 1 ? K / 16 - 1 : 0;
    return g == 0 ? dist[P0] : (g == 1 ? dist[P1] : (g == 2 ? dist[P2] : dist[P3]));
}

template <int D, int K>
__device__ __forceinline__ void scan_range(const float* __restrict__ in_b, int lo, int hi, float ux, float uy,
                                           float uz, float (&dist)[K], int (&idx)[K], float& ev_min,
                                           float* __restrict__ qd, int* __restrict__ qi, int qstride,
                                           volatile float* slots = nullptr, int nw = 1, int w = 0, int group = 1,
                                           float cap = INFINITY) {
    // `cap` (K >= 8 path only): an upper bound of the k-th distance of every result this list feeds, known to the caller
    // (camli_knn_prefixes_prior); candidates beyond it never reach those results and are not even queued.
    constexpr int U = KNN_UNROLL;
    if (K == 1) {
        float best = dist[0];
        int bi = idx[0];
        auto visit = [&](float d, int c) {
            const bool take = !(d > best);  // a later candidate at equal distance replaces (kernel.cu:37)
            best = take ? d : best;
            bi = take ? c : bi;
        };
        for_each_candidate<D>(in_b, lo, hi, ux, uy, uz, visit, [] {});
        dist[0] = best;
        idx[0] = bi;
    } else if (K <= 4) {
        for_each_candidate<D>(in_b, lo, hi, ux, uy, uz,
                              [&](float d, int c) { if (!(d > dist[K - 1])) list_insert<K>(dist, idx, d, c, ev_min); }, [] {});
    } else {
        int cnt = 0;
        float thr = fminf(dist[K - 1], cap);
        auto drain = [&]() {
#pragma nounroll
            for (int t = 0; t < QBUF; ++t) {
                bool live = t < cnt;
                if (!__ballot(live)) break;
                if (live) {
                    float d = qd[t * qstride];
                    int c = qi[t * qstride];
                    if (!(d > dist[K - 1])) list_insert<K>(dist, idx, d, c, ev_min);
                }
            }
            cnt = 0;
            thr = fminf(dist[K - 1], cap);
            if (slots) {
                const int lane = threadIdx.x & 63;
#pragma unroll
                for (int g = 0; g < KNN_SLOT_ROWS; ++g)
                    if ((K >> (g + 1)) >= 1 && (2 << g) <= nw) slots[(g * nw + w) * 64 + lane] = list_entry_for_group<K>(dist, g);
                if (group > 1) {
                    const int g = group == 2 ? 0 : (group == 4 ? 1 : (group == 8 ? 2 : 3));
                    float bnd = slots[(g * nw) * 64 + lane];
                    for (int o = 1; o < group; ++o) bnd = fmaxf(bnd, slots[(g * nw + o) * 64 + lane]);
                    thr = fminf(thr, bnd);
                }
            }
        };
        auto enqueue = [&](float d, int c) {
            if (!(d > thr)) {
                qd[cnt * qstride] = d;
                qi[cnt * qstride] = c;
                ++cnt;
            }
        };
        for_each_candidate<D>(in_b, lo, hi, ux, uy, uz, enqueue, [&] { if (__ballot(cnt > QBUF - U)) drain(); });
        drain();
    }
}

// Merge of two ascending k-lists into the k smallest of their union, as a bitonic network (round 3; the serial form
// inserted the other list's entries one by one: 16 x 80 dependent VALU ops per merged list, on ONE wave while the
// others had already retired -- the tail was ~1/3 of a split search).  `dist/idx` = the list of the EARLIER candidate
// range, `od/oi` the later one.  Half-cleaner against the reversed second list keeps the k smaller of each pair (a tie
// keeps the earlier range's entry), then log2(k) compare-exchange stages order them by (distance, index) -- the order
// sequential insertion produces for entries that stay inside the list.  Every dropped distance goes into ev_min: a
// dropped candidate that ties the final k-th distance is the one case where the in-order result can differ, and it is
// detected from ev_min == dist[K-1] exactly as before.  K must be a power of two.
template <int K>
__device__ __forceinline__ void merge_topk(float (&dist)[K], int (&idx)[K], const float (&od)[K], const int (&oi)[K],
                                           float& ev_min) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float bd = od[K - 1 - i];
        const int bi = oi[K - 1 - i];
        const bool take = bd < dist[i];
        ev_min = fminf(ev_min, take ? dist[i] : bd);
        dist[i] = take ? bd : dist[i];
        idx[i] = take ? bi : idx[i];
    }
#pragma unroll
    for (int s = K / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if ((i & s) == 0) {
                const float a = dist[i], b = dist[i + s];
                const int ai = idx[i], bi = idx[i + s];
                const bool sw = (b < a) || (b == a && bi < ai);
                dist[i] = sw ? b : a;
                dist[i + s] = sw ? a : b;
                idx[i] = sw ? bi : ai;
                idx[i + s] = sw ? ai : bi;
            }
        }
    }
}

namespace xl {
template <int D>
__device__ void redo_query(const float* __restrict__ in_b, int M, float ux, float uy, float uz, int k,
                           int64_t* __restrict__ o, int lane, float bound = 1e9f /* KNN_INIT: no bound */);
}

// The queries of a wave whose merged list may differ from the in-order result (`redo` lanes), one after the other, each by
// the WHOLE wave (xl::redo_query: 64 candidates per trip, the reference's insertion rule; a few microseconds for 2048
// candidates).  Round 5: until then the tied lanes re-ran the lane-per-query scan over the whole range with the other lanes
// idle -- 130 us for ONE tied query of a 2048-candidate search, during which the kernel's other 255 workgroups had long
// finished: a single tie among 16,384 queries took a 59 us launch to 190 us (profiles/r05_experiments.txt 11).
// `o_wave` = the output row of the wave's lane 0 (query q0), rows are K entries apart; results are stored by redo_query.
// `kth` = the lane's merged k-th distance: the true k-th smallest distance (a merge only errs in WHICH tied index it keeps),
// so candidates beyond it never reach the final list and the redo only inserts the <= k + ties candidates within it.
template <int D, int K>
__device__ __forceinline__ void redo_tied_queries(uint64_t tied, const float* __restrict__ in_b, int M, float ux, float uy,
                                                  float uz, float kth, int64_t* __restrict__ o_wave, int lane) {
    while (tied) {
        const int src = (int)__builtin_ctzll(tied);
        tied &= tied - 1;
        const float bound = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kth), src));
        const float qx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ux), src));
        const float qy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uy), src));
        const float qz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uz), src));
        xl::redo_query<D>(in_b, M, qx, qy, qz, K, o_wave + (size_t)src * K, lane, bound);
    }
}

// grid (ceil(Nq/64), B); block 64*NW threads.  LDS (dynamic):
//   queue region : 2 * QBUF * 64*NW dwords            (K >= 8 only)
//   merge region : NW * K * 64 * 2 dwords + NW*64     (NW > 1 only)   -- the two regions alias
template <int D, int K>
__global__ __launch_bounds__(K >= 32 ? 256 : (K >= 16 ? 512 : 1024)) void knn_kernel(const float* __restrict__ input, const float* __restrict__ query,
                                                    int64_t* __restrict__ out, int M, int Nq, int share) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nthreads = blockDim.x;
    const int NW = nthreads >> 6;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int q_raw = blockIdx.x * 64 + lane;
    const int q = q_raw < Nq ? q_raw : Nq - 1;

    const float* __restrict__ in_b = input + (size_t)b * M * D;
    const float* qp = query + ((size_t)b * Nq + q) * D;
    const float ux = qp[0], uy = qp[1], uz = (D == 3) ? qp[2] : 0.0f;

    float dist[K];
    int idx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        dist[j] = KNN_INIT;
        idx[j] = 0;
    }
    float ev_min = INFINITY;

    float* qd = smem + threadIdx.x;
    int* qi = reinterpret_cast<int*>(smem) + QBUF * nthreads + threadIdx.x;

    const int lo = (int)(((long long)w * M) / NW);
    const int hi = (int)(((long long)(w + 1) * M) / NW);
    // published bounds (K >= 8, NW > 1, at least one list entry per wave of the group): after the queue region
    volatile float* slots = nullptr;
    if (K >= 8 && NW > 1 && K / NW >= 1 && share) {
        slots = smem + 2 * QBUF * nthreads;
#pragma unroll
        for (int g = 0; g < KNN_SLOT_ROWS; ++g) slots[(g * NW + w) * 64 + lane] = KNN_INIT;
        __syncthreads();
    }
    scan_range<D, K>(in_b, lo, hi, ux, uy, uz, dist, idx, ev_min, qd, qi, nthreads, slots, NW, w, NW);

    uint64_t tied = 0;      // queries of wave 0 to be redone in order (below)
    if (NW > 1) {
        __syncthreads();  // queues are dead; the merge region aliases them
        float* md = smem;                                          // [NW][K][64]
        int* mi = reinterpret_cast<int*>(smem) + NW * K * 64;      // [NW][K][64]
        float* mev = smem + 2 * NW * K * 64;                       // [NW][64]
        if (K >= 4 && (K & (K - 1)) == 0) {
            // binary tree over the waves (NW is a power of two): round `step` merges list w + step into list w for
            // every w that is a multiple of 2*step -- log2(NW) bitonic merges on the critical path instead of NW - 1
            // serial ones.  The ranges stay in candidate order (w's range precedes w + step's).
            for (int step = 1; step < NW; step <<= 1) {
                if ((w & (2 * step - 1)) == step) {        // this wave's list is consumed in this round
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        md[(w * K + j) * 64 + lane] = dist[j];
                        mi[(w * K + j) * 64 + lane] = idx[j];
                    }
                    mev[w * 64 + lane] = ev_min;
                }
                __syncthreads();
                if ((w & (2 * step - 1)) == 0 && w + step < NW) {
                    float od[K];
                    int oi[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        od[j] = md[((w + step) * K + j) * 64 + lane];
                        oi[j] = mi[((w + step) * K + j) * 64 + lane];
                    }
                    ev_min = fminf(ev_min, mev[(w + step) * 64 + lane]);
                    merge_topk<K>(dist, idx, od, oi, ev_min);
                }
            }
            if (w > 0) return;
        } else {
        if (w > 0) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                md[(w * K + j) * 64 + lane] = dist[j];
                mi[(w * K + j) * 64 + lane] = idx[j];
            }
            mev[w * 64 + lane] = ev_min;
        }
        __syncthreads();
        if (w > 0) return;
        for (int s = 1; s < NW; ++s) {
            ev_min = fminf(ev_min, mev[s * 64 + lane]);
            for (int j = 0; j < K; ++j) {
                float d = md[(s * K + j) * 64 + lane];
                int c = mi[(s * K + j) * 64 + lane];
                bool acc = !(d > dist[K - 1]);
                if (!__ballot(acc)) break;
                if (acc) {
                    if (K == 1) {
                        dist[0] = d;
                        idx[0] = c;
                    } else {
                        list_insert<K>(dist, idx, d, c, ev_min);
                    }
                }
            }
        }
        }
        if (K > 1) {
            // a candidate tying the final k-th distance was dropped somewhere: the merged list
            // may differ from the in-order result in WHICH tied index it keeps -> those queries are redone in order
            tied = __ballot((ev_min == dist[K - 1]) && q_raw < Nq);
        }
    }

    if (q_raw < Nq && !((tied >> lane) & 1ull)) {
        int64_t* o = out + ((size_t)b * Nq + q_raw) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) o[j] = (int64_t)idx[j];
    }
    if (tied) redo_tied_queries<D, K>(tied, in_b, M, ux, uy, uz, dist[K - 1], out + ((size_t)b * Nq + blockIdx.x * 64) * K, lane);
}

// Nested candidate prefixes in ONE scan.  The point-cloud pyramid is a chain of FPS prefixes (level l+1 = the first
// n_{l+1} points of level l, models/utils.py:121-125), and the reference searches every level separately
// (camliraft_l_core.py:62-66 called four times per GRU iteration).  The insertion semantics are sequential in the
// candidate index, so the k-list after the first M_l candidates IS the answer for level l: the candidate range is cut
// into NW = M_0 / chunk equal pieces (chunk = the smallest level), one wave each, and wave 0 merges the partial lists
// in index order, writing a snapshot whenever the merged prefix reaches a level size.  Ties at the k-th distance are
// detected per snapshot exactly as in knn_kernel and those queries are redone by an in-order scan of that prefix.
struct KnnPrefixOut {
    int64_t* out[4];
    const int64_t* prior[4];    // optional (all or none): an earlier result of the same search, queries and / or candidates moved since
    int size[4];        // descending candidate counts, size[0] = M_0; unused entries 0
    int levels;
};

// -DCAMLI_KNN_PROFILE (tools/microbench/knn_prefix_mb.hip only): shader-clock stamps around the phases of the prefix search,
// per wave of workgroup (0, 0): [0] set-up, [1] scan of the own chunk, [2] merge rounds (barriers included), [3] snapshots
// (in-order rescans of tied queries + stores), [4] rescans started, [5] drains (scan_range), [6] insertions executed
#ifdef CAMLI_KNN_PROFILE
__device__ unsigned long long camli_knn_prof[16][8];
__device__ unsigned camli_knn_wg_ticks[64][64][4];      // [blockIdx.y][blockIdx.x]: wave 0's scan / merge / snapshot ticks, rescans
#define KNN_STAMP(k)                                                          \
    {                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
        prof_[k] += now_ - t_;                                                \
        t_ = now_;                                                            \
    }
#define KNN_COUNT(k, n) prof_[k] += (unsigned long long)(n);
#else
#define KNN_STAMP(k)
#define KNN_COUNT(k, n)
#endif

template <int D, int K>
__global__ __launch_bounds__(K >= 32 ? 256 : 512) void knn_prefix_kernel(const float* __restrict__ input,
                                                                          const float* __restrict__ query,
                                                                          KnnPrefixOut po, int M, int Nq, int chunk,
                                                                          int share) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nthreads = blockDim.x;
    const int NW = nthreads >> 6;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int q_raw = blockIdx.x * 64 + lane;
    const int q = q_raw < Nq ? q_raw : Nq - 1;
    const float* __restrict__ in_b = input + (size_t)b * M * D;
    const float* qp = query + ((size_t)b * Nq + q) * D;
    const float ux = qp[0], uy = qp[1], uz = (D == 3) ? qp[2] : 0.0f;

#ifdef CAMLI_KNN_PROFILE
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_ = __builtin_amdgcn_s_memtime();
#endif
    float dist[K];
    int idx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        dist[j] = KNN_INIT;
        idx[j] = 0;
    }
    float ev_min = INFINITY;
    float* qd = smem + threadIdx.x;
    int* qi = reinterpret_cast<int*>(smem) + QBUF * nthreads + threadIdx.x;
    // bound group of this wave: the chunks of the SMALLEST level that contains chunk w (the wave's list only ever
    // contributes to levels at least that large, and k candidates among the group's chunks bound every one of them)
    int group = NW;
    for (int l = 0; l < po.levels; ++l) {
        const int sl = po.size[l] / chunk;
        if (sl > w && sl < group) group = sl;
    }
    if (!(group == 2 || group == 4 || group == 8 || group == 16) || K / group < 1) group = 1;
    volatile float* slots = nullptr;
    if (NW > 1 && share) {
        slots = smem + 2 * QBUF * nthreads;
#pragma unroll
        for (int g = 0; g < KNN_SLOT_ROWS; ++g) slots[(g * NW + w) * 64 + lane] = KNN_INIT;
    }
    // An earlier result of the same search (the GRU iterations search the same clouds, moved a little, twelve times): its
    // K candidates of level l are K different candidates of that prefix, so the largest of their distances NOW is an upper
    // bound of the level's k-th distance now -- and the largest such bound over the levels this wave's chunk belongs to caps
    // what the wave ever needs to queue.  With it the scan keeps ~K / NW + a few candidates per lane instead of
    // ~K (1 + ln(chunk / K)).  Same arithmetic as the scan (sqdist), so the prior candidates themselves pass the cap; an
    // index outside the level, or one that occurs twice, switches the cap off.  Wave l evaluates level l for the 64 queries (2 K scattered reads per
    // lane: with every wave doing its own levels the CU's address unit took 15 kiloticks over it) and leaves it in LDS.
    float cap = INFINITY;
    float* level_cap = smem + 2 * QBUF * nthreads + KNN_SLOT_ROWS * NW * 64;       // [levels][64], behind the slots
    const bool bounded = po.prior[0] != nullptr && NW >= po.levels;
    if (bounded && w < po.levels) {
        const int l = w;
        const int64_t* __restrict__ pr = po.prior[l] + ((size_t)b * Nq + q) * K;
        long long c[K];
        float px[K], py[K], pz[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = pr[j];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool in = c[j] >= 0 && c[j] < po.size[l];
            ok = ok && in;
            const float* __restrict__ pt = in_b + (size_t)(in ? c[j] : 0) * D;
            px[j] = pt[0];
            py[j] = pt[1];
            pz[j] = D == 3 ? pt[2] : 0.0f;
        }
        // K DIFFERENT candidates bound the k-th distance; a prior that repeats an index (a zeros placeholder, say) does not:
        // such a query searches unbounded (r5 advice: range alone was checked, and the cap came out too small)
#pragma unroll
        for (int j = 1; j < K; ++j)
#pragma unroll
            for (int i = 0; i < j; ++i) ok = ok && c[i] != c[j];
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float d = (ux - px[j]) * (ux - px[j]) + (uy - py[j]) * (uy - py[j]);       // sqdist's arithmetic
            if (D == 3) d = d + (uz - pz[j]) * (uz - pz[j]);
            t = fmaxf(t, d);
        }
        level_cap[l * 64 + lane] = ok ? t : INFINITY;
    }
    if (slots || bounded) __syncthreads();
    if (bounded) {
        cap = 0.0f;
        for (int l = 0; l < po.levels; ++l)
            if (po.size[l] > w * chunk) cap = fmaxf(cap, level_cap[l * 64 + lane]);      // wave-uniform: levels holding this chunk
    }
    KNN_STAMP(0)
    // snapshot of the merged prefix of `covered` candidates into every level of that size (wave 0 only)
    auto emit = [&](int covered, bool merged) {
        for (int l = 0; l < po.levels; ++l) {
            if (po.size[l] != covered) continue;            // wave-uniform
            const uint64_t tied = __ballot(merged && (ev_min == dist[K - 1]) && q_raw < Nq);
            KNN_COUNT(4, __builtin_popcountll(tied))
            if (q_raw < Nq && !((tied >> lane) & 1ull)) {
                int64_t* o = po.out[l] + ((size_t)b * Nq + q_raw) * K;
#pragma unroll
                for (int j = 0; j < K; ++j) o[j] = (int64_t)idx[j];
            }
            // a merged list may differ from the in-order result in WHICH index it keeps at the k-th distance: those queries
            // are redone in order over this prefix (the running lists go on unchanged: they only feed larger prefixes)
            if (tied) redo_tied_queries<D, K>(tied, in_b, covered, ux, uy, uz, dist[K - 1], po.out[l] + ((size_t)b * Nq + blockIdx.x * 64) * K, lane);
        }
    };
    // Levels smaller than a chunk lie inside wave 0's range: its in-order list after the first size[l] candidates IS that
    // level's answer (no merge, no tie question), so wave 0 scans its chunk in pieces and takes a snapshot between them.
    // (Round 5: lets k = 32 take the one-launch path on an 8-chunk pyramid -- 4 waves of 2 chunks -- and fewer, larger chunks
    // be measured: profiles/r05_experiments.txt 12.)
    if (w == 0) {
        int lo = 0;
        for (int l = po.levels - 1; l >= 0; --l) {
            if (po.size[l] >= chunk) break;
            scan_range<D, K>(in_b, lo, po.size[l], ux, uy, uz, dist, idx, ev_min, qd, qi, nthreads, slots, NW, w, group, cap);
            emit(po.size[l], false);
            lo = po.size[l];
        }
        scan_range<D, K>(in_b, lo, chunk, ux, uy, uz, dist, idx, ev_min, qd, qi, nthreads, slots, NW, w, group, cap);
    } else {
        scan_range<D, K>(in_b, w * chunk, (w + 1) * chunk, ux, uy, uz, dist, idx, ev_min, qd, qi, nthreads, slots, NW, w, group, cap);
    }
    KNN_STAMP(1)

    __syncthreads();  // queues are dead; the merge region aliases them
    float* md = smem;                                          // [NW][K][64]
    int* mi = reinterpret_cast<int*>(smem) + NW * K * 64;      // [NW][K][64]
    float* mev = smem + 2 * NW * K * 64;                       // [NW][64]
    // binary tree when the wave count and every level size (in chunks) are powers of two -- the FPS pyramids are
    // (2048, 1024, 512, 256): the merged prefix after round r covers 2^(r+1) chunks, which is where the snapshots fall
    bool tree = (NW & (NW - 1)) == 0;
    for (int l = 0; l < po.levels; ++l) {
        const int sl = po.size[l] / chunk;
        tree = tree && (sl & (sl - 1)) == 0;
    }
#ifdef CAMLI_KNN_PROFILE
    auto flush = [&]() {
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
            for (int k2 = 0; k2 < 8; ++k2) camli_knn_prof[w][k2] = prof_[k2];
        if (w == 0 && lane == 0 && blockIdx.x < 64 && blockIdx.y < 64) {
            unsigned* o = camli_knn_wg_ticks[blockIdx.y][blockIdx.x];
            o[0] = (unsigned)prof_[1]; o[1] = (unsigned)prof_[2]; o[2] = (unsigned)prof_[3]; o[3] = (unsigned)prof_[4];
        }
    };
#endif
    if (tree) {
        if (w == 0) emit(chunk, false);
        KNN_STAMP(3)
        for (int step = 1; step < NW; step <<= 1) {
            if ((w & (2 * step - 1)) == step) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    md[(w * K + j) * 64 + lane] = dist[j];
                    mi[(w * K + j) * 64 + lane] = idx[j];
                }
                mev[w * 64 + lane] = ev_min;
            }
            __syncthreads();
            if ((w & (2 * step - 1)) == 0 && w + step < NW) {
                float od[K];
                int oi[K];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    od[j] = md[((w + step) * K + j) * 64 + lane];
                    oi[j] = mi[((w + step) * K + j) * 64 + lane];
                }
                ev_min = fminf(ev_min, mev[(w + step) * 64 + lane]);
                merge_topk<K>(dist, idx, od, oi, ev_min);
            }
            KNN_STAMP(2)
            if (w == 0) emit(2 * step * chunk, true);
            KNN_STAMP(3)
        }
#ifdef CAMLI_KNN_PROFILE
        flush();
#endif
        return;
    }
    if (w > 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            md[(w * K + j) * 64 + lane] = dist[j];
            mi[(w * K + j) * 64 + lane] = idx[j];
        }
        mev[w * 64 + lane] = ev_min;
    }
    __syncthreads();
    if (w > 0) return;
    for (int s = 0; s < NW; ++s) {
        if (s > 0) {
            ev_min = fminf(ev_min, mev[s * 64 + lane]);
            for (int j = 0; j < K; ++j) {
                float d = md[(s * K + j) * 64 + lane];
                int c = mi[(s * K + j) * 64 + lane];
                bool acc = !(d > dist[K - 1]);
                if (!__ballot(acc)) break;
                if (acc) list_insert<K>(dist, idx, d, c, ev_min);
            }
        }
        emit((s + 1) * chunk, s > 0);
    }
}

// Literal one-thread-per-query form for any k in 1..64 that has no specialisation above.
template <int D>
__global__ __launch_bounds__(64) void knn_generic_kernel(const float* __restrict__ input,
                                                         const float* __restrict__ query,
                                                         int64_t* __restrict__ out, int M, int Nq, int k) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= Nq) return;
    const float* __restrict__ in_b = input + (size_t)b * M * D;
    const float* qp = query + ((size_t)b * Nq + q) * D;
    const float ux = qp[0], uy = qp[1], uz = (D == 3) ? qp[2] : 0.0f;
    float nd[64];
    int ni[64];
    for (int i = 0; i < 64; ++i) {
        nd[i] = KNN_INIT;
        ni[i] = 0;
    }
    for (int c = 0; c < M; ++c) {
        float d = sqdist<D>(ux, uy, uz, in_b + (size_t)c * D);
        if (d > nd[k - 1]) continue;
        int j = c < k - 1 ? c : k - 1;
        while (j > 0 && nd[j - 1] > d) {
            nd[j] = nd[j - 1];
            ni[j] = ni[j - 1];
            --j;
        }
        nd[j] = d;
        ni[j] = c;
    }
    int64_t* o = out + ((size_t)b * Nq + q) * k;
    for (int i = 0; i < k; ++i) o[i] = (int64_t)ni[i];
}

// =====================================================================================================================
// Round 4: candidates ACROSS the lanes.
//
// The lane-per-query kernels above spend 7/8 of their instructions on selection (queue test, insertion network,
// partial-list merges) and need the candidate range split over waves to fill the chip when there are few queries.
// Here a wave works on ONE query at a time and its 64 lanes hold the candidates:
//
//   * the candidates of the batch element sit in LDS as three coordinate planes (staged once per workgroup, shared by
//     its four waves); a lane reads 4 consecutive candidates per 16-byte LDS access (slot 4g+u of lane l = candidate
//     4 (64 g + l) + u), the query coordinates are wave-uniform (scalar loads), so the distance arithmetic is 8 VALU
//     operations per 64 pairs (same unfused fp32 expression) next to 3/4 of an LDS read -- 24 KB of LDS traffic per
//     query at M = 2048, a few % of the LDS rate.  (Round 4 first kept them in 96 registers: the kernel then sat at
//     170-200 VGPRs, two waves per SIMD or spills with serialised scratch reloads, 40 us instead of 17.)
//   * phase 1 also keeps the minimum of each lane's J distances.  The k-th smallest of the 64 lane minima (one
//     64-lane bitonic sort of the keys on DPP / v_permlane*_swap compare-exchanges, distances ordered as their bit
//     patterns) is an upper bound T of the k-th neighbour distance: k different lanes hold a candidate within it;
//   * phase 2 ballots `d <= T` per slot; the survivors (k + a few: 18 on average for k = 16, whatever M is) are
//     compacted into a 128-entry LDS list with v_mbcnt prefix counts;
//   * each survivor's final position is its rank by (distance, index) among the survivors -- counted against the
//     list read back as LDS broadcasts, no second sort -- and the lane stores its index at out[rank].
//
// Exactness.  Every candidate within the k-th distance D survives.  Sequential insertion
// (k_nearest_neighbor_kernel.cu:52-95) keeps every candidate closer than D, ordered by (distance, index); of the
// candidates AT distance D it keeps s = k - #closer, and which ones follows from the arrival order in closed form: once
// the list holds k candidates within D it stays full, a later closer candidate pops the LAST tied entry and a later
// tied candidate overwrites it (slot k-1), so the result is the first s-1 tied candidates by index followed by either
// the last tied candidate of all (when no closer candidate arrives after it) or the s-th tied one.  Both are ranks in
// the survivor list, so ties cost a few scalar operations, not a rescan.  Only a survivor-list overflow (more than
// 128 / W survivors in one wave: duplicates-only clouds) and a real candidate at exactly the initial distance 1e9
// send a query to an in-order lane-per-query scan (scan_range above) after the wave's query loop, when the candidate
// registers are dead.  Unfilled slots (fewer than k candidates within 1e9) keep index 0 like the reference's list.
//
// M > 2048: the candidates pass through LDS in chunks of 2048 (knn_xlane_chunked_kernel below): the lane minima and the
// survivor list of a query live across the chunks, the bound tightens chunk by chunk.  (Round 4 first tried TEAMS of
// 2 / 4 / 8 waves holding 2048 candidates each in registers and walking the queries in step -- two barriers per query
// at two waves per SIMD: 216 us at (8192, 4096, k 16) where the lane-per-query kernel takes 157.)
namespace xl {

constexpr uint32_t INIT_BITS = 0x4e6e6b28u;   // 1e9f
constexpr int QPT_MAX = 64;                     // queries per team
constexpr int CAP = 128;                        // survivor list entries (two per lane)

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
// compare-exchange with the lane at a DPP-reachable position inside the 16-lane row; `lower` lanes keep the minimum
template <int CTRL>
__device__ __forceinline__ uint32_t cx_row(uint32_t x, bool lower) {
    const uint32_t p = dpp<CTRL>(x);
    const uint32_t lo = min(x, p), hi = max(x, p);
    return lower ? lo : hi;
}
// partner lane ^ 4: banks 0, 2 of a row read four lanes up, banks 1, 3 four lanes down
__device__ __forceinline__ uint32_t cx4(uint32_t x, bool lower) {
    uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0x5, false);
    p = (uint32_t)__builtin_amdgcn_update_dpp((int)p, (int)x, 0x114, 0xf, 0xA, false);
    const uint32_t lo = min(x, p), hi = max(x, p);
    return lower ? lo : hi;
}
struct LaneBits {
    bool b1, b2, b4, b8, b16, b32;
    __device__ __forceinline__ explicit LaneBits(int lane)
        : b1(!(lane & 1)), b2(!(lane & 2)), b4(!(lane & 4)), b8(!(lane & 8)), b16(!(lane & 16)), b32(!(lane & 32)) {}
};
// ascending sort of one 32-bit key per lane across the 64 lanes of the wave: bitonic network in the form whose every
// merge starts with a mirror (lane ^ (size - 1)) so that all comparators point the same way.  quad_perm / row_mirror /
// row_half_mirror / row_ror:8 / row_sh[lr]:4 inside the rows, v_permlane16_swap and v_permlane32_swap across them.
__device__ __forceinline__ uint32_t wave_sort_u32(uint32_t x, const LaneBits& lb) {
    constexpr int X1 = 0xB1, X2 = 0x4E, X3 = 0x1B, HMIRROR = 0x141, MIRROR = 0x140, ROR8 = 0x128;
    x = cx_row<X1>(x, lb.b1);
    x = cx_row<X3>(x, lb.b2); x = cx_row<X1>(x, lb.b1);
    x = cx_row<HMIRROR>(x, lb.b4); x = cx_row<X2>(x, lb.b2); x = cx_row<X1>(x, lb.b1);
    x = cx_row<MIRROR>(x, lb.b8); x = cx4(x, lb.b4); x = cx_row<X2>(x, lb.b2); x = cx_row<X1>(x, lb.b1);
    {   // lane ^ 31: mirror inside the row, then swap the rows of a pair
        const uint32_t y = dpp<MIRROR>(x);
        const auto r = __builtin_amdgcn_permlane16_swap(y, y, false, false);   // [y0 y0 y2 y2], [y1 y1 y3 y3]
        const uint32_t p = lb.b16 ? r[1] : r[0];
        x = lb.b16 ? min(x, p) : max(x, p);
    }
    x = cx_row<ROR8>(x, lb.b8); x = cx4(x, lb.b4); x = cx_row<X2>(x, lb.b2); x = cx_row<X1>(x, lb.b1);
    {   // lane ^ 63
        const uint32_t y = dpp<MIRROR>(x);
        const auto r = __builtin_amdgcn_permlane16_swap(y, y, false, false);
        const uint32_t y2 = lb.b16 ? r[1] : r[0];                                // x[lane ^ 31]
        const auto s = __builtin_amdgcn_permlane32_swap(y2, y2, false, false);   // [lo lo], [hi hi]
        const uint32_t p = lb.b32 ? s[1] : s[0];
        x = lb.b32 ? min(x, p) : max(x, p);
    }
    {   // lane ^ 16
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        const uint32_t lo = min(r[0], r[1]), hi = max(r[0], r[1]);
        x = lb.b16 ? lo : hi;
    }
    x = cx_row<ROR8>(x, lb.b8); x = cx4(x, lb.b4); x = cx_row<X2>(x, lb.b2); x = cx_row<X1>(x, lb.b1);
    return x;
}
// minimum over the wave, in every lane
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    x = min(x, dpp<0xB1>(x));
    x = min(x, dpp<0x4E>(x));
    x = min(x, dpp<0x141>(x));
    x = min(x, dpp<0x140>(x));
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    x = min(r[0], r[1]);
    const auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return min(s[0], s[1]);
}
// the k-th smallest of the 64 lane values (wave-uniform), never above the initial distance of the reference's list
__device__ __forceinline__ uint32_t kth_bound(uint32_t m, int k, const LaneBits& lb) {
    uint32_t t;
    if (k == 1) t = wave_min_u32(m);
    else t = wave_sort_u32(m, lb);
    t = (uint32_t)__builtin_amdgcn_readlane((int)t, k - 1);
    return t < INIT_BITS ? t : INIT_BITS;
}

template <int D>
__device__ __forceinline__ uint32_t dist_bits(float ux, float uy, float uz, float x, float y, float z) {
    float d = (ux - x) * (ux - x) + (uy - y) * (uy - y);
    if (D == 3) d = d + (uz - z) * (uz - z);
    return __float_as_uint(d);
}

// Survivor list: (CAP + 4 pad) entries of (index, distance bits); unused entries hold the sentinel ~0 (no real key
// reaches it: distances stay <= 1e9).
constexpr int LIST_DW = 2 * (CAP + 4);
static_assert(LIST_DW % 4 == 0, "16-byte aligned LDS regions");

// The exact in-order insertion of ONE query by a whole wave, for the two cases the closed forms do not cover (a
// survivor list overflow: clouds that are mostly duplicates; a real candidate at exactly the initial distance).  Lane
// j < k holds entry j of the reference's list; 64 candidates are tested per trip against the running k-th distance
// and the accepted ones are inserted one by one in index order with the reference's own rule
// (k_nearest_neighbor_kernel.cu:80-90: start at slot min(idx, k-1), move left past entries with dist > d).  Slow (a
// dozen instructions per accepted candidate) and rare; needs no LDS and ~10 registers.
// `bound`: an upper bound of the final k-th distance when the caller has one (KNN_INIT otherwise): candidates beyond it are
// inserted and pushed out again by the reference without a trace in the final list, so they are skipped.
template <int D>
__device__ __noinline__ void redo_query(const float* __restrict__ in_b, int M, float ux, float uy, float uz, int k,
                                        int64_t* __restrict__ o, int lane, float bound) {
    float ld = KNN_INIT;
    int li = 0;
    for (int c0 = 0; c0 < M; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < M;
        const float* p = in_b + (size_t)(in ? c : M - 1) * D;
        float d = (ux - p[0]) * (ux - p[0]) + (uy - p[1]) * (uy - p[1]);
        if (D == 3) d = d + (uz - p[2]) * (uz - p[2]);
        float kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ld), k - 1));
        uint64_t acc = __ballot(in && !(d > kth) && !(d > bound));
        while (acc) {
            const int src = (int)__builtin_ctzll(acc);
            acc &= acc - 1;
            const float dv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), src));
            kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ld), k - 1));
            if (dv > kth) continue;
            const int ci = c0 + src;
            const int j0 = ci < k - 1 ? ci : k - 1;
            const int pos = __builtin_popcountll(__ballot(lane < j0 && !(ld > dv)));     // entries left of j0 that stay
            const float up_d = __shfl_up(ld, 1, 64);
            const int up_i = __shfl_up(li, 1, 64);
            const bool shifted = lane > pos && lane <= j0;
            ld = shifted ? up_d : (lane == pos ? dv : ld);
            li = shifted ? up_i : (lane == pos ? ci : li);
        }
    }
    if (lane < k) o[lane] = (int64_t)li;
}

__device__ __forceinline__ int first_lane(uint64_t mask) { return (int)__builtin_ctzll(mask); }

// Rank + store one query from its survivor list and leave the list all-sentinel again.  Entry t of team wave w sits
// at slot t * W + w (lane l reads slots l and 64 + l); `nslots` = W * the largest per-wave count, `two` = some slot
// >= 64 is in use, `total` = the number of entries.  Ties at the k-th distance in closed form (header).  Returns true
// when a real candidate sits at exactly the reference's initial distance (nothing is stored then: the caller redoes the
// query in order).
// CHAIN (nested prefixes, knn_xlane_chain_kernel): the list starts with the k results of the next smaller prefix -- the
// reference's list after that prefix, from which its insertion simply continues -- followed by the new survivors
// (index >= new_from).  The stored results also go to `next` (slot j = output j) as the start of the next larger
// prefix's list; `fo` receives the k-th distance (the bound of that prefix) and the number of results.  Among the
// carried-over entries "a closer candidate arrives after the last tied one" only holds for NEW candidates: the carried
// ones arrive in list order, closer before tied, whatever their indices.
struct FinishOut {
    uint32_t dk;
    int kept;
};
template <bool CHAIN = false>
__device__ __forceinline__ bool finish_query(uint32_t* list, int nslots, int total, bool two, int k,
                                             int64_t* __restrict__ o, int lane, uint32_t new_from = 0,
                                             uint32_t* next = nullptr, FinishOut* fo = nullptr) {
    auto emit = [&](int slot, uint2 e) {
        o[slot] = (int64_t)e.x;
        if (CHAIN && next) *reinterpret_cast<uint2*>(next + 2 * slot) = e;
    };
    const uint2 e0 = *reinterpret_cast<const uint2*>(list + 2 * lane);                 // (index, distance bits)
    uint2 e1 = make_uint2(0xffffffffu, 0xffffffffu);
    if (two) e1 = *reinterpret_cast<const uint2*>(list + 2 * (64 + lane));
    const uint64_t key0 = ((uint64_t)e0.y << 32) | e0.x, key1 = ((uint64_t)e1.y << 32) | e1.x;
    const bool v0 = e0.y != 0xffffffffu, v1 = e1.y != 0xffffffffu;
    int rank0 = 0, rank1 = 0;
    // wave-uniform addresses: LDS broadcasts, four entries per trip (the pad keeps s + 3 in range; sixteen per trip --
    // fewer LDS round trips, more padding work -- measured slower: 24.0 vs 23.1 us at k 16, 12.2 vs 10.7 us at M = 256)
    for (int s = 0; s < nslots; s += 4) {
        const uint4 a = *reinterpret_cast<const uint4*>(list + 2 * s);
        const uint4 b = *reinterpret_cast<const uint4*>(list + 2 * s + 4);
        const uint64_t s0 = ((uint64_t)a.y << 32) | a.x, s1 = ((uint64_t)a.w << 32) | a.z;
        const uint64_t s2 = ((uint64_t)b.y << 32) | b.x, s3 = ((uint64_t)b.w << 32) | b.z;
        rank0 += (s0 < key0) + (s1 < key0) + (s2 < key0) + (s3 < key0);
        if (two) rank1 += (s0 < key1) + (s1 < key1) + (s2 < key1) + (s3 < key1);
    }
    // every lane has its entries in registers: reset them for the next query
    if (v0) *reinterpret_cast<uint2*>(list + 2 * lane) = make_uint2(0xffffffffu, 0xffffffffu);
    if (v1) *reinterpret_cast<uint2*>(list + 2 * (64 + lane)) = make_uint2(0xffffffffu, 0xffffffffu);

    if (__ballot((v0 && e0.y == INIT_BITS) || (v1 && e1.y == INIT_BITS)) != 0) return true;
    if (total < k) {            // fewer than k candidates within 1e9: the rest of the reference's list keeps index 0
        if (v0) emit(rank0, e0);
        if (v1) emit(rank1, e1);
        if (lane >= total && lane < k) o[lane] = 0;
        if (CHAIN && fo) fo->dk = INIT_BITS, fo->kept = total;
        return false;
    }
    uint32_t dk;                // the k-th distance = that of rank k - 1
    {
        const uint64_t m0 = __ballot(v0 && rank0 == k - 1);
        if (m0) dk = (uint32_t)__builtin_amdgcn_readlane((int)e0.y, first_lane(m0));
        else dk = (uint32_t)__builtin_amdgcn_readlane((int)e1.y, first_lane(__ballot(v1 && rank1 == k - 1)));
    }
    if (CHAIN && fo) fo->dk = dk, fo->kept = k;
    // the common case: nothing beyond rank k - 1 shares the k-th distance -> the ranks are the output slots
    if (__ballot((v0 && rank0 >= k && e0.y == dk) || (v1 && rank1 >= k && e1.y == dk)) == 0) {
        if (v0 && rank0 < k) emit(rank0, e0);
        if (v1 && rank1 < k) emit(rank1, e1);
        return false;
    }
    const bool less0 = v0 && e0.y < dk, less1 = v1 && e1.y < dk;
    const bool tie0 = v0 && e0.y == dk, tie1 = v1 && e1.y == dk;
    const int n_less = __builtin_popcountll(__ballot(less0)) + __builtin_popcountll(__ballot(less1));
    const int n_tie = __builtin_popcountll(__ballot(tie0)) + __builtin_popcountll(__ballot(tie1));
    int top_rank = k - 1;                         // rank of the entry that ends in slot k - 1
    if (n_less + n_tie > k) {                     // more tied candidates than slots
        const int last = n_less + n_tie - 1;      // rank of the tied candidate with the highest index
        uint32_t last_idx;
        const uint64_t m0 = __ballot(v0 && rank0 == last);
        if (m0) last_idx = (uint32_t)__builtin_amdgcn_readlane((int)e0.x, first_lane(m0));
        else last_idx = (uint32_t)__builtin_amdgcn_readlane((int)e1.x, first_lane(__ballot(v1 && rank1 == last)));
        const bool closer_after = __ballot((less0 && e0.x > last_idx && e0.x >= new_from) ||
                                           (less1 && e1.x > last_idx && e1.x >= new_from)) != 0;
        if (!closer_after) top_rank = last;
    }
    if (v0) {
        if (rank0 < k - 1) emit(rank0, e0);
        else if (rank0 == top_rank) emit(k - 1, e0);
    }
    if (v1) {
        if (rank1 < k - 1) emit(rank1, e1);
        else if (rank1 == top_rank) emit(k - 1, e1);
    }
    return false;
}

// phase 2 of one wave: entries with d <= T go to slots (cnt + prefix) * W + tw of `list`, at most R per wave (the
// count keeps running past R: the caller sees the overflow); returns the new count
template <int J, int J0, int J1, typename IndexOf>
__device__ __forceinline__ int collect(const uint32_t (&d)[J], uint32_t T, uint32_t* list, int W, int tw, int R,
                                       IndexOf&& index_of, int cnt = 0) {
    auto append = [&](uint64_t mask, bool hit, uint32_t dv, int j) {
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, (uint32_t)cnt));
        if (hit && pos < R) *reinterpret_cast<uint2*>(list + 2 * (pos * W + tw)) = make_uint2((uint32_t)index_of(j), dv);
        cnt += __builtin_popcountll(mask);
    };
    static_assert(J0 % 4 == 0 && J1 % 4 == 0, "slots come in groups of four");
#pragma unroll
    for (int j = J0; j < J1; j += 4) {
        const bool h0 = d[j] <= T, h1 = d[j + 1] <= T, h2 = d[j + 2] <= T, h3 = d[j + 3] <= T;
        const uint64_t m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
        if ((m0 | m1) | (m2 | m3)) {
            if (m0) append(m0, h0, d[j], j);
            if (m1) append(m1, h1, d[j + 1], j + 1);
            if (m2) append(m2, h2, d[j + 2], j + 2);
            if (m3) append(m3, h3, d[j + 3], j + 3);
        }
    }
    return cnt;
}

// ---- M > 2048: the candidates pass through LDS in chunks of 2048 -----------------------------------------------------
// A workgroup (4 waves x QB queries each) stages one chunk at a time; every wave runs phase 1 + bound + phase 2 of each
// of its QB queries on the chunk before the next one is staged (two barriers per CHUNK, none per query).  A query's lane
// minima accumulate over the chunks, so the bound after chunk c is the k-th smallest lane minimum of the first c + 1
// chunks -- it only tightens, and every bound is above the final k-th distance, so the survivor list (one per query,
// kept in LDS over the chunk loop) is a superset of what the single-chunk kernel would collect: ~18 + 9 + 6 + 5 entries
// for four chunks at k = 16.  Ranking, ties and overflow as everywhere else.
constexpr int CH_J = 32, CH_QB = 4;
constexpr int CH_CAND_DW = 3 * 64 * CH_J, CH_WAVE_DW = CH_QB * LIST_DW;

template <int D>
__global__ __launch_bounds__(256) void knn_xlane_chunked_kernel(const float* __restrict__ input,
                                                                const float* __restrict__ query,
                                                                int64_t* __restrict__ out, int M, int Nq, int k, int qpt) {
    constexpr int J = CH_J, QB = CH_QB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + w) * qpt;
    const int q1 = q0 + qpt < Nq ? q0 + qpt : Nq;
    const LaneBits lb(lane);
    float* cand = smem;
    uint32_t* lists = reinterpret_cast<uint32_t*>(smem + CH_CAND_DW + w * CH_WAVE_DW);

    const float* __restrict__ in_b = input + (size_t)b * M * D;
    const float* __restrict__ query_b = query + (size_t)b * Nq * D;
    int64_t* __restrict__ out_b = out + (size_t)b * Nq * k;
    for (int i = lane; i < QB * LIST_DW; i += 64) lists[i] = 0xffffffffu;

    const float4* X4 = reinterpret_cast<const float4*>(cand) + lane;
    const float4* Y4 = reinterpret_cast<const float4*>(cand + 64 * J) + lane;
    const float4* Z4 = reinterpret_cast<const float4*>(cand + 128 * J) + lane;
    const int cl4 = 4 * lane;
    const int nchunks = (M + 64 * J - 1) / (64 * J);

    for (int qb = 0; qb < qpt; qb += QB) {          // same trip count in every wave of the workgroup (barriers inside)
        float ux[QB], uy[QB], uz[QB];
        uint32_t m[QB];
        int cnt[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int q = q0 + qb + u < q1 ? q0 + qb + u : (q1 > 0 ? q1 - 1 : 0);
            const float* qp = query_b + (size_t)q * D;
            ux[u] = qp[0];
            uy[u] = qp[1];
            uz[u] = (D == 3) ? qp[2] : 0.0f;
            m[u] = 0xffffffffu;
            cnt[u] = 0;
        }
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();                        // every wave is done with the previous chunk
            const int c0 = ch * 64 * J;
            for (int c = threadIdx.x; c < 64 * J; c += 256) {
                const bool in = c0 + c < M;
                const float* p = in_b + (size_t)(in ? c0 + c : M - 1) * D;
                cand[c] = in ? p[0] : INFINITY;      // past the end: distance inf / NaN, never <= T
                cand[64 * J + c] = p[1];
                if (D == 3) cand[128 * J + c] = p[2];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                if (q0 + qb + u < q1) {             // wave-uniform
                    uint32_t d[J];
                    uint32_t mm = m[u];
#pragma unroll
                    for (int g = 0; g < J / 4; ++g) {
                        const float4 x = X4[g * 64], y = Y4[g * 64];
                        float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (D == 3) z = Z4[g * 64];
                        d[4 * g + 0] = dist_bits<D>(ux[u], uy[u], uz[u], x.x, y.x, z.x);
                        d[4 * g + 1] = dist_bits<D>(ux[u], uy[u], uz[u], x.y, y.y, z.y);
                        d[4 * g + 2] = dist_bits<D>(ux[u], uy[u], uz[u], x.z, y.z, z.z);
                        d[4 * g + 3] = dist_bits<D>(ux[u], uy[u], uz[u], x.w, y.w, z.w);
                        mm = min(min(mm, d[4 * g]), min(d[4 * g + 1], min(d[4 * g + 2], d[4 * g + 3])));
                    }
                    m[u] = mm;
                    const uint32_t T = kth_bound(mm, k, lb);
                    cnt[u] = collect<J, 0, J>(d, T, lists + u * LIST_DW, 1, 0, CAP,
                                        [&](int slot) { return c0 + cl4 + 256 * (slot >> 2) + (slot & 3); }, cnt[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            if (q0 + qb + u < q1) {
                const int n = cnt[u] < CAP ? cnt[u] : CAP;
                int64_t* o = out_b + (size_t)(q0 + qb + u) * k;
                const bool init_tie = finish_query(lists + u * LIST_DW, n, n, n > 64, k, o, lane);
                if (cnt[u] > CAP || init_tie) redo_query<D>(in_b, M, ux[u], uy[u], uz[u], k, o, lane);
            }
        }
    }
}

// ---- M <= 2048: one wave per query, candidates in LDS --------------------------------------------------------------
// LDS of a workgroup (dwords): candidate planes X | Y | Z of 64 J floats each | per wave: survivor lists.
template <int J>
constexpr int cand_dw() { return 3 * 64 * J; }

// stage the candidates of batch element `in_b`: plane p at cand + 64 J p; entries past M get an infinite x (distance
// inf / NaN: never <= T)
template <int D, int J>
__device__ __forceinline__ void stage_candidates(float* cand, const float* __restrict__ in_b, int M) {
    for (int c = threadIdx.x; c < 64 * J; c += 256) {
        const bool in = c < M;
        const float* p = in_b + (size_t)(in ? c : M - 1) * D;
        cand[c] = in ? p[0] : INFINITY;
        cand[64 * J + c] = p[1];
        if (D == 3) cand[128 * J + c] = p[2];
    }
}

// phase 1 of one query: the J distances of every lane (slot 4g+u of lane l = candidate 4 (64 g + l) + u) and the lane
// minimum over the first G_MIN slot groups
template <int D, int J, int G_MIN>
__device__ __forceinline__ uint32_t distances(const float4* X4, const float4* Y4, const float4* Z4, float ux, float uy,
                                              float uz, uint32_t (&d)[J]) {
    uint32_t m = 0xffffffffu;
#pragma unroll
    for (int g = 0; g < J / 4; ++g) {
        const float4 x = X4[g * 64], y = Y4[g * 64];
        float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (D == 3) z = Z4[g * 64];
        d[4 * g + 0] = dist_bits<D>(ux, uy, uz, x.x, y.x, z.x);
        d[4 * g + 1] = dist_bits<D>(ux, uy, uz, x.y, y.y, z.y);
        d[4 * g + 2] = dist_bits<D>(ux, uy, uz, x.z, y.z, z.z);
        d[4 * g + 3] = dist_bits<D>(ux, uy, uz, x.w, y.w, z.w);
        if (g < G_MIN) m = min(min(m, d[4 * g]), min(d[4 * g + 1], min(d[4 * g + 2], d[4 * g + 3])));
    }
    return m;
}

template <int D, int J>
__global__ __launch_bounds__(256) void knn_xlane_kernel(const float* __restrict__ input, const float* __restrict__ query,
                                                        int64_t* __restrict__ out, int M, int Nq, int k, int qpt) {
    static_assert(J % 4 == 0, "slot groups of four");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + w) * qpt;
    const int q1 = q0 + qpt < Nq ? q0 + qpt : Nq;
    const LaneBits lb(lane);
    float* cand = smem;
    uint32_t* list = reinterpret_cast<uint32_t*>(smem + cand_dw<J>() + w * LIST_DW);

    const float* __restrict__ in_b = input + (size_t)b * M * D;
    const float* __restrict__ query_b = query + (size_t)b * Nq * D;
    int64_t* __restrict__ out_b = out + (size_t)b * Nq * k;
    stage_candidates<D, J>(cand, in_b, M);
    for (int i = lane; i < LIST_DW; i += 64) list[i] = 0xffffffffu;
    __syncthreads();
    if (q0 >= Nq) return;

    const float4* X4 = reinterpret_cast<const float4*>(cand) + lane;
    const float4* Y4 = reinterpret_cast<const float4*>(cand + 64 * J) + lane;
    const float4* Z4 = reinterpret_cast<const float4*>(cand + 128 * J) + lane;
    const int cl4 = 4 * lane;
    float nx = query_b[(size_t)q0 * D], ny = query_b[(size_t)q0 * D + 1], nz = (D == 3) ? query_b[(size_t)q0 * D + 2] : 0.0f;
    for (int qi = q0; qi < q1; ++qi) {
        const float ux = nx, uy = ny, uz = nz;
        {   // next query's coordinates: requested now, waited for at the end of the iteration
            const int qn = qi + 1 < q1 ? qi + 1 : qi;
            const float* qp = query_b + (size_t)qn * D;
            nx = qp[0];
            ny = qp[1];
            nz = (D == 3) ? qp[2] : 0.0f;
        }
        uint32_t d[J];
        const uint32_t m = distances<D, J, J / 4>(X4, Y4, Z4, ux, uy, uz, d);
        const uint32_t T = kth_bound(m, k, lb);
        const int cnt = collect<J, 0, J>(d, T, list, 1, 0, CAP, [&](int slot) { return cl4 + 256 * (slot >> 2) + (slot & 3); });
        const int n = cnt < CAP ? cnt : CAP;
        int64_t* o = out_b + (size_t)qi * k;
        const bool init_tie = finish_query(list, n, n, n > 64, k, o, lane);
        if (cnt > CAP || init_tie) redo_query<D>(in_b, M, ux, uy, uz, k, o, lane);
    }
}

// ---- nested prefixes (camli_knn_prefixes): level l = the first M >> l candidates = the first (J / 4) >> l slot groups ----
// ONE pass of distance arithmetic serves all levels, and only the SMALLEST prefix pays for a bound of its own (the sort of
// its lane minima).  The reference's insertion over a larger prefix is the insertion over the smaller one continued with
// the remaining candidates, so level l starts from the k results of level l + 1, takes their k-th distance -- an exact
// upper bound of its own -- as T, and collects survivors from the INCREMENT slots only (~k of them: the increment holds as
// many candidates as the smaller prefix).  finish_query<CHAIN> ranks results + survivors together (ties: see there).
// Two lists per wave, used alternately by the levels.  (First form of this kernel, kept in the history: an own sorted
// bound, a full-prefix collection and a list per level -- 64 us per call against 60 us for the lane-per-query prefix kernel.)
template <int J, int L>
__global__ __launch_bounds__(256) void knn_xlane_chain_kernel(const float* __restrict__ input, const float* __restrict__ query,
                                                              KnnPrefixOut po, int M, int Nq, int k, int qpt) {
    constexpr int D = 3;
    static_assert(J % 4 == 0 && ((J / 4) >> (L - 1)) >= 1, "levels end on slot-group boundaries");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + w) * qpt;
    const int q1 = q0 + qpt < Nq ? q0 + qpt : Nq;
    const LaneBits lb(lane);
    float* cand = smem;
    uint32_t* lists = reinterpret_cast<uint32_t*>(smem + cand_dw<J>() + w * 2 * LIST_DW);

    const float* __restrict__ in_b = input + (size_t)b * M * D;
    const float* __restrict__ query_b = query + (size_t)b * Nq * D;
    stage_candidates<D, J>(cand, in_b, M);
    for (int i = lane; i < 2 * LIST_DW; i += 64) lists[i] = 0xffffffffu;
    __syncthreads();
    if (q0 >= Nq) return;

    const float4* X4 = reinterpret_cast<const float4*>(cand) + lane;
    const float4* Y4 = reinterpret_cast<const float4*>(cand + 64 * J) + lane;
    const float4* Z4 = reinterpret_cast<const float4*>(cand + 128 * J) + lane;
    const int cl4 = 4 * lane;
    auto index_of = [&](int slot) { return cl4 + 256 * (slot >> 2) + (slot & 3); };
    float nx = query_b[(size_t)q0 * D], ny = query_b[(size_t)q0 * D + 1], nz = query_b[(size_t)q0 * D + 2];
    for (int qi = q0; qi < q1; ++qi) {
        const float ux = nx, uy = ny, uz = nz;
        {
            const int qn = qi + 1 < q1 ? qi + 1 : qi;
            const float* qp = query_b + (size_t)qn * D;
            nx = qp[0];
            ny = qp[1];
            nz = qp[2];
        }
        constexpr int G_BASE = (J / 4) >> (L - 1);
        uint32_t d[J];
        const uint32_t m = distances<D, J, G_BASE>(X4, Y4, Z4, ux, uy, uz, d);
        uint32_t T = kth_bound(m, k, lb);
        int cnt = collect<J, 0, 4 * G_BASE>(d, T, lists, 1, 0, CAP, index_of);
        bool broken = false;
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            uint32_t* cur = lists + ((L - 1 - l) & 1) * LIST_DW;
            uint32_t* nxt = lists + ((L - l) & 1) * LIST_DW;
            int64_t* o = po.out[l] + ((size_t)b * Nq + qi) * k;
            if (!broken) {
                const int n = cnt < CAP ? cnt : CAP;
                FinishOut fo;
                const bool init_tie = finish_query<true>(cur, n, n, n > 64, k, o, lane, l == L - 1 ? 0u : (uint32_t)(M >> (l + 1)),
                                                         l > 0 ? nxt : nullptr, &fo);
                broken = init_tie || cnt > CAP;
                if (!broken && l > 0) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    T = fo.dk;
                    cnt = fo.kept;
                    // the increment of level l - 1: slot groups [(J/4) >> l, (J/4) >> (l - 1))
                    if (l == 1) cnt = collect<J, 4 * ((J / 4) >> 1), 4 * ((J / 4) >> 0)>(d, T, nxt, 1, 0, CAP, index_of, cnt);
                    if (L >= 3 && l == 2) cnt = collect<J, 4 * ((J / 4) >> 2), 4 * ((J / 4) >> 1)>(d, T, nxt, 1, 0, CAP, index_of, cnt);
                    if (L >= 4 && l == 3) cnt = collect<J, 4 * ((J / 4) >> 3), 4 * ((J / 4) >> 2)>(d, T, nxt, 1, 0, CAP, index_of, cnt);
                }
            }
            if (broken) redo_query<D>(in_b, M >> l, ux, uy, uz, k, o, lane);      // this level and every larger one, in order
        }
        if (broken)       // rare: whatever the chain left in the two lists must not reach the next query
            for (int i = lane; i < 2 * LIST_DW; i += 64) lists[i] = 0xffffffffu;
    }
}

static int mode() {   // CAMLI_KNN=lane forces the lane-per-query kernels, =xlane the cross-lane ones wherever they apply
    const char* e = getenv("CAMLI_KNN");      // read per call: tests and A/B tools flip it inside one process
    if (!e) return 0;
    return e[0] == 'l' ? 1 : (e[0] == 'x' ? 2 : 0);
}
static bool k_supported(int k) { return k >= 1 && k <= 32; }   // beyond 32 the lane-minimum bound admits too many survivors

static long long target_waves() {     // waves a launch is cut into (two rounds of 1024 SIMDs x 4 by default: measured)
    const char* e = getenv("CAMLI_KNN_XL_WAVES");
    const long long v = e ? atoll(e) : 8192LL;
    return v > 0 ? v : 8192LL;
}

template <int D>
int launch_chunked(const float* input, const float* query, int64_t* out, int B, int M, int Nq, int k, hipStream_t stream) {
    const long long waves = target_waves() * 3 / 4;                     // 42 KB of LDS per workgroup: three per CU
    const long long total = (long long)B * Nq;
    int qpt = (int)((total + waves - 1) / waves);
    qpt = (qpt + CH_QB - 1) / CH_QB * CH_QB;                            // whole batches of CH_QB queries
    if (qpt < CH_QB) qpt = CH_QB;
    if (qpt > QPT_MAX) qpt = QPT_MAX;
    dim3 grid(camli_divup(Nq, qpt * 4), B);
    const size_t lds = (size_t)(CH_CAND_DW + 4 * CH_WAVE_DW) * 4;
    hipLaunchKernelGGL((knn_xlane_chunked_kernel<D>), grid, dim3(256), lds, stream, input, query, out, M, Nq, k, qpt);
    return camli_check_launch("camli_knn(xlane chunked)");
}

static int queries_per_wave(int B, int Nq) {
    const long long occ = target_waves();
    const long long total = (long long)B * Nq;
    int qpt = (int)((total + occ - 1) / occ);
    if (qpt < 2) qpt = 2;
    return qpt > QPT_MAX ? QPT_MAX : qpt;
}

template <int D, int J>
int launch(const float* input, const float* query, int64_t* out, int B, int M, int Nq, int k, hipStream_t stream) {
    const int qpt = queries_per_wave(B, Nq);
    dim3 grid(camli_divup(Nq, qpt * 4), B);
    const size_t lds = (size_t)(cand_dw<J>() + 4 * LIST_DW) * 4;
    hipLaunchKernelGGL((knn_xlane_kernel<D, J>), grid, dim3(256), lds, stream, input, query, out, M, Nq, k, qpt);
    return camli_check_launch("camli_knn(xlane)");
}

template <int J, int L>
int launch_chain(const float* input, const float* query, const KnnPrefixOut& po, int B, int M, int Nq, int k, hipStream_t stream) {
    const int qpt = queries_per_wave(B, Nq);
    dim3 grid(camli_divup(Nq, qpt * 4), B);
    const size_t lds = (size_t)(cand_dw<J>() + 4 * 2 * LIST_DW) * 4;
    hipLaunchKernelGGL((knn_xlane_chain_kernel<J, L>), grid, dim3(256), lds, stream, input, query, po, M, Nq, k, qpt);
    return camli_check_launch("camli_knn_prefixes(xlane chain)");
}

// returns 1 when the shape is not served here (the caller falls back to the lane-per-query kernels).  Default choice
// (CAMLI_KNN unset) from the A/B table of tools/ab_knn.py on the MI355X (profiles/r04_ab_knn.json): the cross-lane
// kernels win wherever the per-query selection overhead of the lane-per-query form dominates -- k >= 2 up to 6144
// candidates (2.4x at (2048, 2048, k 16), 3.6x at k 32, 1.5x at (4096, 2048)); from 8192 candidates on (an accepted
// candidate is rare there: 11 VALU operations per pair) and for k = 1 (no selection at all) the lane-per-query scan stays.
template <int D>
int dispatch(const float* input, const float* query, int64_t* out, int B, int M, int Nq, int k, hipStream_t stream, int* rc) {
    if (!k_supported(k)) return 1;
    if (mode() == 0 && (k == 1 || M > 6144)) return 1;
    if (M <= 256) *rc = launch<D, 4>(input, query, out, B, M, Nq, k, stream);
    else if (M <= 512) *rc = launch<D, 8>(input, query, out, B, M, Nq, k, stream);
    else if (M <= 1024) *rc = launch<D, 16>(input, query, out, B, M, Nq, k, stream);
    else if (M <= 2048) *rc = launch<D, 32>(input, query, out, B, M, Nq, k, stream);
    else *rc = launch_chunked<D>(input, query, out, B, M, Nq, k, stream);
    return 0;
}

// 1 = shape not served here
int dispatch_prefix(const float* input, const float* query, const KnnPrefixOut& po, int B, int M, int Nq, int D, int k,
                    hipStream_t stream, int* rc) {
    if (D != 3 || !k_supported(k) || po.levels < 2) return 1;
    if (mode() == 0) return 1;      // measured (profiles/r04_ab_knn.json): 67 us against 60 us for the lane-per-query prefix kernel
    const int L = po.levels;
    for (int l = 0; l < L; ++l)
        if (po.size[l] != (M >> l) || (po.size[l] & 255)) return 1;       // levels end on slot-group boundaries
    const int J = M / 64;
    if (J * 64 != M) return 1;
#define CAMLI_XL_PREFIX(JJ, LL) \
    if (J == JJ && L == LL) { *rc = launch_chain<JJ, LL>(input, query, po, B, M, Nq, k, stream); return 0; }
    CAMLI_XL_PREFIX(32, 4) CAMLI_XL_PREFIX(32, 3) CAMLI_XL_PREFIX(32, 2)
    CAMLI_XL_PREFIX(16, 3) CAMLI_XL_PREFIX(16, 2) CAMLI_XL_PREFIX(8, 2)
#undef CAMLI_XL_PREFIX
    return 1;
}

}  // namespace xl

// CAMLI_KNN_SHARE=0 switches the published bounds off (A/B runs)
static int knn_share() {
    static const int v = [] { const char* e = getenv("CAMLI_KNN_SHARE"); return e ? atoi(e) : 1; }();
    return v;
}

template <int D, int K>
int launch_knn(const float* input, const float* query, int64_t* out, int B, int M, int Nq, hipStream_t stream) {
    const int qblocks = camli_divup(Nq, 64);
    // split the candidate range until the launch carries >= ~2 waves per SIMD (1024 SIMDs)
    int nw = 1;
    const long long base_waves = (long long)qblocks * B;
    const int lds_cap_nw = K >= 32 ? 4 : (K >= 16 ? 8 : 16);   // merge region <= 64 KiB
    static const long long target_waves = [] { const char* e = getenv("CAMLI_KNN_TARGET_WAVES"); return e ? atoll(e) : 4096LL; }();
    while (nw < lds_cap_nw && base_waves * nw < target_waves && M / (nw * 2) >= 128) nw *= 2;
    size_t q_bytes = (K >= 8) ? (size_t)2 * QBUF * 64 * nw * 4 : 0;
    if (K >= 8 && nw > 1) q_bytes += (size_t)KNN_SLOT_ROWS * nw * 64 * 4;      // published bounds
    size_t m_bytes = (nw > 1) ? ((size_t)2 * nw * K * 64 + (size_t)nw * 64) * 4 : 0;
    size_t lds = q_bytes > m_bytes ? q_bytes : m_bytes;
    dim3 grid(qblocks, B);
    hipLaunchKernelGGL((knn_kernel<D, K>), grid, dim3(64 * nw), lds, stream, input, query, out, M, Nq, knn_share());
    return camli_check_launch("camli_knn");
}

template <int D>
int dispatch_knn(const float* input, const float* query, int64_t* out, int B, int M, int Nq, int k,
                 hipStream_t stream) {
    switch (k) {
        case 1: return launch_knn<D, 1>(input, query, out, B, M, Nq, stream);
        case 3: return launch_knn<D, 3>(input, query, out, B, M, Nq, stream);
        case 4: return launch_knn<D, 4>(input, query, out, B, M, Nq, stream);
        case 8: return launch_knn<D, 8>(input, query, out, B, M, Nq, stream);
        case 16: return launch_knn<D, 16>(input, query, out, B, M, Nq, stream);
        case 32: return launch_knn<D, 32>(input, query, out, B, M, Nq, stream);
        default: {
            dim3 grid(camli_divup(Nq, 64), B);
            hipLaunchKernelGGL((knn_generic_kernel<D>), grid, dim3(64), 0, stream, input, query, out, M, Nq, k);
            return camli_check_launch("camli_knn(generic)");
        }
    }
}

}  // namespace

extern "C" int camli_knn(const float* input, const float* query, int64_t* out_idx, int B, int M, int Nq, int D,
                         int k, void* stream) {
    if (B == 0 || Nq == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!input || !query || !out_idx) {
        camli_set_error("camli_knn: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || M < 1 || Nq < 0 || (D != 2 && D != 3) || k < 1 || k > 64) {
        camli_set_error("camli_knn: bad shape B=%d M=%d Nq=%d D=%d k=%d (need D in {2,3}, 1<=k<=64, M>=1)", B, M, Nq,
                        D, k);
        return CAMLI_EINVAL;
    }
    if (B == 0 || Nq == 0) return CAMLI_OK;
    if (B > 65535) {
        camli_set_error("camli_knn: batch %d exceeds grid.y limit", B);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (xl::mode() != 1) {
        int rc = CAMLI_OK;
        const int served = D == 2 ? xl::dispatch<2>(input, query, out_idx, B, M, Nq, k, s, &rc)
                                  : xl::dispatch<3>(input, query, out_idx, B, M, Nq, k, s, &rc);
        if (served == 0) return rc;
    }
    return D == 2 ? dispatch_knn<2>(input, query, out_idx, B, M, Nq, k, s)
                  : dispatch_knn<3>(input, query, out_idx, B, M, Nq, k, s);
}

// Nested prefixes: out_levels[l] [B,Nq,k] = the k nearest among the FIRST sizes[l] inputs (sizes strictly descending,
// sizes[0] = M).  One launch when every size is a multiple of the smallest one and M / smallest <= 8 (k = 16; 4 for
// k = 32); otherwise one camli_knn per level.
// `prior_levels` (may be null; then this is camli_knn_prefixes): out_levels of an EARLIER call with the same sizes and k on the
// same clouds moved a little (a GRU iteration later) -- an upper bound of every level's k-th distance comes out of its K
// candidates, and the scan queues nothing beyond it.  The results do not depend on the prior (any K different in-range
// candidates per query and level bound the k-th distance); a prior from different clouds only costs the speed-up.
extern "C" int camli_knn_prefixes_prior(const float* input, const float* query, int64_t* const* out_levels,
                                        const int64_t* const* prior_levels, const int* sizes, int L, int B, int M, int Nq, int D,
                                        int k, void* stream) {
    if (B == 0 || Nq == 0) return CAMLI_OK;
    if (!input || !query || !out_levels || !sizes) { camli_set_error("camli_knn_prefixes: null pointer"); return CAMLI_EINVAL; }
    if (L < 1 || L > 4 || B < 0 || M < 1 || (D != 2 && D != 3) || k < 1 || k > 64 || sizes[0] != M || B > 65535) {
        camli_set_error("camli_knn_prefixes: bad arguments L=%d B=%d M=%d D=%d k=%d (need 1<=L<=4, sizes[0]=M)", L, B, M, D, k);
        return CAMLI_EINVAL;
    }
    for (int l = 0; l < L; ++l)
        if (!out_levels[l] || sizes[l] < 1 || (l > 0 && sizes[l] >= sizes[l - 1])) {
            camli_set_error("camli_knn_prefixes: level sizes must be strictly descending and positive");
            return CAMLI_EINVAL;
        }
    if (xl::mode() != 1 && !prior_levels) {
        KnnPrefixOut xpo;
        xpo.levels = L;
        for (int l = 0; l < 4; ++l) {
            xpo.out[l] = l < L ? out_levels[l] : nullptr;
            xpo.prior[l] = nullptr;
            xpo.size[l] = l < L ? sizes[l] : 0;
        }
        int rc = CAMLI_OK;
        if (xl::dispatch_prefix(input, query, xpo, B, M, Nq, D, k, reinterpret_cast<hipStream_t>(stream), &rc) == 0) return rc;
    }
    // one wave per chunk of the candidate range; a chunk is a level size (levels below it are snapshots inside wave 0's scan):
    // the smallest level that leaves at most max_nw waves.  CAMLI_KNN_PREFIX_NW=4 asks for fewer, larger chunks (measured on the
    // 2048/1024/512/256 pyramid: 84 us against 58 us with 8 -- a wave's scan is a chain, two waves per SIMD overlap theirs)
    const char* nw_env = getenv("CAMLI_KNN_PREFIX_NW");
    const int want_nw = nw_env ? atoi(nw_env) : 8;
    const int max_nw = k >= 32 ? 4 : 8;
    int chunk = sizes[L - 1];
    for (int l = L - 1; l >= 0; --l)
        if (M % sizes[l] == 0 && M / sizes[l] <= max_nw && (M / sizes[l] >= want_nw || l == L - 1)) chunk = sizes[l];
    bool one_launch = (D == 3) && (k == 16 || k == 32) && chunk >= 64;
    for (int l = 0; l < L; ++l) one_launch = one_launch && (sizes[l] % chunk == 0 || sizes[l] < chunk);
    const int nw = M / chunk;
    one_launch = one_launch && nw >= 1 && nw <= max_nw && nw * chunk == M;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!one_launch) {
        // general shapes: one plain search per (level, sample) -- a prefix of sample b starts at b * M * D
        for (int l = 0; l < L; ++l)
            for (int b = 0; b < B; ++b) {
                const int rc = camli_knn(input + (size_t)b * M * D, query + (size_t)b * Nq * D,
                                         out_levels[l] + (size_t)b * Nq * k, 1, sizes[l], Nq, D, k, stream);
                if (rc != CAMLI_OK) return rc;
            }
        return CAMLI_OK;
    }
    KnnPrefixOut po;
    po.levels = L;
    for (int l = 0; l < 4; ++l) {
        po.out[l] = l < L ? out_levels[l] : nullptr;
        po.prior[l] = (prior_levels && l < L) ? prior_levels[l] : nullptr;
        po.size[l] = l < L ? sizes[l] : 0;
    }
    if (prior_levels)
        for (int l = 0; l < L; ++l)
            if (!prior_levels[l]) { camli_set_error("camli_knn_prefixes_prior: prior level %d is null", l); return CAMLI_EINVAL; }
    const size_t q_bytes = (size_t)2 * QBUF * 64 * nw * 4 + (size_t)KNN_SLOT_ROWS * nw * 64 * 4 + (size_t)4 * 64 * 4;   // queues, slots, level caps
    const size_t m_bytes = ((size_t)2 * nw * k * 64 + (size_t)nw * 64) * 4;
    const size_t lds = q_bytes > m_bytes ? q_bytes : m_bytes;
    dim3 grid(camli_divup(Nq, 64), B);
    if (k == 16)
        hipLaunchKernelGGL((knn_prefix_kernel<3, 16>), grid, dim3(64 * nw), lds, s, input, query, po, M, Nq, chunk, knn_share());
    else
        hipLaunchKernelGGL((knn_prefix_kernel<3, 32>), grid, dim3(64 * nw), lds, s, input, query, po, M, Nq, chunk, knn_share());
    return camli_check_launch("camli_knn_prefixes");
}

extern "C" int camli_knn_prefixes(const float* input, const float* query, int64_t* const* out_levels, const int* sizes,
                                  int L, int B, int M, int Nq, int D, int k, void* stream) {
    return camli_knn_prefixes_prior(input, query, out_levels, nullptr, sizes, L, B, M, Nq, D, k, stream);
}

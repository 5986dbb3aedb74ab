// Furthest-point sampling, gfx950.
//
// Replaces models/csrc/furthest_point_sampling/furthest_point_sampling_kernel.cu:34-85 of the
// reference (1024 threads, every step re-reads xyz + dist from global memory and tree-reduces
// through shared memory with 11 barriers).  Index-exact against oracle_fps
// (oracle/camli_oracle.c): start at index 0, dist init 1e10, unfused fp32 distance, arg-max with
// the LOWEST index among equal maxima.
//
// Tie rule, precisely: "lowest index" is the rule of the reference's importable path (wrapper.py:83-96, torch.max on
// the CPU) and of the goldens (fps_a / fps_b / fps_dup are generated with cpp_impl=False).  The reference's CUDA
// kernel resolves equal maxima differently: its shared-memory tree (furthest_point_sampling_kernel.cu:5-10,23-32)
// compares with `<=`, so among tied candidates the one held by the higher thread of each pair survives.  That order
// cannot be executed here (no nvcc), is an artefact of a 1024-thread reduction tree rather than a specification, and
// only matters on clouds with exactly repeated points (all-zero distance ties); it is deliberately NOT imitated.
// "Index-exact" therefore means: exact against the reference's Python path, and against its CUDA path on tie-free data.
//
// Design (CDNA4): the algorithm is a chain of n_samples dependent steps, so the only lever is
// the latency of one step, and on one CU that latency is instruction issue: (update + reductions)
// x resident waves.  One workgroup (one CU) owns a cloud; every thread keeps P = ceil(N/T) points
// AND their running distances in VGPRs for the whole kernel (no memory traffic inside the loop:
// the algorithmic traffic is one read of the cloud and one write of the indices).  T = 256 or 512
// threads (1-2 waves per SIMD): the per-wave reduction overhead is paid 4-8 times per step instead
// of 16, which measured 2x faster than a 1024-thread block.  Per step:
//   * register update + thread-local arg-max (packed fp32 math)
//   * wave all-max of the distance bits with DPP + permlane-swap (no LDS); the winning lane is
//     found with ONE ballot -- a unique maximum is the common case, ties take a min-index reduce
//   * one 8-byte LDS slot per wave, ONE barrier (slots double-buffered by step parity), then
//     every wave re-reduces the <= 8 slots itself
//   * the new centre's coordinates come from an LDS copy of the cloud (one broadcast ds_read)
//     when it fits, otherwise from the owner lane through the slot
// Distances are non-negative, so their fp32 bit patterns order like unsigned integers.
// Picks are collected in LDS and written once at the end.
#include "camli_common.h"
#include <stdlib.h>
#include <string.h>

namespace {


template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// all-lanes max / min over the wave: 4 DPP steps inside each row of 16, then the two gfx950
// permlane swaps (rows 1,3 <-> 0,2 and upper half <-> lower half)
__device__ __forceinline__ unsigned wave_allmax_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x141>(v));  // row_half_mirror
    v = max(v, dpp_u32<0x140>(v));  // row_mirror
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max((unsigned)r16[0], (unsigned)r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max((unsigned)r32[0], (unsigned)r32[1]);
}
__device__ __forceinline__ unsigned wave_allmin_u32(unsigned v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min((unsigned)r16[0], (unsigned)r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return min((unsigned)r32[0], (unsigned)r32[1]);
}

struct __attribute__((aligned(16))) FpsSlot {
    unsigned key;  // fp32 bits of the wave's max distance
    unsigned idx;  // lowest point index attaining it
    float x, y, z; // its coordinates (only used when the cloud does not fit in LDS)
    unsigned pad[3];
};

constexpr unsigned FPS_NOIDX = 0xffffffffu;
constexpr size_t FPS_LDS_BUDGET = 150 * 1024;

// (max key, lowest index among the lanes holding it) over the wave, result in every lane.
__device__ __forceinline__ void wave_argmax(unsigned key, unsigned idx, unsigned& out_key, unsigned& out_idx) {
    out_key = wave_allmax_u32(key);
    const unsigned long long hit = __ballot(key == out_key);
    if (__builtin_popcountll(hit) == 1) {      // wave-uniform branch; unique maximum: no second reduction
        out_idx = (unsigned)__builtin_amdgcn_readlane((int)idx, (int)__builtin_ctzll(hit));
    } else {
        out_idx = wave_allmin_u32(key == out_key ? idx : FPS_NOIDX);
    }
}

// point i = j*T + tid (j = register slot).  Padding slots (i >= N) hold distance -1 forever and
// report key 0 / index FPS_NOIDX, so they lose against any real point.
// dynamic LDS: picks[n_samples] ints, then (LDS_COORDS) the cloud as 3 float planes of N.
template <int P, int T, bool LDS_COORDS>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz_all, int64_t* __restrict__ out_all,
                                                 int N, int n_samples) {
    constexpr int NW = T / 64;
    __shared__ FpsSlot slots[2][NW];
    extern __shared__ __attribute__((aligned(16))) int dyn[];
    int* picks = dyn;
    float* tab = reinterpret_cast<float*>(dyn + ((n_samples + 3) & ~3));   // x[N], y[N], z[N]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
    int64_t* __restrict__ out = out_all + (size_t)blockIdx.x * n_samples;

    float px[P], py[P], pz[P], dist[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int i = j * T + tid;
        const bool ok = i < N;
        const int ic = ok ? i : 0;
        px[j] = xyz[ic * 3 + 0];
        py[j] = xyz[ic * 3 + 1];
        pz[j] = xyz[ic * 3 + 2];
        dist[j] = ok ? 1e10f : -1.0f;
        if (LDS_COORDS && ok) {
            tab[i] = px[j];
            tab[N + i] = py[j];
            tab[2 * N + i] = pz[j];
        }
    }
    int cur = 0;
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    if (LDS_COORDS) __syncthreads();

    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) picks[s] = cur;
        if (s == n_samples - 1) break;

        // ---- register update + thread-local arg-max (lowest slot wins ties: strict >) ----
        float best = -1.0f;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            const float d = dx * dx + dy * dy + dz * dz;
            const float nd = fminf(dist[j], d);
            dist[j] = nd;
            const bool gt = nd > best;
            best = gt ? nd : best;
            bj = gt ? j : bj;
        }
        const bool real = best >= 0.0f;
        const unsigned key = real ? __float_as_uint(best) : 0u;
        const unsigned bi = real ? (unsigned)(bj * T + tid) : FPS_NOIDX;

        // ---- wave arg-max, one slot per wave ----
        unsigned wkey, widx;
        wave_argmax(key, bi, wkey, widx);
        FpsSlot& mine = slots[s & 1][w];
        if (LDS_COORDS) {
            if (lane == 0) {
                mine.key = wkey;
                mine.idx = widx;
            }
        } else if (bi == widx && widx != FPS_NOIDX) {
            float ox = px[0], oy = py[0], oz = pz[0];   // the owner lane publishes the coordinates
#pragma unroll
            for (int j = 1; j < P; ++j) {
                const bool m = bj == j;
                ox = m ? px[j] : ox;
                oy = m ? py[j] : oy;
                oz = m ? pz[j] : oz;
            }
            mine.key = wkey;
            mine.idx = widx;
            mine.x = ox;
            mine.y = oy;
            mine.z = oz;
        } else if (lane == 0 && widx == FPS_NOIDX) {
            mine.key = 0u;
            mine.idx = FPS_NOIDX;
        }
        __syncthreads();

        // ---- every wave reduces the NW slots on its own (no second barrier) ----
        const FpsSlot* sp = &slots[s & 1][lane & (NW - 1)];
        const unsigned k2 = sp->key, i2 = sp->idx;
        unsigned gkey, gidx;
        wave_argmax(k2, i2, gkey, gidx);
        cur = (int)gidx;
        if (LDS_COORDS) {
            cx = tab[cur];
            cy = tab[N + cur];
            cz = tab[2 * N + cur];
        } else {
            const unsigned long long hit = __ballot(k2 == gkey && i2 == gidx);
            const FpsSlot* win = &slots[s & 1][__builtin_ctzll(hit) & (NW - 1)];
            cx = win->x;
            cy = win->y;
            cz = win->z;
        }
    }
    __syncthreads();
    for (int s = tid; s < n_samples; s += T) out[s] = (int64_t)picks[s];
}


template <int P, int T>
int launch_fps(const float* xyz, int64_t* out, int B, int N, int n_samples, hipStream_t stream) {
    const size_t pick_bytes = (size_t)((n_samples + 3) & ~3) * sizeof(int);
    const size_t tab_bytes = (size_t)3 * N * sizeof(float);
    if (pick_bytes + tab_bytes <= FPS_LDS_BUDGET) {
        hipLaunchKernelGGL((fps_kernel<P, T, true>), dim3(B), dim3(T), pick_bytes + tab_bytes, stream, xyz, out, N,
                           n_samples);
    } else if (pick_bytes <= FPS_LDS_BUDGET) {
        hipLaunchKernelGGL((fps_kernel<P, T, false>), dim3(B), dim3(T), pick_bytes, stream, xyz, out, N, n_samples);
    } else {
        camli_set_error("camli_fps: n_samples=%d does not fit the LDS pick buffer", n_samples);
        return CAMLI_ENOTSUP;
    }
    return camli_check_launch("camli_fps");
}


// =====================================================================================================================
// Bucket-pruned form (round 3) -- same picks, a fraction of the arithmetic.
//
// dist[p] <- min(dist[p], d(p, c)) leaves every point untouched whose distance to the new centre c is at least its
// current value.  The cloud is therefore sorted along a space-filling curve once (bitonic sort of (code, index) in LDS;
// round 5: a HILBERT curve -- consecutive cells of a Morton curve jump across the cloud at every octant boundary, the 64
// points on either side of a jump share a bucket whose box then spans the cloud and is updated in up to 40 % of all steps;
// on a Hilbert curve consecutive cells are always adjacent: 8.8 -> 5.6 bucket updates per step, 0.89 -> 0.77 us), a
// BUCKET is the 64 consecutive sorted points one wave holds in one register slot, and each bucket keeps its bounding
// box and the maximum of its running distances.  A step first evaluates, one bucket per lane,
//      lb = ((ex*ex + ey*ey) + ez*ez),  e = max(lo - c, c - hi, 0) per axis        (same fp32 operation order as d)
// and skips every bucket with lb >= bucket max.  The skip is EXACT, not approximate: fp32 subtraction, multiplication
// and addition are monotone, |p - c| >= e holds per axis for every p inside the box, hence d(p, c) >= lb as computed,
// so fminf would have returned the old value.  After a few hundred picks a new centre touches a handful of the
// (128 .. 256) buckets of a cloud, all held by one or two waves; the rest of the workgroup only runs the 15-instruction
// box test.  The wave's arg-max comes from the bucket maxima (one row-local DPP reduction) and looks inside a bucket
// only when it holds the maximum.  Tie rule unchanged: lowest ORIGINAL index among equal maxima (the permutation
// carries the original index per point), so the picks are bit-identical to fps_kernel and to the oracle.
// The winner's coordinates travel through the wave slots (no LDS copy of the cloud): one LDS round trip per step.
// =====================================================================================================================
template <bool MAX>
__device__ __forceinline__ float wave_all_f32(float v) {
#define CAMLI_F32_STEP(CTRL)                                                         \
    {                                                                                \
        const float o = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));          \
        v = MAX ? fmaxf(v, o) : fminf(v, o);                                         \
    }
    CAMLI_F32_STEP(0xB1)
    CAMLI_F32_STEP(0x4E)
    CAMLI_F32_STEP(0x141)
    CAMLI_F32_STEP(0x140)
#undef CAMLI_F32_STEP
    auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = MAX ? fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1])) : fminf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return MAX ? fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1])) : fminf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}

// max / min over groups of G lanes (G = 16: a DPP row; G = 32: two rows) when every group holds the same G values
template <int G>
__device__ __forceinline__ unsigned group_allmax_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x141>(v));
    v = max(v, dpp_u32<0x140>(v));
    if (G == 32) {
        auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = max((unsigned)r16[0], (unsigned)r16[1]);
    }
    return v;
}
template <int G>
__device__ __forceinline__ unsigned group_allmin_u32(unsigned v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    if (G == 32) {
        auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = min((unsigned)r16[0], (unsigned)r16[1]);
    }
    return v;
}

__device__ __forceinline__ unsigned morton_spread10(unsigned v) {      // 10 bits -> every third bit (bit interleave)
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// -DCAMLI_FPS_PROFILE (tools/microbench/fps_mb.hip only): shader-clock stamps around the phases of a step, summed per wave
// of workgroup 0 into camli_fps_prof[wave][0..4] = box test / bucket updates / wave arg-max / slot + barrier / reduce of the
// slots, [5] = buckets updated, [6] = steps with at least one bucket, [7] = steps.
#ifdef CAMLI_FPS_PROFILE
__device__ unsigned long long camli_fps_prof[16][8];
__device__ unsigned camli_fps_touch[16][32];       // [wave][slot]: steps in which the bucket was updated
#define FPS_STAMP(k)                                                          \
    {                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
        prof_[k] += now_ - t_;                                                \
        t_ = now_;                                                            \
    }
#define FPS_COUNT(k, n) prof_[k] += (unsigned long long)(n);
#else
#define FPS_STAMP(k)
#define FPS_COUNT(k, n)
#endif

// Hilbert curve index of a 10-bit lattice point, in place, "transposed" form (bit k of the index triple = bit k of
// x[0], x[1], x[2], most significant first): Skilling's axes-to-transpose (AIP Conf. Proc. 707, 2004).
__device__ __forceinline__ void hilbert_transpose10(unsigned (&x)[3]) {
    for (unsigned q = 512u; q > 1u; q >>= 1) {
        const unsigned p = q - 1u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (x[a] & q) {
                x[0] ^= p;
            } else {
                const unsigned t = (x[0] ^ x[a]) & p;
                x[0] ^= t;
                x[a] ^= t;
            }
        }
    }
    x[1] ^= x[0];
    x[2] ^= x[1];
    unsigned t = 0;
    for (unsigned q = 512u; q > 1u; q >>= 1)
        if (x[2] & q) t ^= q - 1u;
    x[0] ^= t;
    x[1] ^= t;
    x[2] ^= t;
}

// P register slots per thread (16 or 32), T threads; sorted position of (wave w, slot j, lane l) = (j*NW + w)*64 + l:
// the handful of buckets around a new centre are neighbours on the curve, so round-robin puts them into DIFFERENT waves
// (with wave-major order one wave carried 5 of the 7 touched buckets of a step -- the step's critical path).
// dynamic LDS: max(NP2 * 8 bytes for the sort, n_samples * 4 bytes for the picks); NP2 = next power of two >= N.
template <int P, int T>
__global__ __launch_bounds__(T) void fps_pruned_kernel(const float* __restrict__ xyz_all, int64_t* __restrict__ out_all,
                                                        int N, int n_samples, int NP2) {
    constexpr int NW = T / 64;
    static_assert(NW <= 16 && (P == 8 || P == 16 || P == 32), "slot / wave reductions are group-local");
    __shared__ FpsSlot slots[2][NW];
    __shared__ float box[NW][6];
    extern __shared__ __attribute__((aligned(16))) int dyn[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(dyn);
    int* picks = dyn;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
    int64_t* __restrict__ out = out_all + (size_t)blockIdx.x * n_samples;

    // ---- cloud bounding box ----
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += T) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[i * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_all_f32<false>(lo[a]);
        hi[a] = wave_all_f32<true>(hi[a]);
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            box[w][a] = lo[a];
            box[w][3 + a] = hi[a];
        }
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int v = 0; v < NW; ++v) {
            lo[a] = fminf(lo[a], box[v][a]);
            hi[a] = fmaxf(hi[a], box[v][3 + a]);
        }
        const float ext = hi[a] - lo[a];
        scale[a] = ext > 0.0f ? 1023.0f / ext : 0.0f;       // any monotone quantisation does: only locality matters
    }

    // ---- (Hilbert index, point index) keys, bitonic sort in LDS ----
    for (int i = tid; i < NP2; i += T) {
        unsigned long long key = ~0ull;
        if (i < N) {
            unsigned qv[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float q = (xyz[i * 3 + a] - lo[a]) * scale[a];
                qv[a] = (unsigned)fminf(fmaxf(q, 0.0f), 1023.0f);
            }
            hilbert_transpose10(qv);
            const unsigned code = morton_spread10(qv[0]) << 2 | morton_spread10(qv[1]) << 1 | morton_spread10(qv[2]);
            key = ((unsigned long long)code << 32) | (unsigned)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= NP2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (NP2 >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const unsigned long long a = keys[i], b = keys[i + j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[i + j] = a;
                }
            }
            __syncthreads();
        }
    }

    // ---- this thread's points, in sorted order; bucket boxes and maxima (lane l holds bucket l & (P-1)) ----
    float px[P], py[P], pz[P], dist[P];
    unsigned orig[P];
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float bmax = -1.0f;
    const int myslot = lane & (P - 1);
#pragma unroll
    for (int j = 0; j < P; ++j) {
        // consecutive buckets go to different waves (see header), and the wave index is rotated by half the slot number:
        // the buckets around the top-level splits of the curve sit at multiples of 16 buckets (1024 points of an 8192-point
        // cloud) and are the ones updated most often -- unrotated they all landed in wave 0, which (Morton order) updated
        // 2.5 buckets per step where the others updated 0.9 and held every step back (profiles/r05_experiments.txt 10)
        const int pos = (j * NW + ((w - (j >> 1)) & (NW - 1))) * 64 + lane;
        const bool ok = pos < N;
        const unsigned o = ok ? (unsigned)(keys[pos] & 0xffffffffu) : 0u;
        orig[j] = ok ? o : FPS_NOIDX;
        px[j] = xyz[o * 3 + 0];
        py[j] = xyz[o * 3 + 1];
        pz[j] = xyz[o * 3 + 2];
        dist[j] = ok ? 1e10f : -1.0f;
        const float l0 = wave_all_f32<false>(ok ? px[j] : INFINITY), h0 = wave_all_f32<true>(ok ? px[j] : -INFINITY);
        const float l1 = wave_all_f32<false>(ok ? py[j] : INFINITY), h1 = wave_all_f32<true>(ok ? py[j] : -INFINITY);
        const float l2 = wave_all_f32<false>(ok ? pz[j] : INFINITY), h2 = wave_all_f32<true>(ok ? pz[j] : -INFINITY);
        const bool any = __ballot(ok) != 0ull;
        if (myslot == j) {
            blo[0] = l0; blo[1] = l1; blo[2] = l2;
            bhi[0] = h0; bhi[1] = h1; bhi[2] = h2;
            bmax = any ? 1e10f : -1.0f;
        }
    }
    __syncthreads();      // the sort buffer becomes the pick list

    int cur = 0;
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    // per bucket (in its lane): the lowest original index attaining bmax and that point's coordinates -- refreshed when
    // the bucket is updated, so the wave arg-max never looks into the register slots again
    unsigned bidx = FPS_NOIDX;
    float bx = 0.0f, by = 0.0f, bz = 0.0f;
    int pub = -1;                             // lane whose bucket is the wave's arg-max (wave-uniform); -1: no real point
    constexpr unsigned long long SLOT_LANES = (1ull << P) - 1ull;
    constexpr unsigned WAVE_LANES = (1u << NW) - 1u;

#ifdef CAMLI_FPS_PROFILE
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_ = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) picks[s] = cur;
        if (s == n_samples - 1) break;
        FPS_STAMP(4)

        // ---- box test, one bucket per lane ----
        const float ex = fmaxf(fmaxf(blo[0] - cx, cx - bhi[0]), 0.0f);
        const float ey = fmaxf(fmaxf(blo[1] - cy, cy - bhi[1]), 0.0f);
        const float ez = fmaxf(fmaxf(blo[2] - cz, cz - bhi[2]), 0.0f);
        const float lb = ex * ex + ey * ey + ez * ez;
        const unsigned todo = (unsigned)(__ballot(lb < bmax) & SLOT_LANES);
        FPS_STAMP(0)
        FPS_COUNT(5, __builtin_popcount(todo))
        FPS_COUNT(6, todo != 0u ? 1 : 0)
        FPS_COUNT(7, 1)
        if (todo != 0u) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if ((todo >> j) & 1u) {     // wave-uniform
#ifdef CAMLI_FPS_PROFILE
                    if (blockIdx.x == 0 && lane == 0) camli_fps_touch[w][j] += 1;
#endif
                    const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const float nd = fminf(dist[j], d);
                    dist[j] = nd;
                    const bool real = nd >= 0.0f;
                    const unsigned m = wave_allmax_u32(real ? __float_as_uint(nd) : 0u);
                    // a bucket in `todo` holds real points (bmax > lb >= 0); the lowest original index among its maxima
                    unsigned long long at = __ballot(real && __float_as_uint(nd) == m);
                    if (__builtin_popcountll(at) != 1) {
                        const unsigned low = wave_allmin_u32(real && __float_as_uint(nd) == m ? orig[j] : FPS_NOIDX);
                        at = __ballot(orig[j] == low);
                    }
                    const int src = (int)__builtin_ctzll(at);
                    const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)orig[j], src);
                    const float sx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(px[j]), src));
                    const float sy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(py[j]), src));
                    const float sz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(pz[j]), src));
                    if (myslot == j) {
                        bmax = __uint_as_float(m);
                        bidx = so;
                        bx = sx;
                        by = sy;
                        bz = sz;
                    }
                }
            }
            FPS_STAMP(1)
            // ---- wave arg-max over the bucket maxima (group-local: every group of P lanes holds the same P buckets) ----
            const bool live = bmax >= 0.0f;
            const unsigned kb = live ? __float_as_uint(bmax) : 0u;
            const unsigned wkey = group_allmax_u32<(P > 16 ? 32 : 16)>(kb);
            unsigned long long top = __ballot(live && kb == wkey) & SLOT_LANES;
            if (__builtin_popcountll(top) > 1) {
                const unsigned low = group_allmin_u32<(P > 16 ? 32 : 16)>(live && kb == wkey ? bidx : FPS_NOIDX);
                top = __ballot(live && kb == wkey && bidx == low) & SLOT_LANES;
            }
            pub = top != 0ull ? (int)__builtin_ctzll(top) : -1;
            FPS_STAMP(2)
        }

        // ---- one slot per wave, one barrier, every wave reduces the NW slots on its own ----
        FpsSlot& mine = slots[s & 1][w];
        if (lane == pub) {
            mine.key = __float_as_uint(bmax);
            mine.idx = bidx;
            mine.x = bx;
            mine.y = by;
            mine.z = bz;
        } else if (lane == 0 && pub < 0) {
            mine.key = 0u;
            mine.idx = FPS_NOIDX;
        }
        __syncthreads();
        FPS_STAMP(3)
        const FpsSlot* sp = &slots[s & 1][lane & (NW - 1)];
        const unsigned k2 = sp->key, i2 = sp->idx;
        const float x2 = sp->x, y2 = sp->y, z2 = sp->z;
        const unsigned gkey = group_allmax_u32<16>(k2);
        unsigned winners = (unsigned)__ballot(k2 == gkey) & WAVE_LANES;
        if (__builtin_popcount(winners) != 1) {
            const unsigned gidx = group_allmin_u32<16>(k2 == gkey ? i2 : FPS_NOIDX);
            winners = (unsigned)__ballot(k2 == gkey && i2 == gidx) & WAVE_LANES;
        }
        const int src = __builtin_ctz(winners);
        cur = __builtin_amdgcn_readlane((int)i2, src);
        cx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x2), src));
        cy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(y2), src));
        cz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(z2), src));
    }
#ifdef CAMLI_FPS_PROFILE
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 8; ++k) camli_fps_prof[w][k] = prof_[k];
#endif
    __syncthreads();
    for (int s = tid; s < n_samples; s += T) out[s] = (int64_t)picks[s];
}

template <int P, int T>
int launch_fps_pruned(const float* xyz, int64_t* out, int B, int N, int n_samples, hipStream_t stream) {
    int np2 = 64;
    while (np2 < N) np2 <<= 1;
    const size_t pick_bytes = (size_t)((n_samples + 3) & ~3) * sizeof(int);
    const size_t sort_bytes = (size_t)np2 * sizeof(unsigned long long);
    const size_t bytes = pick_bytes > sort_bytes ? pick_bytes : sort_bytes;
    static bool attr_set = false;       // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_pruned_kernel<P, T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)FPS_LDS_BUDGET);
        attr_set = true;
    }
    hipLaunchKernelGGL((fps_pruned_kernel<P, T>), dim3(B), dim3(T), bytes, stream, xyz, out, N, n_samples, np2);
    return camli_check_launch("camli_fps");
}

}  // namespace

extern "C" int camli_fps(const float* xyz, int64_t* out_idx, int B, int N, int n_samples, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!xyz || !out_idx) {
        camli_set_error("camli_fps: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || N < 1 || n_samples < 1 || n_samples > N) {
        camli_set_error("camli_fps: bad shape B=%d N=%d n_samples=%d", B, N, n_samples);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // Bucket-pruned kernel (same picks) for the clouds the models sample (8192 and 16384 points): measured on MI355X,
    // 16 x 8192 -> 4096: 4.16 -> 3.5 ms, 2 x 16384 -> 8192: 20.5 -> 7.5 ms (tools/ab_fps.py).  A pruned step costs
    // ~0.3 us plus ~0.33 us per bucket its busiest wave updates, so the first few hundred steps (most buckets live) are
    // slower than a full update and short runs on small clouds lose (64 x 4096 -> 1024: 0.68 -> 0.93 ms): those keep
    // the full-update kernel.  CAMLI_FPS=legacy|p16|p32|p8t1024|p16t1024 pins a variant (A/B runs, tests).
    const char* variant = getenv("CAMLI_FPS");
    const bool legacy = variant && !strcmp(variant, "legacy");
    if (variant && !legacy) {
        if (!strcmp(variant, "p16") && N <= 16 * 512) return launch_fps_pruned<16, 512>(xyz, out_idx, B, N, n_samples, s);
        if (!strcmp(variant, "p32") && N <= 32 * 512) return launch_fps_pruned<32, 512>(xyz, out_idx, B, N, n_samples, s);
        if (!strcmp(variant, "p8t1024") && N <= 8 * 1024) return launch_fps_pruned<8, 1024>(xyz, out_idx, B, N, n_samples, s);
        if (!strcmp(variant, "p16t1024") && N <= 16 * 1024) return launch_fps_pruned<16, 1024>(xyz, out_idx, B, N, n_samples, s);
    }
    if (!legacy && N > 4096 && N <= 16 * 1024) {
        if (N <= 16 * 512) return launch_fps_pruned<16, 512>(xyz, out_idx, B, N, n_samples, s);
        return launch_fps_pruned<16, 1024>(xyz, out_idx, B, N, n_samples, s);
    }
    // register-resident full update (every point, every step): threads per cloud 512 up to 8192 points, 1024 above
    const int T = N <= 8192 ? 512 : 1024;
    const int P = camli_divup(N, T);
#define CAMLI_FPS_CASE(PP)                                                        \
    if (P <= PP) return T == 512 ? launch_fps<PP, 512>(xyz, out_idx, B, N, n_samples, s) \
                                 : launch_fps<(PP > 24 ? 24 : PP), 1024>(xyz, out_idx, B, N, n_samples, s)
    CAMLI_FPS_CASE(1);
    CAMLI_FPS_CASE(2);
    CAMLI_FPS_CASE(4);
    CAMLI_FPS_CASE(8);
    CAMLI_FPS_CASE(16);
    CAMLI_FPS_CASE(24);
#undef CAMLI_FPS_CASE
    camli_set_error("camli_fps: N=%d exceeds the register-resident limit of %d points per cloud", N, 24 * T);
    return CAMLI_ENOTSUP;
}

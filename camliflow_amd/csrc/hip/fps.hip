// Furthest-point sampling, gfx950.
//
// Replaces models/csrc/furthest_point_sampling/furthest_point_sampling_kernel.cu:34-85 of the
// reference (1024 threads, every step re-reads xyz + dist from global memory and tree-reduces
// through shared memory with 11 barriers).  Index-exact against oracle_fps
// (oracle/camli_oracle.c): start at index 0, dist init 1e10, unfused fp32 distance, arg-max with
// the LOWEST index among equal maxima.
//
// Design (CDNA4): the algorithm is a chain of n_samples dependent steps, so the only lever is
// the latency of one step.  One 1024-thread workgroup (16 waves, one CU) owns a cloud; every
// thread keeps P = ceil(N/1024) points AND their running distances in VGPRs for the whole kernel
// (no memory traffic inside the loop: the algorithmic traffic is one read of the cloud and one
// write of the indices).  Per step: register update + thread-local arg-max, wave arg-max with
// DPP / permlane-swap (no LDS), one LDS slot per wave carrying (dist, index, xyz of the wave's
// winner), ONE barrier (slots are double-buffered by step parity), then every wave re-reduces
// the 16 slots itself.  Distances are non-negative, so their fp32 bit patterns order like
// unsigned integers and the reductions run as u32 max / u32 min.
#include "camli_common.h"

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / 64;

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
    float x, y, z;
    unsigned pad[3];
};

// point i = j*1024 + tid  (j = register slot).  Slots past N carry dist = -1 bits?  No: keys are
// compared as unsigned, so padding points use key 0 with index 0xffffffff and can only win when
// every real distance is 0 too, in which case the index min still prefers a real point.
template <int P, bool PICKS_IN_LDS>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz_all,
                                                           int64_t* __restrict__ out_all, int N, int n_samples) {
    __shared__ FpsSlot slots[2][FPS_WAVES];
    // picks are collected in LDS and written once at the end: a global store inside the step loop
    // drags an `s_waitcnt vmcnt(0)` (store round trip to L2) into every one of the n_samples steps
    extern __shared__ int picks[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
    int64_t* __restrict__ out = out_all + (size_t)blockIdx.x * n_samples;

    float px[P], py[P], pz[P], dist[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        int i = j * FPS_THREADS + tid;
        bool ok = i < N;
        int ic = ok ? i : 0;
        px[j] = xyz[ic * 3 + 0];
        py[j] = xyz[ic * 3 + 1];
        pz[j] = xyz[ic * 3 + 2];
        dist[j] = ok ? 1e10f : -1.0f;  // negative: never updated upward, never selected (see below)
    }

    int cur = 0;
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];

    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) {
            if (PICKS_IN_LDS) picks[s] = cur; else out[s] = (int64_t)cur;
        }
        if (s == n_samples - 1) break;

        // ---- register update + thread-local arg-max (lowest slot wins ties: strict >) ----
        float best = -1.0f;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            float d = dx * dx + dy * dy + dz * dz;
            float nd = fminf(dist[j], d);
            dist[j] = nd;
            bool gt = nd > best;
            best = gt ? nd : best;
            bj = gt ? j : bj;
        }
        // padding points hold -1 forever (min(-1, d) = -1); a thread with only padding reports
        // best = -1 -> clamp to key 0 / index max so it cannot beat a real point
        const bool real = best >= 0.0f;
        const unsigned key = real ? __float_as_uint(best) : 0u;
        const unsigned bi = real ? (unsigned)(bj * FPS_THREADS + tid) : 0xffffffffu;

        // ---- wave arg-max ----
        const unsigned wkey = wave_allmax_u32(key);
        const unsigned widx = wave_allmin_u32(key == wkey ? bi : 0xffffffffu);
        if (bi == widx && widx != 0xffffffffu) {
            // exactly one lane: the owner of the wave's winner publishes its coordinates
            float ox = px[0], oy = py[0], oz = pz[0];
#pragma unroll
            for (int j = 1; j < P; ++j) {
                bool m = bj == j;
                ox = m ? px[j] : ox;
                oy = m ? py[j] : oy;
                oz = m ? pz[j] : oz;
            }
            FpsSlot& sl = slots[s & 1][w];
            sl.key = wkey;
            sl.idx = widx;
            sl.x = ox;
            sl.y = oy;
            sl.z = oz;
        } else if (lane == 0 && widx == 0xffffffffu) {
            FpsSlot& sl = slots[s & 1][w];
            sl.key = 0u;
            sl.idx = 0xffffffffu;
        }
        __syncthreads();

        // ---- every wave reduces the 16 slots on its own (no second barrier) ----
        const FpsSlot* sp = &slots[s & 1][lane & (FPS_WAVES - 1)];
        const unsigned k2 = sp->key;
        const unsigned i2 = sp->idx;
        const unsigned gkey = wave_allmax_u32(k2);
        const unsigned gidx = wave_allmin_u32(k2 == gkey ? i2 : 0xffffffffu);
        const unsigned long long hit = __ballot(k2 == gkey && i2 == gidx);
        const int wsel = __builtin_ctzll(hit) & (FPS_WAVES - 1);
        const FpsSlot* win = &slots[s & 1][wsel];
        cur = (int)gidx;
        cx = win->x;
        cy = win->y;
        cz = win->z;
    }
    if (PICKS_IN_LDS) {
        __syncthreads();
        for (int s = tid; s < n_samples; s += FPS_THREADS) out[s] = (int64_t)picks[s];
    }
}

template <int P>
int launch_fps(const float* xyz, int64_t* out, int B, int N, int n_samples, hipStream_t stream) {
    const size_t pick_bytes = (size_t)n_samples * sizeof(int);
    if (pick_bytes <= 96 * 1024)
        hipLaunchKernelGGL((fps_kernel<P, true>), dim3(B), dim3(FPS_THREADS), pick_bytes, stream, xyz, out, N, n_samples);
    else
        hipLaunchKernelGGL((fps_kernel<P, false>), dim3(B), dim3(FPS_THREADS), 0, stream, xyz, out, N, n_samples);
    return camli_check_launch("camli_fps");
}

}  // namespace

extern "C" int camli_fps(const float* xyz, int64_t* out_idx, int B, int N, int n_samples, void* stream) {
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
    const int P = camli_divup(N, FPS_THREADS);
    if (P <= 1) return launch_fps<1>(xyz, out_idx, B, N, n_samples, s);
    if (P <= 2) return launch_fps<2>(xyz, out_idx, B, N, n_samples, s);
    if (P <= 4) return launch_fps<4>(xyz, out_idx, B, N, n_samples, s);
    if (P <= 8) return launch_fps<8>(xyz, out_idx, B, N, n_samples, s);
    if (P <= 16) return launch_fps<16>(xyz, out_idx, B, N, n_samples, s);
    if (P <= 24) return launch_fps<24>(xyz, out_idx, B, N, n_samples, s);
    camli_set_error("camli_fps: N=%d exceeds the register-resident limit of %d points per cloud", N,
                    24 * FPS_THREADS);
    return CAMLI_ENOTSUP;
}

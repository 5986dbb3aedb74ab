// Weight gradient of the channels-last tap convolution (convcl.h) on the fp32 matrix cores, gfx950.
//
//     part[s][t][c][n] = sum over the pixels p of K range s of   x[p + (dy_t, dx_t)][c] * gy[p][n]      (zero outside the image)
//
// The contraction runs over PIXELS, i.e. both operands are "k-major" exactly as NHWC tensors lie in memory (a k row = the
// channels of one pixel, contiguous): the machinery of gemm_w128.h -- one global_load_lds row per pixel straight to LDS,
// ds_read_b128 fragments along the channels, 128-row wave tiles on accumulators pinned to the ACC registers -- applies
// unchanged.  What is added:
//  * the tap shift and the zero padding of the x operand: the row of pixel p + (dy, dx) is a wave-uniform decision per
//    DMA instruction (one instruction = one pixel's channels), taken on scalar registers; an outside row gets an
//    out-of-range buffer offset and arrives as zeros.  The (x, y) coordinates of the four rows a wave brings per step are
//    advanced incrementally (no division in the loop).
//  * split K: T taps x Cin / 256 row tiles are a handful of output tiles, so the pixel range is cut into S parts, one
//    workgroup per (part, tap, tile); parts are written to a workspace and added in a fixed order by wrw_reduce_kernel,
//    which also writes the [Cout][Cin][T] layout of the framework's weight tensor.  Deterministic, no atomics.
// M = input channels (x rows are the A operand), N = output channels (gy): the result [t][c][n] is also the packing the
// data-gradient convolution reads (convcl.h: w'[c][t][n] up to the order of the first two indices).
#pragma once
#include "convcl.h"

namespace wrw {

using ccl::f32x4;
using ccl::u32x4;
using ccl::OOB;

struct Problem {
    const float* x;       // input channels [0, C0): NHWC [P][ldx]
    const float* x1;      // input channels [C0, Cin): NHWC [P][ldx1] (unused when C0 == Cin)
    const float* zero;    // 1 KB of zeros (the source of padded rows)
    const float* gy;      // NHWC [P][ldg], channels [0, Cout)
    float* part;          // [S][T][Cin][Cout]
    int B, H, W, Cin, Cout, T;
    int C0, ldx, ldx1, ldg;
    int S, ksplit;        // parts and pixels per part (a multiple of 16)
    int tiles_m, tiles_n; // Cin / 256, Cout / (32 TBN)
    signed char dy[ccl::MAX_TAPS], dx[ccl::MAX_TAPS];
};

// 4 MFMAs c_j += a x b[j]
template <bool ZERO>
__device__ __forceinline__ void mfma_x4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float a, const f32x4& b) {
    ccl::mfma_x4<ZERO>(c0, c1, c2, c3, a, b[0], b[1], b[2], b[3]);
}

// TBN: 16-column tiles per wave along N (8: 256 output channels per workgroup, 4: 128).  Workgroup tile 256 (input
// channels) x 32 TBN, waves 2 x 2.  LDS: NBUF stages of 16 pixel rows x (256 + 32 TBN) floats.
template <int TBN, int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wrw_kernel(Problem p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(NBUF >= 3 && (TBN == 8 || TBN == 4), "");
    constexpr int NB = 32 * TBN, WN = 16 * TBN;
    constexpr int STAGE = 16 * (256 + NB);        // floats: [16][256] of x rows, then [16][NB] of gy rows
    constexpr int GSLOTS = NB == 256 ? 4 : 2;     // gy DMA instructions per wave and step (128-channel rows: two per instruction)
    constexpr int IPS = 4 + GSLOTS;               // DMA instructions per wave and step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int P = p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy), 0, (int)((uint32_t)P * p.ldg * 4u), 0x00020000);

    // unit -> (part, tap, tile): consecutive workgroups share a part (the same pixels)
    int u = blockIdx.x;
    const int tn = u % p.tiles_n; u /= p.tiles_n;
    const int tm = u % p.tiles_m; u /= p.tiles_m;
    const int tap = u % p.T, part = u / p.T;
    const int k_begin = part * p.ksplit, k_end = min(P, k_begin + p.ksplit);
    const int steps = (k_end - k_begin + 15) >> 4;
    if (steps <= 0) return;       // (the reduce kernel never reads a part without pixels: S is chosen by the host)
    const int dy = p.dy[tap], dx = p.dx[tap];
    const int shift = dy * p.W + dx;

    // the four x rows this wave brings per step are pixels k0 + wave + 4 j; their image coordinates, advanced by 16 per step
    // (named scalars: indexed arrays captured by the lambdas below end up in scratch memory)
    int px0, px1, px2, px3, py0, py1, py2, py3;
    auto coords = [&](int j, int& px, int& py) {
        const int pix = k_begin + wave + 4 * j;
        px = pix % p.W;
        py = (pix / p.W) % p.H;
    };
    coords(0, px0, py0); coords(1, px1, py1); coords(2, px2, py2); coords(3, px3, py3);
    int l_k = k_begin, l_left = steps;
    // x rows through per-lane pointers (global_load_lds): the 256 channels of a tile may come from two tensors, which one
    // buffer descriptor cannot address; a padded row is read from the zero page instead
    const int ch = tm * 256 + 4 * lane;
    const bool sec = ch >= p.C0;
    const int64_t lane_ld = sec ? p.ldx1 : p.ldx;
    const float* const lane_zero = p.zero + 4 * lane;
    const float* const lane_x = (sec ? p.x1 + (ch - p.C0) : p.x + ch) + (int64_t)(k_begin + wave + shift) * lane_ld;
    const float *xrow0 = lane_x, *xrow1 = lane_x + 4 * lane_ld, *xrow2 = lane_x + 8 * lane_ld, *xrow3 = lane_x + 12 * lane_ld;
    // gy rows: NB = 256 -> one row per instruction; NB = 128 -> a row is 512 bytes, an instruction brings rows 2 i and
    // 2 i + 1 (lanes 0-31 / 32-63; the LDS destination is lane-linear, so the pair must be adjacent in LDS)
    const uint32_t lane_g = (uint32_t)(tn * NB + 4 * (lane & (NB / 4 - 1))) * 4u;
    const int lane_grow = NB == 256 ? 0 : (lane >> 5);
    auto dma_slot = [&](int buf, int slot) {
        float* st = lds + buf * STAGE;
        if (slot < 4) {
            auto xdma = [&](int j, const float*& xrow, int& px, int& py) {
                const int pix = l_k + wave + 4 * j;
                const bool ok = pix < k_end && (unsigned)(px + dx) < (unsigned)p.W && (unsigned)(py + dy) < (unsigned)p.H;
                __builtin_amdgcn_global_load_lds(ok ? xrow : lane_zero, (__attribute__((address_space(3))) void*)(st + (wave + 4 * j) * 256), 16, 0, 0);
                xrow += 16 * lane_ld;
                px += 16;
                while (px >= p.W) { px -= p.W; if (++py == p.H) py = 0; }
            };
            if (slot == 0) xdma(0, xrow0, px0, py0);
            else if (slot == 1) xdma(1, xrow1, px1, py1);
            else if (slot == 2) xdma(2, xrow2, px2, py2);
            else xdma(3, xrow3, px3, py3);
        } else {
            const int i = wave + 4 * (slot - 4);              // instruction index within the stage
            const int pix = l_k + (NB == 256 ? i : 2 * i + lane_grow);
            const uint32_t row = pix < k_end ? (uint32_t)pix * (uint32_t)p.ldg * 4u : OOB;
            ccl::dma16(rs_g, lane_g + row, st + 16 * 256 + i * 256);
            if (slot == IPS - 1) { l_k += 16; --l_left; }
        }
    };

#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
#pragma unroll
        for (int s = 0; s < IPS; ++s)
            if (i < steps) dma_slot(i, s);
    if (steps >= NBUF - 1) ccl::wait_vm<IPS*(NBUF - 2)>(); else ccl::wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const float* fa = lds + q * 256 + 128 * wm + 4 * r;                 // x rows: k row q of a sub-step, 4 consecutive channels
    const float* fb = lds + 16 * 256 + q * NB + WN * wn + 4 * r;
    f32x4 acc[8][TBN];
    f32x4 pa[2], pb[TBN / 4], qa[2], qb[TBN / 4];
    auto read_frag = [&](int buf, int s, int idx, f32x4 (&ya)[2], f32x4 (&yb)[TBN / 4]) {      // idx 0 .. 1 + TBN / 4
        if (idx < 2) ya[idx] = *reinterpret_cast<const f32x4*>(fa + buf * STAGE + s * 1024 + idx * 64);
        else yb[idx - 2] = *reinterpret_cast<const f32x4*>(fb + buf * STAGE + s * 4 * NB + (idx - 2) * 64);
    };
    constexpr int NFR = 2 + TBN / 4;
#pragma unroll
    for (int i = 0; i < NFR; ++i) read_frag(0, 0, i, pa, pb);

    // sub-step (K = 4): 8 x TBN / 4 slots of 4 MFMAs
    auto substep = [&](auto zero, const f32x4 (&xa)[2], const f32x4 (&xb)[TBN / 4], auto&& between) {
#pragma unroll
        for (int sl = 0; sl < 2 * TBN; ++sl) {
            const int ta = sl / (TBN / 4), hb = sl % (TBN / 4);
            mfma_x4<decltype(zero)::value>(acc[ta][4 * hb], acc[ta][4 * hb + 1], acc[ta][4 * hb + 2], acc[ta][4 * hb + 3],
                                           xa[ta >> 2][ta & 3], xb[hb]);
            __builtin_amdgcn_sched_barrier(0);
            between(sl);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int buf = 0;
    auto step = [&](auto first, int st) {
        const bool more = st + 1 < steps;
        const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;
        const int fbuf = buf == 0 ? NBUF - 1 : buf - 1;
        const bool issue = l_left > 0;
        // sub-steps 0 .. 2 prefetch the next sub-step's fragments; the last one syncs, refills and prefetches the next stage
        substep(first, pa, pb, [&](int sl) { if (sl >= 1 && sl <= NFR) read_frag(buf, 1, sl - 1, qa, qb); });
        substep(std::false_type{}, qa, qb, [&](int sl) { if (sl >= 1 && sl <= NFR) read_frag(buf, 2, sl - 1, pa, pb); });
        substep(std::false_type{}, pa, pb, [&](int sl) { if (sl >= 1 && sl <= NFR) read_frag(buf, 3, sl - 1, qa, qb); });
        // actions of the last sub-step: sync, IPS DMAs, NFR fragment reads -- spread over its 2 TBN slots
        constexpr int NACT = 1 + IPS + NFR, NSL = 2 * TBN;
        substep(std::false_type{}, qa, qb, [&](int sl) {
#pragma unroll
            for (int a = 0; a < NACT; ++a) {
                if (a * NSL / NACT != sl) continue;
                if (a == 0) {
                    if (issue) ccl::wait_vm<IPS*(NBUF - 3)>(); else ccl::wait_vm<0>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                } else if (a <= IPS) {
                    if (issue) dma_slot(fbuf, a - 1);
                } else if (more) {
                    read_frag(nbuf, 0, a - IPS - 1, pa, pb);
                }
            }
        });
        buf = nbuf;
    };
    step(std::true_type{}, 0);
    for (int st = 1; st < steps; ++st) step(std::false_type{}, st);

    // ---- epilogue: part[part][tap][m][n], rows m = 128 wm + 64 (ta / 4) + 16 q + 4 v + ta % 4, 4 consecutive n per lane
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
    float* out = p.part + ((size_t)(part * p.T + tap) * p.Cin + tm * 256) * p.Cout + tn * NB;
    const uint32_t obase = ((uint32_t)(128 * wm + 16 * q) * p.Cout + WN * wn + 4 * r) * 4u;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(256u * p.Cout * 4u), 0x00020000);
#pragma unroll
    for (int ta = 0; ta < 8; ++ta) {
        if (TBN == 8)
            asm volatile("" : "+a"(acc[ta][0]), "+a"(acc[ta][1]), "+a"(acc[ta][2]), "+a"(acc[ta][3]), "+a"(acc[ta][4]),
                         "+a"(acc[ta][5]), "+a"(acc[ta][6]), "+a"(acc[ta][7]));
        else
            asm volatile("" : "+a"(acc[ta][0]), "+a"(acc[ta][1]), "+a"(acc[ta][2]), "+a"(acc[ta][3]));
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint32_t row_off = (uint32_t)(64 * (ta >> 2) + 4 * v + (ta & 3)) * (uint32_t)p.Cout * 4u;
#pragma unroll
            for (int hb = 0; hb < TBN / 4; ++hb) {
                f32x4 o = {acc[ta][4 * hb][v], acc[ta][4 * hb + 1][v], acc[ta][4 * hb + 2][v], acc[ta][4 * hb + 3][v]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_o, obase + row_off + 256u * hb, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// gw[(n * Cin + c) * T + t] (+)= sum_s part[s][t][c][n]   (parts added in order; thread = one (t, c, n)).
// swapped != 0: the parts were computed with the operands' roles exchanged (camli_convcl_wrw on 128-wide inputs): part[s][t][n][c],
// i.e. the kernel's M index is the output channel -- `Cin` / `Cout` are still those of the convolution.
static __global__ __launch_bounds__(256) void wrw_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int S, int T,
                                                          int Cin, int Cout, int accumulate, int swapped) {
    const size_t n_el = (size_t)T * Cin * Cout;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_el) return;
    float acc = part[i];
    for (int s = 1; s < S; ++s) acc += part[(size_t)s * n_el + i];
    const int inner = swapped ? Cin : Cout, outer = swapped ? Cout : Cin;
    const int a = (int)(i % inner);
    const int b = (int)((i / inner) % outer);
    const int t = (int)(i / ((size_t)Cout * Cin));
    const int n = swapped ? b : a, c = swapped ? a : b;
    float* dst = gw + ((size_t)n * Cin + c) * T + t;
    *dst = accumulate ? *dst + acc : acc;
}

}  // namespace wrw

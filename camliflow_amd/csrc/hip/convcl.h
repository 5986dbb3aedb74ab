// Channels-last ("tap") convolution as an implicit GEMM on the fp32 matrix cores, gfx950: the wave-tile machinery of
// gemm_w128.h (128 x 128 per wave, one wave per SIMD, asm MFMA rows on accumulators pinned to the ACC registers, pinned
// instruction order, counted vmcnt, one barrier per K step) for operands whose CONTRACTION index is the contiguous one.
//
//     y[p][n] = sum_t sum_c x[p + (dy_t, dx_t)][c] * w[n][t][c]          p = (b, y, x) pixel, zero outside the image
//
// x NHWC [B][H][W][ldx] (channels 0 .. Cin - 1 used), w packed [Cout][T][Cin], y NHWC [B*H*W][ldy].  Stride 1, any tap
// list: GRU2D's 1x5 / 5x1 convolutions (models/raft_core.py:110-140) are T = 5 with (0, t - 2) / (t - 2, 0); their data
// gradient is the same kernel on the negated taps and the transposed packing (w'[c][t][n]).
//
//  * GEMM roles: MFMA rows i = output channels (weights are the MFMA A operand), MFMA columns j = pixels.  A lane then
//    holds 4 consecutive output channels of one pixel per accumulator quad: one 16-byte store, 64 contiguous bytes per
//    pixel and instruction, no shuffles.
//  * K is walked in steps of 16 channels x one tap.  Both operands sit in LDS as [row][16 floats] (64-byte rows, row =
//    output channel resp. pixel of the tile); a fragment read is ONE ds_read_b128 per 16-row tile and step: lane
//    (r = lane % 16, q = lane / 16) takes floats 4 q .. 4 q + 3 of row r, the four registers feed four MFMAs (the K
//    index of an MFMA is arbitrary as long as both operands agree: MFMA j of a step contracts k = 4 q + j).  64-byte rows
//    put rows r and r + 4 on the same banks, so the 16-byte slot s of row R is stored at slot s ^ bank_swizzle(R >> 2)
//    (below): conflict-free over the lane groups the LDS serves a b128 read in, and free to produce, because ...
//  * ... operands go DIRECT TO LDS (buffer_load_dwordx4 ... lds: the destination is lane-linear, the source address is per
//    lane).  The per-lane source carries the swizzle, the tap shift, and the ZERO PADDING: a lane whose source pixel lies
//    outside the image gets an out-of-range buffer offset and the hardware writes zeros.  No halo tiles, no padded
//    layouts, no branches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ccl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_TAPS = 32;

// Slot permutation of a 64-byte LDS row, by (row >> 2) & 3.  A ds_read_b128 is served in four groups of 16 lanes that are
// NOT lanes 16 g .. 16 g + 15 but {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS
// table): with lane = 16 q + r the first group reads (q 0, rows 0-3), (q 0, rows 12-15), (q 1, rows 4-11).  Rows r and r + 4
// share banks, so within a group the four (q, r >> 2) pairs of one r & 3 class need four different physical slots:
// slot = q ^ g(r >> 2) with g = {0, 2, 3, 1} does it for all four groups (q ^ (r >> 2), the obvious choice, is 2-way
// conflicted on every read: SQ_LDS_BANK_CONFLICT 5.2 M cycles per launch, profiles/r05_experiments.txt item 9).
__device__ __forceinline__ int bank_swizzle(int row_quad) { return (0x78 >> (2 * (row_quad & 3))) & 3; }
constexpr uint32_t OOB = 0x7FFFFFF0u;       // beyond any num_records this file accepts (< 2^31 - 2^20)

struct Problem {
    const float* x;               // input channels [0, C0): NHWC [P][ldx]
    const float* x1;              // input channels [C0, Cin): NHWC [P][ldx1] (unused when C0 == Cin)
    const float* w;               // [Cout][T][Cin]
    float* y;                     // output channels [0, N0): [P][ldy]
    float* y1;                    // output channels [N0, Cout): [P][ldy1] (unused when N0 == Cout)
    int B, H, W, Cin, Cout, T;
    int C0, N0;                   // multiples of 16 / of the wave's channel range
    int ldx, ldx1, ldy, ldy1;     // floats per pixel
    int ldw;                      // floats per weight row (T * Cin for the packed weights)
    int xk, wk;                   // floats between consecutive 16-channel K chunks of a row (16: rows are contiguous in k)
    uint32_t xrec, x1rec, wrec;   // bytes the buffer descriptors of x / x1 / w span (P * ldx * 4, ..., Cout * ldw * 4 for dense rows)
    int tiles_p, tiles_n;         // pixel tiles (256) x channel tiles (NT)
    // epilogue operands (see the EPI_* forms below)
    const float* add;             // [P][ld_add]
    const float* h;               // [P][ld_h]
    const float* z;               // [P][ld_z]
    float* y2;                    // [P][ldy2]
    int ld_add, ld_h, ld_z, ldy2;
    int acc0, acc1;               // EPI_PLAIN: add into y / y1 instead of overwriting
    int sanitize;                 // EPI_BLEND: torch.nan_to_num on h'
    signed char dy[MAX_TAPS], dx[MAX_TAPS];
};

// Epilogues.  PLAIN: y / y1 (= or +=) the convolution.  The two GRU2D forms take the convolution as the pre-activation of a
// half-step (models/raft_core.py:124-130 / 132-138, context term hoisted: cores/raft2d.GRU2D.prepare):
//   GATES (Cout = 256 = z | r):  z = sigmoid(conv[:128] + add[:128]) -> y;   r = sigmoid(conv[128:] + add[128:]) -> y2,  r * h -> y1
//   BLEND (Cout = 128):          q = tanh(conv + add) -> y1;   h' = (1 - z) h + z q (nan_to_num when `sanitize`) -> y
// all operands NHWC; the arithmetic is that of gru.hip's stand-alone kernels.
enum { EPI_PLAIN = 0, EPI_GATES = 1, EPI_BLEND = 2 };

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float nan_to_num_(float v) { return v != v ? 0.0f : fminf(fmaxf(v, -3.402823466e+38f), 3.402823466e+38f); }
__device__ __forceinline__ f32x4 load4(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ void store4(f32x4 v, __amdgpu_buffer_rsrc_t rs, uint32_t off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 0);
}

template <int V>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((V & 15) | (7 << 4) | (15 << 8) | ((V >> 4) << 14));
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voffset, float* lds_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_uniform, 16, voffset, 0, 0, 0);
}

// 4 MFMAs c_j += a x b_j (asm: accumulators stay in the ACC half of the register file, see gemm_w128.h)
template <bool ZERO>
__device__ __forceinline__ void mfma_x4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float a, float b0, float b1, float b2,
                                        float b3) {
    if (ZERO) {
        asm volatile(
            "s_nop 1\n\t"
            "v_mfma_f32_16x16x4_f32 %0, %4, %5, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %1, %4, %6, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %4, %7, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %3, %4, %8, 0"
            : "=a"(c0), "=a"(c1), "=a"(c2), "=a"(c3)
            : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        return;
    }
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %4, %6, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %4, %8, %3"
        : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
        : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

// NTW: 16-channel tiles per wave along the output channels (8: 256-channel workgroup tile, 6: 192, 4: 128, 2: 64 -- the small
// tiles are for problems whose pixel tiles alone do not fill the chip: batch 4 at 68 x 120 is 128 pixel tiles on 256 CUs).
// Workgroup tile =
// 256 pixels x 32 NTW channels, waves 2 (pixels) x 2 (channels), 128 pixels x 16 NTW channels each.  NBUF LDS stages of
// (256 + 32 NTW) rows x 64 bytes.  The body is a device function: convcl_kernel runs it on the problem it is launched with,
// winograd_wrw.h on one problem per (plane, K split) of a batched, split contraction.
// What a workgroup contracts: the problem's own operands (convcl_kernel) or a slice of them -- one plane and K range of a
// batched, split contraction (winograd_wrw.h).  Kept apart from Problem so that the latter stays in the kernel arguments
// (its tap tables are indexed at run time: a modified private copy would live in scratch memory).
struct Operands {
    const float* x; const float* x1; const float* w;
    float* y; float* y1;
    int Cin, C0;
};

template <int NTW, int NBUF, int EPI = EPI_PLAIN>
__device__ __forceinline__ void convcl_body(const Problem& p, const Operands& o, float* lds) {
    static_assert(NBUF >= 3 && (NTW == 8 || NTW == 6 || NTW == 4 || NTW == 2), "");
    constexpr int NT = 32 * NTW;                  // output channels per workgroup tile
    constexpr int WROWS = 16 * NTW;               // per wave
    constexpr int STAGE = (256 + NT) * 16;        // floats: pixel rows, then weight rows
    constexpr int IPX = 4, IPW = NT / 64;         // DMA instructions per wave and step: pixel rows / weight rows
    constexpr int IPS = IPX + IPW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int P = p.B * p.H * p.W;
    const int chunks = o.Cin >> 4, steps = chunks * p.T;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.x), 0, (int)p.xrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.x1), 0, (int)p.x1rec, 0x00020000);
    const int chunks0 = o.C0 >> 4;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.w), 0, (int)p.wrec, 0x00020000);

    // DMA geometry of this lane: row (lane >> 2) of a 16-row group, physical 16-byte slot lane & 3 holds logical slot
    // (lane & 3) ^ bank_swizzle(row >> 2)
    const int drow = lane >> 2, kq = (lane & 3) ^ bank_swizzle(lane >> 4);
    // fragment geometry: row r of a 16-row tile, logical slot q
    const int fslot = q ^ bank_swizzle(r >> 2);
    const float* fw = lds + 256 * 16 + (WROWS * wn + r) * 16 + fslot * 4;      // + tile * 256 floats
    const float* fx = lds + (128 * wm + r) * 16 + fslot * 4;

    for (int tile = blockIdx.x; tile < p.tiles_p * p.tiles_n; tile += gridDim.x) {
        const int tp = tile / p.tiles_n, tn = tile - tp * p.tiles_n;
        const int p0 = tp * 256, n0 = tn * NT;
        // ---- this lane's four pixel rows: byte offset of (pixel, logical slot) and the taps that stay inside the image
        uint32_t xoff[IPX], xoff1[IPX], xmask[IPX];
#pragma unroll
        for (int j = 0; j < IPX; ++j) {
            const int pix = p0 + 64 * wave + 16 * j + drow;
            const int pc = min(pix, P - 1);
            const int xx = pc % p.W, yy = (pc / p.W) % p.H;
            uint32_t m = 0;
            for (int t = 0; t < p.T; ++t) {
                const bool ok = pix < P && (unsigned)(xx + p.dx[t]) < (unsigned)p.W && (unsigned)(yy + p.dy[t]) < (unsigned)p.H;
                m |= (ok ? 1u : 0u) << t;
            }
            xmask[j] = m;
            xoff[j] = ((uint32_t)pc * p.ldx + kq * 4) * 4u;
            xoff1[j] = ((uint32_t)pc * p.ldx1 + kq * 4) * 4u;
        }
        uint32_t woff[IPW];
#pragma unroll
        for (int j = 0; j < IPW; ++j) woff[j] = ((uint32_t)(n0 + 16 * (wave + 4 * j) + drow) * p.ldw + kq * 4) * 4u;

        int l_chunk = 0, l_tap = 0, l_left = steps;
        // one DMA of the stage being loaded (slot 0 .. IPS - 1); the step's scalars are read when slot 0 is issued
        int s_xshift = 0, s_wshift = 0, s_tap = 0;
        bool s_second = false;
        auto dma_slot = [&](int buf, int slot) {
            float* st = lds + buf * STAGE;
            if (slot == 0) {
                s_tap = l_tap;
                s_second = l_chunk >= chunks0;
                s_xshift = s_second ? ((p.dy[l_tap] * p.W + p.dx[l_tap]) * p.ldx1 + (l_chunk - chunks0) * p.xk) * 4
                                    : ((p.dy[l_tap] * p.W + p.dx[l_tap]) * p.ldx + l_chunk * p.xk) * 4;
                s_wshift = (l_tap * o.Cin + l_chunk * p.wk) * 4;
            }
            if (slot < IPX) {
                const uint32_t v = ((xmask[slot] >> s_tap) & 1u) ? (s_second ? xoff1[slot] : xoff[slot]) + (uint32_t)s_xshift : OOB;
                if (s_second) dma16(rs_x1, v, st + (64 * wave + 16 * slot) * 16);
                else dma16(rs_x, v, st + (64 * wave + 16 * slot) * 16);
            } else {
                const int j = slot - IPX;
                dma16(rs_w, woff[j] + (uint32_t)s_wshift, st + (256 + 16 * (wave + 4 * j)) * 16);
            }
            if (slot == IPS - 1) {
                --l_left;
                if (++l_tap == p.T) { l_tap = 0; ++l_chunk; }
            }
        };

        // ---- prologue ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i)
#pragma unroll
            for (int s = 0; s < IPS; ++s) dma_slot(i, s);
        wait_vm<IPS*(NBUF - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        f32x4 acc[NTW][8];
        f32x4 pw[NTW], px[8], qw[NTW], qx[8];
        auto read_frag = [&](int buf, int idx, f32x4 (&yw)[NTW], f32x4 (&yx)[8]) {      // idx 0 .. NTW + 7
            if (idx < NTW) yw[idx] = *reinterpret_cast<const f32x4*>(fw + buf * STAGE + idx * 256);
            else yx[idx - NTW] = *reinterpret_cast<const f32x4*>(fx + buf * STAGE + (idx - NTW) * 256);
        };
#pragma unroll
        for (int i = 0; i < NTW + 8; ++i) read_frag(0, i, pw, px);

        // one step = 4 (k) x NTW (channel tile) x 2 (pixel half) slots of 4 MFMAs; `between(slot)` runs in the shadow of
        // the slot's last MFMA
        auto step = [&](auto zero, const f32x4 (&xw)[NTW], const f32x4 (&xx)[8], auto&& between) {
#pragma unroll
            for (int sl = 0; sl < 8 * NTW; ++sl) {
                const int j = sl / (2 * NTW), tco = (sl >> 1) % NTW, h = sl & 1;
                if (j == 0)
                    mfma_x4<decltype(zero)::value>(acc[tco][4 * h], acc[tco][4 * h + 1], acc[tco][4 * h + 2], acc[tco][4 * h + 3],
                                                   xw[tco][j], xx[4 * h][j], xx[4 * h + 1][j], xx[4 * h + 2][j], xx[4 * h + 3][j]);
                else
                    mfma_x4<false>(acc[tco][4 * h], acc[tco][4 * h + 1], acc[tco][4 * h + 2], acc[tco][4 * h + 3], xw[tco][j],
                                   xx[4 * h][j], xx[4 * h + 1][j], xx[4 * h + 2][j], xx[4 * h + 3][j]);
                __builtin_amdgcn_sched_barrier(0);
                between(sl);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int buf = 0;
        auto full_step = [&](auto zero, auto parity, int st) {
            const bool more = st + 1 < steps;
            const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;
            const int fbuf = buf == 0 ? NBUF - 1 : buf - 1;
            const bool issue = l_left > 0;
            auto between = [&](int sl) {
                if (sl == 1) {
                    // the next stage has landed (this wave's share), every wave is done with the previous one
                    if (issue) wait_vm<IPS*(NBUF - 3)>(); else wait_vm<0>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                if (sl >= 2 && sl < 2 + IPS && issue) dma_slot(fbuf, sl - 2);
                // fragments of the next step: RPS reads per slot behind the DMA slots (one on the wide tiles; the 64-channel
                // tile has 16 slots for 1 barrier + 5 DMAs + 10 reads)
                constexpr int RPS = (NTW + 8 + (8 * NTW - 2 - IPS) - 1) / (8 * NTW - 2 - IPS);
                if (more && sl >= 2 + IPS) {
#pragma unroll
                    for (int i = (sl - 2 - IPS) * RPS; i < (sl - 1 - IPS) * RPS && i < NTW + 8; ++i) {
                        if (decltype(parity)::value) read_frag(nbuf, i, pw, px);
                        else read_frag(nbuf, i, qw, qx);
                    }
                }
            };
            if (decltype(parity)::value) step(zero, qw, qx, between);
            else step(zero, pw, px, between);
            buf = nbuf;
        };
        full_step(std::true_type{}, std::false_type{}, 0);
        int st = 1;
        for (; st + 1 < steps; st += 2) {
            full_step(std::false_type{}, std::true_type{}, st);
            full_step(std::false_type{}, std::false_type{}, st + 1);
        }
        if (st < steps) full_step(std::false_type{}, std::true_type{}, st);

        // ---- epilogue: y[pixel][channel], 4 channels per lane and accumulator quad ----------------------------------
        asm volatile("s_nop 15");
        __builtin_amdgcn_sched_barrier(0);
        const int nw0 = n0 + WROWS * wn;                  // the wave's first output channel
        const uint32_t prow = (uint32_t)(p0 + 128 * wm + r);    // + 16 tpx
        auto rsrc = [&](const float* ptr, int ld) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, (int)((uint32_t)P * ld * 4u), 0x00020000); };
        // operands of one accumulator row (8 quads) are loaded one row ahead of the arithmetic that uses them
        if constexpr (EPI == EPI_PLAIN) {
            // the wave's channel range lies in one of the two outputs (N0 is a multiple of it)
            const bool second_out = nw0 >= p.N0;
            const int ldy = second_out ? p.ldy1 : p.ldy;
            const bool accum = second_out ? p.acc1 != 0 : p.acc0 != 0;
            const __amdgpu_buffer_rsrc_t rs_y = rsrc(second_out ? o.y1 : o.y, ldy);
            const uint32_t ybase = (prow * ldy + (second_out ? nw0 - p.N0 : nw0) + 4 * q) * 4u;
            f32x4 old[2][8];
            auto fetch = [&](int tco, f32x4 (&o)[8]) {
#pragma unroll
                for (int tpx = 0; tpx < 8; ++tpx) o[tpx] = load4(rs_y, ybase + (uint32_t)(16 * tpx * ldy + 16 * tco) * 4u);
            };
            if (accum) fetch(0, old[0]);
#pragma unroll
            for (int tco = 0; tco < NTW; ++tco) {
                if (accum && tco + 1 < NTW) fetch(tco + 1, old[(tco + 1) & 1]);
                // (the accumulators of this row become visible to hipcc only here: see gemm_w128.h)
                asm volatile("" : "+a"(acc[tco][0]), "+a"(acc[tco][1]), "+a"(acc[tco][2]), "+a"(acc[tco][3]), "+a"(acc[tco][4]),
                             "+a"(acc[tco][5]), "+a"(acc[tco][6]), "+a"(acc[tco][7]));
#pragma unroll
                for (int tpx = 0; tpx < 8; ++tpx) {
                    f32x4 o = acc[tco][tpx];
                    if (accum) o += old[tco & 1][tpx];
                    store4(o, rs_y, ybase + (uint32_t)(16 * tpx * ldy + 16 * tco) * 4u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (EPI == EPI_GATES) {
            static_assert(EPI != EPI_GATES || NTW == 8 || NTW == 4, "z | r = 256 output channels: one tile, or a z tile and an r tile");
            const bool rwave = nw0 >= 128;                    // channels [128, 256) are r
            const __amdgpu_buffer_rsrc_t rs_add = rsrc(p.add, p.ld_add), rs_h = rsrc(p.h, p.ld_h);
            const __amdgpu_buffer_rsrc_t rs_o = rsrc(rwave ? o.y1 : o.y, rwave ? p.ldy1 : p.ldy), rs_r = rsrc(p.y2, p.ldy2);
            const int ldo = rwave ? p.ldy1 : p.ldy;
            const uint32_t c4 = (nw0 & 127) + 4 * q;          // + 16 tco: channel inside its half (z or r)
            f32x4 av[2][8], hv[2][8];
            auto fetch = [&](int tco, f32x4 (&a)[8], f32x4 (&hh)[8]) {
#pragma unroll
                for (int tpx = 0; tpx < 8; ++tpx) {
                    const uint32_t row = prow + 16 * tpx;
                    a[tpx] = load4(rs_add, (row * p.ld_add + nw0 + 16 * tco + 4 * q) * 4u);
                    if (rwave) hh[tpx] = load4(rs_h, (row * p.ld_h + 16 * tco + c4) * 4u);
                }
            };
            fetch(0, av[0], hv[0]);
#pragma unroll
            for (int tco = 0; tco < NTW; ++tco) {
                if (tco + 1 < NTW) fetch(tco + 1, av[(tco + 1) & 1], hv[(tco + 1) & 1]);
                asm volatile("" : "+a"(acc[tco][0]), "+a"(acc[tco][1]), "+a"(acc[tco][2]), "+a"(acc[tco][3]), "+a"(acc[tco][4]),
                             "+a"(acc[tco][5]), "+a"(acc[tco][6]), "+a"(acc[tco][7]));
#pragma unroll
                for (int tpx = 0; tpx < 8; ++tpx) {
                    const uint32_t row = prow + 16 * tpx;
                    const f32x4 pre = acc[tco][tpx] + av[tco & 1][tpx];
                    f32x4 g;
                    g[0] = sigmoid_(pre[0]); g[1] = sigmoid_(pre[1]); g[2] = sigmoid_(pre[2]); g[3] = sigmoid_(pre[3]);
                    if (rwave) {
                        store4(g, rs_r, (row * p.ldy2 + 16 * tco + c4) * 4u);
                        store4(g * hv[tco & 1][tpx], rs_o, (row * ldo + 16 * tco + c4) * 4u);
                    } else {
                        store4(g, rs_o, (row * ldo + 16 * tco + c4) * 4u);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            static_assert(EPI != EPI_BLEND || NTW == 4 || NTW == 2, "q = 128 output channels: one tile or two");
            const __amdgpu_buffer_rsrc_t rs_add = rsrc(p.add, p.ld_add), rs_h = rsrc(p.h, p.ld_h), rs_z = rsrc(p.z, p.ld_z);
            const __amdgpu_buffer_rsrc_t rs_o = rsrc(o.y, p.ldy), rs_q = rsrc(o.y1, p.ldy1);
            const uint32_t c4 = nw0 + 4 * q;                  // + 16 tco
            f32x4 av[2][8], hv[2][8], zv[2][8];
            auto fetch = [&](int tco, f32x4 (&a)[8], f32x4 (&hh)[8], f32x4 (&zz)[8]) {
#pragma unroll
                for (int tpx = 0; tpx < 8; ++tpx) {
                    const uint32_t row = prow + 16 * tpx;
                    a[tpx] = load4(rs_add, (row * p.ld_add + 16 * tco + c4) * 4u);
                    hh[tpx] = load4(rs_h, (row * p.ld_h + 16 * tco + c4) * 4u);
                    zz[tpx] = load4(rs_z, (row * p.ld_z + 16 * tco + c4) * 4u);
                }
            };
            fetch(0, av[0], hv[0], zv[0]);
#pragma unroll
            for (int tco = 0; tco < NTW; ++tco) {
                if (tco + 1 < NTW) fetch(tco + 1, av[(tco + 1) & 1], hv[(tco + 1) & 1], zv[(tco + 1) & 1]);
                asm volatile("" : "+a"(acc[tco][0]), "+a"(acc[tco][1]), "+a"(acc[tco][2]), "+a"(acc[tco][3]), "+a"(acc[tco][4]),
                             "+a"(acc[tco][5]), "+a"(acc[tco][6]), "+a"(acc[tco][7]));
#pragma unroll
                for (int tpx = 0; tpx < 8; ++tpx) {
                    const uint32_t row = prow + 16 * tpx;
                    const f32x4 pre = acc[tco][tpx] + av[tco & 1][tpx];
                    const f32x4 zq = zv[tco & 1][tpx], hq = hv[tco & 1][tpx];
                    f32x4 qv, hn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qv[e] = tanhf(pre[e]);
                        hn[e] = (1.0f - zq[e]) * hq[e] + zq[e] * qv[e];
                        if (p.sanitize) hn[e] = nan_to_num_(hn[e]);
                    }
                    store4(qv, rs_q, (row * p.ldy1 + 16 * tco + c4) * 4u);
                    store4(hn, rs_o, (row * p.ldy + 16 * tco + c4) * 4u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // before this workgroup's next tile reuses the LDS stages and the accumulators: everything has drained
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

template <int NTW, int NBUF, int EPI = EPI_PLAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convcl_kernel(Problem p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Operands o = {p.x, p.x1, p.w, p.y, p.y1, p.Cin, p.C0};
    convcl_body<NTW, NBUF, EPI>(p, o, lds);
}

}  // namespace ccl

// fp32 GEMM core for gfx950, wave tile 128 x 128: one wave per SIMD, 256 accumulator registers.
//
//     C[b][m][n] = alpha * sum_k A[b][k][m] * B[b][k][n]            A, B m- / n-contiguous ("k-major"), C row-major
//
// This is the contraction of the all-pairs volume build (models/raft_core.py:52-68: corr = f1^T f2 / sqrt(C)) and of the
// RAFT point cost volume (models/camliraft_l_core.py:51-60).  What differs from gemm_f32_mfma_kernel (allpairs.hip):
//
//  * 256 x 256 macro tile per 256-thread workgroup, 128 x 128 per wave = 8 x 8 tiles of v_mfma_f32_16x16x4_f32 whose 256
//    accumulators fill half of the unified 512-entry register file; ONE workgroup per CU.  Operand bytes per flop are half
//    of the 128 x 128 tile's, LDS fragment traffic a quarter.
//  * fragments by ds_read_b128 from the k-major LDS image the operands arrive in: lane (r = lane % 16, q = lane / 16)
//    reads four consecutive m of k row 4 s + q.  The four registers are the A fragments of FOUR 16-row tiles whose rows
//    are interleaved (tile j of a 64-row group = rows 4 r + j) -- a row permutation of the wave tile that costs nothing
//    and makes the epilogue's four values per lane consecutive in n: 16-byte stores.  A 16-lane group reads 64 consecutive
//    floats, i.e. every bank once.
//  * a PERSISTENT tile loop with one software pipeline across tiles: operands go direct to LDS
//    (global_load_lds_dwordx4, one instruction = one 256-float k row) NBUF - 1 steps ahead of the matrix cores, also
//    across the end of a tile, so the epilogue's stores drain while the next tile's first K steps already run.
//  * instruction-level interleave inside the wave: the fragment reads of sub-step s + 1 and the step's DMA issues are
//    spread between the MFMAs of sub-step s (sched_group_barrier), ONE workgroup barrier per K step, counted vmcnt.
//  * epilogue through a buffer descriptor: every wave issues exactly 64 buffer_store_dwordx4 per tile, rows / columns
//    beyond the matrix are dropped by the descriptor's range check (no branches -> the vmcnt arithmetic is exact).
//
// Numerics: every accumulator is the k-ascending fmaf chain of its products (v_mfma_f32_16x16x4_f32 adds k = 4 s .. 4 s + 3
// in order), bit-identical to the 32x32x2 kernel and to the oracle's loop.
//
// r6: the wave arrangement is a template parameter (GA x GB groups of 64 rows / columns per wave, WM x WN waves): the
// default <2, 2, 2> is the 256 x 256 tile above, instruction for instruction; <3, 1, 1> is a 192 x 256 tile (each wave all
// 192 rows x 64 columns, 192 accumulators) and <2, 1, 1> a 128 x 256 tile for the Winograd-domain contractions whose M is
// a convolution's output-channel count (winograd.hip).  The LDS image keeps its 256-float k rows whatever the tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace w128 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 256;          // macro tile edge
constexpr int GROUP_M = 4;       // tile rows per band of the tile order (consecutive tile ids = 4 rows x consecutive columns)

template <int V>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((V & 15) | (7 << 4) | (15 << 8) | ((V >> 4) << 14));
}
__device__ __forceinline__ void dma16(const float* src, float* lds_uniform) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}


// Half a row of the wave tile: 4 MFMAs acc[j] += a x b[j].  Inline assembly with the accumulators in the ACC half of the
// register file ("a" constraints): left to itself hipcc spreads 256 accumulators over both halves and shuttles them with
// v_accvgpr_read / _write inside the K loop (measured: 430 copies per step, 31 spilled registers).  The leading s_nop
// covers a VALU write of an operand right in front of the statement (hipcc pads nothing in front of an asm statement).
template <bool ZERO>
__device__ __forceinline__ void mfma_x4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float a, const f32x4& b) {
    if (ZERO) {         // first K = 4 of a tile: C = 0, the accumulators are (re)defined here and never zeroed separately
        asm volatile(
            "s_nop 1\n\t"
            "v_mfma_f32_16x16x4_f32 %0, %4, %5, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %1, %4, %6, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %4, %7, 0\n\t"
            "v_mfma_f32_16x16x4_f32 %3, %4, %8, 0"
            : "=a"(c0), "=a"(c1), "=a"(c2), "=a"(c3)
            : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
        return;
    }
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %1, %4, %6, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t"
        "v_mfma_f32_16x16x4_f32 %3, %4, %8, %3"
        : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
        : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

struct Problem {
    const float* A;      // [batch][K][M]   (lda = row stride in floats, sa = batch stride)
    const float* B;      // [batch][K][N]
    float* C;            // [batch][M][N]
    int M, N, K;
    int64_t lda, ldb, ldc, sa, sb, sc;
    float alpha;
    int tiles_m, tiles_n, tiles;      // tiles = batch * tiles_m * tiles_n
};

// tile id -> (batch, tile row, tile column): bands of GROUP_M tile rows, column-major inside a band
__device__ __forceinline__ void decode_tile(const Problem& p, int t, int& b, int& tm, int& tn) {
    const int per_batch = p.tiles_m * p.tiles_n;
    b = t / per_batch;
    const int u = t - b * per_batch;
    const int band = u / (GROUP_M * p.tiles_n);
    const int within = u - band * (GROUP_M * p.tiles_n);
    const int rows = min(GROUP_M, p.tiles_m - band * GROUP_M);
    tn = within / rows;
    tm = band * GROUP_M + (within - tn * rows);
}

// KS: K step per barrier (8 or 16); NBUF: LDS stages.  LDS: NBUF * KS * 512 floats.
// ABL (timing-only ablations, results wrong by design): 1 = no epilogue stores, 2 = no operand loads
// GA, GB: 64-row / 64-column groups per wave; WM: waves along M (4 / WM along N).  Macro tile 64 GA WM x 64 GB (4 / WM);
// the host sets Problem::tiles_m / tiles_n for THAT tile (tile_m<GA, WM>() / tile_n<GB, WM>()).
template <int GA, int WM> constexpr int tile_m() { return 64 * GA * WM; }
template <int GB, int WM> constexpr int tile_n() { return 64 * GB * (4 / WM); }

template <int KS, int NBUF, int ABL = 0, int GA = 2, int GB = 2, int WM = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w128_kernel(Problem p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert((KS == 8 || KS == 16) && NBUF >= 3, "one barrier per step needs three stages; SUB must be even");
    constexpr int WN = 4 / WM;
    constexpr int MTM = 64 * GA * WM, MTN = 64 * GB * WN;
    static_assert(WM * WN == 4 && MTM <= 256 && MTN <= 256, "four waves; the LDS image holds 256-float k rows");
    constexpr int NSLOT = 4 * GA * GB;        // slots of 4 MFMAs per sub-step
    constexpr int NFRAG = GA + GB;            // ds_read_b128 per sub-step
    constexpr int ESTORES = 16 * GA * GB;     // epilogue stores per wave and tile
    constexpr int SUB = KS / 4;               // sub-steps (one K = 4 MFMA row) per step
    constexpr int IPS = KS / 2;               // DMA instructions per wave and stage
    constexpr int DPS = (IPS + NSLOT - 3) / (NSLOT - 2);      // DMA instructions per slot of the last sub-step (1 on the 256 x 256 tile)
    constexpr int DSLOTS = (IPS + DPS - 1) / DPS;
    static_assert(NSLOT >= 8 && 2 * NFRAG <= NSLOT && DSLOTS + 2 <= NSLOT, "");
    constexpr int STAGE = KS * 512;           // floats: [KS][256] of A, then [KS][256] of B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, q = lane >> 4;
    const int steps = p.K / KS;

    // this workgroup's tile sequence: chunk c = round * 8 + xcd holds CPX consecutive tile ids (one XCD works on 4 tile
    // rows x CPX / 4 tile columns at a time: its L2 serves the shared operand panels)
    const int nwg = gridDim.x, cpx = nwg >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    auto tile_of = [&](int round) { return (round * 8 + xcd) * cpx + slot; };

    // ---- load side -------------------------------------------------------------------------------------------------
    int l_round = 0, l_step = 0;
    bool l_valid = tile_of(0) < p.tiles;
    const float *ga = p.A, *gb = p.B;
    auto load_setup = [&]() {
        int b, tm, tn;
        decode_tile(p, tile_of(l_round), b, tm, tn);
        ga = p.A + (int64_t)b * p.sa + (int64_t)wave * p.lda + min(tm * MTM + 4 * lane, p.M - 4);
        gb = p.B + (int64_t)b * p.sb + (int64_t)wave * p.ldb + min(tn * MTN + 4 * lane, p.N - 4);
        l_step = 0;
    };
    if (l_valid) load_setup();
    auto advance_loader = [&]() {
        ga += (int64_t)KS * p.lda;
        gb += (int64_t)KS * p.ldb;
        if (++l_step == steps) {
            ++l_round;
            l_valid = tile_of(l_round) < p.tiles;
            if (l_valid) load_setup();
        }
    };
    // stage `buf`: wave w brings k rows w, w + 4, ... of both operands
    auto issue_stage = [&](int buf) {
        float* sa_ = lds + buf * STAGE + wave * 256;
        float* sb_ = sa_ + KS * 256;
        if (ABL != 2) {
#pragma unroll
            for (int j = 0; j < KS / 4; ++j) dma16(ga + (int64_t)(4 * j) * p.lda, sa_ + 4 * j * 256);
#pragma unroll
            for (int j = 0; j < KS / 4; ++j) dma16(gb + (int64_t)(4 * j) * p.ldb, sb_ + 4 * j * 256);
        }
        advance_loader();
    };

    if (!l_valid) return;
    // ---- prologue: NBUF - 1 stages in flight -----------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (l_valid) issue_stage(i);
    wait_vm<IPS*(NBUF - 2)>();              // stage 0 landed (this wave's rows)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const float* fa = lds + q * 256 + 64 * GA * wm + 4 * r;              // fragment addresses inside a stage
    const float* fb = lds + KS * 256 + q * 256 + 64 * GB * wn + 4 * r;
    f32x4 acc[4 * GA][4 * GB];
    f32x4 pa[GA], pb[GB], qa[GA], qb[GB];
    // fragment WHICH of a sub-step: the wave's GA groups of A, then its GB groups of B
#define W128_FRAG_LOAD(BUF, S, WHICH)                                                                            \
    *reinterpret_cast<const f32x4*>(((WHICH) < GA ? fa + (WHICH) * 64 : fb + ((WHICH) - GA) * 64) + (BUF) * STAGE + (S) * 1024)
    auto frag_into = [&](f32x4 (&ya)[GA], f32x4 (&yb)[GB], int fbuf_, int s_, int which) {
        if (which < GA) ya[which] = W128_FRAG_LOAD(fbuf_, s_, which);
        else yb[which - GA] = W128_FRAG_LOAD(fbuf_, s_, which);
    };
    // one sub-step = NSLOT slots of 4 MFMAs (128 matrix-pipe cycles each); what `between(slot)` issues runs in the shadow
    // of the slot's last MFMA (32 cycles), so it should stay within a handful of instructions.  Everything is pinned: the
    // order written here is the order executed.
    auto substep = [&](auto zero, const f32x4 (&xa)[GA], const f32x4 (&xb)[GB], auto&& between) {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int ta = sl / GB, hb = sl % GB;
            mfma_x4<decltype(zero)::value>(acc[ta][4 * hb], acc[ta][4 * hb + 1], acc[ta][4 * hb + 2], acc[ta][4 * hb + 3], xa[ta >> 2][ta & 3], xb[hb]);
            __builtin_amdgcn_sched_barrier(0);
            between(sl);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the next sub-step's fragments: one read every second slot from slot 1 on (every slot on the 8-slot tiles)
    constexpr int FSTRIDE = 1 + 2 * (NFRAG - 1) < NSLOT ? 2 : 1;
    auto prefetch_into = [&](f32x4 (&ya)[GA], f32x4 (&yb)[GB], int fbuf_, int s_) {
        return [&ya, &yb, fbuf_, s_, &frag_into](int sl) {
            if (sl >= 1 && (sl - 1) % FSTRIDE == 0 && (sl - 1) / FSTRIDE < NFRAG) frag_into(ya, yb, fbuf_, s_, (sl - 1) / FSTRIDE);
        };
    };
    // slot of the last sub-step in which fragment i of the NEXT stage is read (behind the barrier of slot 0 and the DMAs)
    auto last_frag_slot = [](int i) constexpr {
        return NSLOT == 16 && NFRAG == 4 ? (i == 0 ? 9 : i == 1 ? 11 : i == 2 ? 13 : 14) : NSLOT - NFRAG + i;
    };

    int c_round = 0, buf = 0;
    bool after_epilogue = false;
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) frag_into(pa, pb, 0, 0, i);
    for (;;) {      // one tile per trip
        int cb, ctm, ctn;
        decode_tile(p, tile_of(c_round), cb, ctm, ctn);
        const bool more_tiles = tile_of(c_round + 1) < p.tiles;
        auto step = [&](auto first, int st) {
            // sub-steps 0 .. SUB - 2: prefetch the next sub-step's fragments, multiply the current one
#pragma unroll
            for (int s = 0; s + 1 < SUB; ++s) {
                if ((s & 1) == 0) {
                    if (s == 0) substep(first, pa, pb, prefetch_into(qa, qb, buf, s + 1));
                    else substep(std::false_type{}, pa, pb, prefetch_into(qa, qb, buf, s + 1));
                } else {
                    substep(std::false_type{}, qa, qb, prefetch_into(pa, pb, buf, s + 1));
                }
            }
            // last sub-step: the next stage must have landed for every wave before its first fragments are read, and
            // the stage every wave has finished reading is handed back to the loader
            const bool last = (st + 1 == steps) && !more_tiles;
            const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;
            const int fbuf = buf == 0 ? NBUF - 1 : buf - 1;          // stage st - 1's buffer: free after the barrier
            const bool issue = l_valid;
            float* const sa_ = lds + fbuf * STAGE + wave * 256;
            substep(std::false_type{}, qa, qb, [&](int sl) {
                if (sl == 0) {
                    // allowed in flight: the NBUF - 3 stages issued after the one needed now.  Right after an epilogue
                    // its ESTORES stores (returned in issue order with the loads) sit between the needed stage and anything
                    // younger: IPS + ESTORES (+ younger stages) operations were issued from that stage on, so "at most
                    // ESTORES (+ younger) in flight" retires it (62 = the counter's range on the 64-store tile).
                    // Once the loader has run dry the younger stages do not exist: wait for everything.
                    constexpr int AFTER_EPI = IPS * (NBUF - 3) + ESTORES < 62 ? IPS * (NBUF - 3) + ESTORES : 62;
                    if (!l_valid) wait_vm<0>();
                    else if (after_epilogue && st < NBUF - 2) wait_vm<AFTER_EPI>();
                    else wait_vm<IPS*(NBUF - 3)>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // DPS DMAs per slot: wave w brings k rows w, w + 4, ... of both operands
                if (sl >= 1 && sl <= DSLOTS && issue && ABL != 2) {
#pragma unroll
                    for (int j = (sl - 1) * DPS; j < sl * DPS && j < IPS; ++j) {
                        if (j < IPS / 2) dma16(ga + (int64_t)(4 * j) * p.lda, sa_ + 4 * j * 256);
                        else dma16(gb + (int64_t)(4 * (j - IPS / 2)) * p.ldb, sa_ + KS * 256 + 4 * (j - IPS / 2) * 256);
                    }
                }
                if (sl == DSLOTS + 1 && issue) advance_loader();
                if (!last) {
#pragma unroll
                    for (int i = 0; i < NFRAG; ++i)
                        if (sl == last_frag_slot(i)) frag_into(pa, pb, nbuf, 0, i);
                }
            });
            buf = nbuf;
        };
        step(std::true_type{}, 0);
        for (int st = 1; st < steps; ++st) step(std::false_type{}, st);
        // ---- epilogue: 64 x buffer_store_dwordx4 per wave, unconditional -------------------------------------------
        {
            asm volatile("s_nop 15");       // last MFMA's result -> first accumulator read (the asm MFMAs are opaque to hipcc)
            __builtin_amdgcn_sched_barrier(0);
            const int m0 = ctm * MTM + 64 * GA * wm, n0 = ctn * MTN + 64 * GB * wn;
            float* Cb = p.C + (int64_t)cb * p.sc;
            const uint32_t bytes = (uint32_t)p.M * (uint32_t)p.ldc * 4u;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cb, 0, bytes, 0x00020000);
            const uint32_t oob = 0x7FFFFFF0u;
            const int col = n0 + 4 * r;
            const uint32_t lane_off = ((uint32_t)(m0 + 16 * q) * (uint32_t)p.ldc + (uint32_t)col) * 4u;
            uint32_t off[GB];
#pragma unroll
            for (int hb = 0; hb < GB; ++hb) off[hb] = col + 64 * hb < p.N ? lane_off + 256u * hb : oob;
#pragma unroll
            for (int ta = 0; ta < 4 * GA; ++ta) {
                // the accumulators of this row become visible to hipcc only here (left alone it copies all 256 of them
                // into vector registers ahead of the first store: spills), are stored, and are zeroed for the next tile
#pragma unroll
                for (int hb = 0; hb < GB; ++hb)
                    asm volatile("" : "+a"(acc[ta][4 * hb]), "+a"(acc[ta][4 * hb + 1]), "+a"(acc[ta][4 * hb + 2]), "+a"(acc[ta][4 * hb + 3]));
                if (ABL != 1) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const uint32_t row_off = (uint32_t)(64 * (ta >> 2) + 4 * v + (ta & 3)) * (uint32_t)p.ldc * 4u;
#pragma unroll
                        for (int hb = 0; hb < GB; ++hb) {
                            f32x4 o = {acc[ta][4 * hb][v], acc[ta][4 * hb + 1][v], acc[ta][4 * hb + 2][v], acc[ta][4 * hb + 3][v]};
                            o *= p.alpha;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, off[hb] + row_off, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        after_epilogue = true;
        ++c_round;
        if (!more_tiles) break;
    }
#undef W128_FRAG_LOAD
}

}  // namespace w128

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

// KS: K step per barrier (16 or 32); NBUF: LDS stages.  LDS: NBUF * KS * 512 floats.
// ABL (timing-only ablations, results wrong by design): 1 = no epilogue stores, 2 = no operand loads
template <int KS, int NBUF, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w128_kernel(Problem p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert((KS == 8 || KS == 16) && NBUF >= 3, "one barrier per step needs three stages; SUB must be even");
    constexpr int SUB = KS / 4;               // sub-steps (one K = 4 MFMA row) per step
    constexpr int IPS = KS / 2;               // DMA instructions per wave and stage
    constexpr int STAGE = KS * 512;           // floats: [KS][256] of A, then [KS][256] of B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
        ga = p.A + (int64_t)b * p.sa + (int64_t)wave * p.lda + min(tm * MT + 4 * lane, p.M - 4);
        gb = p.B + (int64_t)b * p.sb + (int64_t)wave * p.ldb + min(tn * MT + 4 * lane, p.N - 4);
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

    const float* fa = lds + q * 256 + 128 * wm + 4 * r;                  // fragment addresses inside a stage
    const float* fb = lds + KS * 256 + q * 256 + 128 * wn + 4 * r;
    f32x4 acc[8][8];
    f32x4 pa[2], pb[2], qa[2], qb[2];
#define W128_FRAG(X, BUF, S, WHICH)                                                                              \
    X = *reinterpret_cast<const f32x4*>(((WHICH) < 2 ? fa : fb) + (BUF) * STAGE + (S) * 1024 + ((WHICH) & 1) * 64)
    // one sub-step = 16 slots of 4 MFMAs (128 matrix-pipe cycles); what `between(slot)` issues runs in the shadow of the
    // slot's last MFMA (32 cycles), so it should stay within a handful of instructions.  Everything is pinned: the order
    // written here is the order executed.
    auto substep = [&](auto zero, const f32x4 (&xa)[2], const f32x4 (&xb)[2], auto&& between) {
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) {
            const int ta = sl >> 1, hb = sl & 1;
            mfma_x4<decltype(zero)::value>(acc[ta][4 * hb], acc[ta][4 * hb + 1], acc[ta][4 * hb + 2], acc[ta][4 * hb + 3], xa[ta >> 2][ta & 3], xb[hb]);
            __builtin_amdgcn_sched_barrier(0);
            between(sl);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto prefetch_into = [&](f32x4 (&ya)[2], f32x4 (&yb)[2], int fbuf_, int s_, int first_row) {
        return [&ya, &yb, fbuf_, s_, first_row, fa, fb](int sl) {
            if (sl == first_row) W128_FRAG(ya[0], fbuf_, s_, 0);
            if (sl == first_row + 2) W128_FRAG(ya[1], fbuf_, s_, 1);
            if (sl == first_row + 4) W128_FRAG(yb[0], fbuf_, s_, 2);
            if (sl == first_row + 6) W128_FRAG(yb[1], fbuf_, s_, 3);
        };
    };

    int c_round = 0, buf = 0;
    bool after_epilogue = false;
    W128_FRAG(pa[0], 0, 0, 0); W128_FRAG(pa[1], 0, 0, 1); W128_FRAG(pb[0], 0, 0, 2); W128_FRAG(pb[1], 0, 0, 3);
    for (;;) {      // one tile per trip
        int cb, ctm, ctn;
        decode_tile(p, tile_of(c_round), cb, ctm, ctn);
        const bool more_tiles = tile_of(c_round + 1) < p.tiles;
        auto step = [&](auto first, int st) {
            // sub-steps 0 .. SUB - 2: prefetch the next sub-step's fragments, multiply the current one
#pragma unroll
            for (int s = 0; s + 1 < SUB; ++s) {
                if ((s & 1) == 0) {
                    if (s == 0) substep(first, pa, pb, prefetch_into(qa, qb, buf, s + 1, 1));
                    else substep(std::false_type{}, pa, pb, prefetch_into(qa, qb, buf, s + 1, 1));
                } else {
                    substep(std::false_type{}, qa, qb, prefetch_into(pa, pb, buf, s + 1, 1));
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
                    // its 64 stores (returned in issue order with the loads) sit between the needed stage and anything
                    // younger: >= 72 operations were issued from that stage on, so "at most 62 in flight" retires it.
                    // Once the loader has run dry the younger stages do not exist: wait for everything.
                    if (!l_valid) wait_vm<0>();
                    else if (after_epilogue && st < NBUF - 2) wait_vm<62>();
                    else wait_vm<IPS*(NBUF - 3)>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // one DMA per slot: wave w brings k rows w, w + 4, ... of both operands
                if (sl >= 1 && sl <= IPS && issue && ABL != 2) {
                    const int j = sl - 1;
                    if (j < IPS / 2) dma16(ga + (int64_t)(4 * j) * p.lda, sa_ + 4 * j * 256);
                    else dma16(gb + (int64_t)(4 * (j - IPS / 2)) * p.ldb, sa_ + KS * 256 + 4 * (j - IPS / 2) * 256);
                }
                if (sl == IPS + 1 && issue) advance_loader();
                if (!last) {
                    if (sl == 9) W128_FRAG(pa[0], nbuf, 0, 0);
                    if (sl == 11) W128_FRAG(pa[1], nbuf, 0, 1);
                    if (sl == 13) W128_FRAG(pb[0], nbuf, 0, 2);
                    if (sl == 14) W128_FRAG(pb[1], nbuf, 0, 3);
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
            const int m0 = ctm * MT + 128 * wm, n0 = ctn * MT + 128 * wn;
            float* Cb = p.C + (int64_t)cb * p.sc;
            const uint32_t bytes = (uint32_t)p.M * (uint32_t)p.ldc * 4u;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cb, 0, bytes, 0x00020000);
            const uint32_t oob = 0x7FFFFFF0u;
            const int col = n0 + 4 * r;
            const uint32_t lane_off = ((uint32_t)(m0 + 16 * q) * (uint32_t)p.ldc + (uint32_t)col) * 4u;
            const uint32_t off0 = col < p.N ? lane_off : oob;
            const uint32_t off1 = col + 64 < p.N ? lane_off + 256u : oob;
#pragma unroll
            for (int ta = 0; ta < 8; ++ta) {
                // the accumulators of this row become visible to hipcc only here (left alone it copies all 256 of them
                // into vector registers ahead of the first store: spills), are stored, and are zeroed for the next tile
                asm volatile("" : "+a"(acc[ta][0]), "+a"(acc[ta][1]), "+a"(acc[ta][2]), "+a"(acc[ta][3]), "+a"(acc[ta][4]),
                             "+a"(acc[ta][5]), "+a"(acc[ta][6]), "+a"(acc[ta][7]));
                if (ABL != 1) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const uint32_t row_off = (uint32_t)(64 * (ta >> 2) + 4 * v + (ta & 3)) * (uint32_t)p.ldc * 4u;
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb) {
                            f32x4 o = {acc[ta][4 * hb][v], acc[ta][4 * hb + 1][v], acc[ta][4 * hb + 2][v], acc[ta][4 * hb + 3][v]};
                            o *= p.alpha;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, (hb ? off1 : off0) + row_off, 0, 0);
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
#undef W128_FRAG
}

}  // namespace w128

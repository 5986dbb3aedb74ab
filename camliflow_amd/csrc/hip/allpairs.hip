// All-pairs cost-volume pyramid: build and adjoint on the fp32 matrix cores, gfx950.
//
// Replaces Correlation2D.build_cost_volume_pyramid of the reference (models/raft_core.py:52-68):
//     corr = matmul(f1^T, f2) / sqrt(C)          [B, P, P]      (P = h*w source / target pixels, C = 256)
//     pyramid = [corr, avg_pool2d(corr), avg_pool2d(avg_pool2d(corr)), ...]     over the TARGET dims
// i.e. one library GEMM, one elementwise pass for the scale and three pooling passes that re-read the 2.1 GB volume
// (batch 8, 68x120), and in the backward a volume-sized fold per level before two more GEMMs.
//
// avg_pool2d is linear and acts on the target pixel only, so level l of the pyramid IS a GEMM against the l-times
// pooled target features:   V_l = f1^T . pool_l(f2) / sqrt(C)    (equal to the reference up to fp32 summation order;
// the floor-cropping of odd sizes, 17x30 -> 8x15, is inherited from pool_l).  Every level is therefore produced by
// the same kernel straight from the feature maps, with the 1/sqrt(C) scale in the epilogue: each volume element is
// written exactly once and never re-read.  The adjoint needs no fold over volume-sized tensors either:
//     g_f1           = sum_l  pool_l(f2) . gV_l^T / sqrt(C)          [C, P]
//     g_pool_l(f2)   =        f1 . gV_l        / sqrt(C)             [C, P_l]     (un-pooled on the small maps)
//
// Kernel: 128x128 output tile per 256-thread workgroup, 2x2 tiles of v_mfma_f32_32x32x2_f32 per wave (exact fp32,
// a k-ordered fmaf chain), K walked in steps of 32 through a double-buffered LDS image that is always k-major
// ([k][m]: the MFMA fragment of a lane is then one ds_read_b32 and the 32 lanes of a half-wave read 32 consecutive
// floats, conflict-free); the next step's global loads are in flight in registers while the current one is on the
// matrix cores.  Operands may be m-contiguous ([K][M], staged with 16-byte loads / ds_write_b128) or k-contiguous
// ([M][K], transposed by the staging stores; row stride 130 floats keeps those stores on 32 distinct banks).
// The forward build (both operands m-contiguous, K = C) takes its own loop instead: K steps of 16 loaded DIRECT TO LDS
// (global_load_lds_dwordx4, no staging registers, two tiles in flight), see gemm_tile_loop_dma.  Edge tiles of aligned
// operands run the same unchecked loops as interior ones (clamped addresses, tile_load_one).
#include "camli_common.h"
#include "gemm_w128.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GB_T = 128;      // tile edge (M and N)
constexpr int GB_K = 32;       // K step

// One operand tile [GB_K][GB_T] goes from global memory into registers (4 float4 per thread, named scalars: an
// indexed member array here ends up in scratch memory) and on to LDS.
//   KC = false: element (k, m) at base[k * ld + m]   (m contiguous)
//   KC = true : element (k, m) at base[m * ld + k]   (k contiguous)
// CHECK = false: the tile is known to lie inside the operand and 16-byte loads are legal -- unconditional loads (a
// per-element "in range?" select makes the compiler branch around every load and wait for each one in turn).
// CHECK = true : out-of-range elements read as zero; `vec` = base and ld allow aligned 16-byte loads.
// KS = K step of the tile (32, or 16 for the unmarked loops: half the LDS image and two staging registers less per
// operand -> three workgroups per CU instead of two); a tile is KS * 32 float4, KS / 8 per thread.
template <bool KC, int KS = GB_K>
struct TileGeom {
    static constexpr int LD = KC ? 130 : 132;      // LDS row stride in floats
    static __device__ __forceinline__ int k_of(int f) { return KC ? 4 * (f & (KS / 4 - 1)) : (f >> 5); }
    static __device__ __forceinline__ int m_of(int f) { return KC ? (f / (KS / 4)) : 4 * (f & 31); }
};
template <bool KC>
using OperandTile = TileGeom<KC>;

template <bool KC, bool CHECK, int KS = GB_K>
__device__ __forceinline__ float4 tile_load_one(const float* __restrict__ base, int64_t ld, int m0, int k0, int M, int K,
                                                bool vec, int f) {
    const int gk = k0 + TileGeom<KC, KS>::k_of(f), gm = m0 + TileGeom<KC, KS>::m_of(f);
    if (!CHECK) {
        // Unconditional 16-byte loads for EVERY tile of an aligned operand, edge tiles included.  Rows / columns of the
        // tile beyond M only feed outputs that are never stored, so their addresses are clamped into the operand and
        // whatever they read is ignored; k beyond K must contribute zero: clamped address + a select, in the last step
        // only (block-uniform test).  Needs M % 4 == 0 (m contiguous) resp. K % 4 == 0 (k contiguous): a float4 is
        // inside or outside as a whole.
        const int gmc = min(gm, KC ? M - 1 : M - 4);
        int gkc = gk;
        const bool tail = k0 + KS > K;
        if (tail) gkc = min(gk, KC ? K - 4 : K - 1);
        const float* p = KC ? base + (int64_t)gmc * ld + gkc : base + (int64_t)gkc * ld + gmc;
        float4 v = *reinterpret_cast<const float4*>(p);
        if (tail && gk >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    }
    // CHECK = true: an operand whose rows are not 16-byte addressable (a level of 510 = 17 x 30 pixels, an odd channel
    // count).  Still no branches: four 4-byte loads at clamped addresses (every address lies inside the operand), k beyond
    // K reads as zero by a select, m beyond M feeds outputs that are never stored.  (The earlier form tested each element
    // and branched around its load: the compiler then waits for every load in turn -- 279 us for the 510-pixel level of
    // the forward build, 0.39 of the matrix rate where the aligned levels run at 0.67.)
    float4 v;
    if (KC) {
        const float* p = base + (int64_t)min(gm, M - 1) * ld;
        v.x = p[min(gk, K - 1)];
        v.y = p[min(gk + 1, K - 1)];
        v.z = p[min(gk + 2, K - 1)];
        v.w = p[min(gk + 3, K - 1)];
        if (gk >= K) v.x = 0.f;
        if (gk + 1 >= K) v.y = 0.f;
        if (gk + 2 >= K) v.z = 0.f;
        if (gk + 3 >= K) v.w = 0.f;
    } else {
        const float* p = base + (int64_t)min(gk, K - 1) * ld;
        v.x = p[min(gm, M - 1)];
        v.y = p[min(gm + 1, M - 1)];
        v.z = p[min(gm + 2, M - 1)];
        v.w = p[min(gm + 3, M - 1)];
        if (gk >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return v;
}

template <bool KC, int KS = GB_K>
__device__ __forceinline__ void tile_store_one(float* __restrict__ s, const float4& v, int f) {
    constexpr int LD = TileGeom<KC>::LD;
    float* d = s + TileGeom<KC, KS>::k_of(f) * LD + TileGeom<KC, KS>::m_of(f);
    if (KC) {
        d[0] = v.x; d[LD] = v.y; d[2 * LD] = v.z; d[3 * LD] = v.w;
    } else {
        *reinterpret_cast<float4*>(d) = v;
    }
}

#define GB_LOAD4(KCF, R, base, ld, x0, k0, X, K, vec)                                        \
    R##0 = tile_load_one<KCF, CHECK>(base, ld, x0, k0, X, K, vec, tid);                     \
    R##1 = tile_load_one<KCF, CHECK>(base, ld, x0, k0, X, K, vec, tid + 256);               \
    R##2 = tile_load_one<KCF, CHECK>(base, ld, x0, k0, X, K, vec, tid + 512);               \
    R##3 = tile_load_one<KCF, CHECK>(base, ld, x0, k0, X, K, vec, tid + 768)
#define GB_STORE4(KCF, R, s)                                                                 \
    tile_store_one<KCF>(s, R##0, tid);                                                       \
    tile_store_one<KCF>(s, R##1, tid + 256);                                                 \
    tile_store_one<KCF>(s, R##2, tid + 512);                                                 \
    tile_store_one<KCF>(s, R##3, tid + 768)
// the same for a K step of KS (16: registers 2 and 3 stay unused and are dropped by the compiler)
#define GB_LOADK(KCF, R, base, ld, x0, k0, X, K, vec)                                        \
    R##0 = tile_load_one<KCF, CHECK, KS>(base, ld, x0, k0, X, K, vec, tid);                 \
    R##1 = tile_load_one<KCF, CHECK, KS>(base, ld, x0, k0, X, K, vec, tid + 256);           \
    if (KS == 32) {                                                                          \
        R##2 = tile_load_one<KCF, CHECK, KS>(base, ld, x0, k0, X, K, vec, tid + 512);       \
        R##3 = tile_load_one<KCF, CHECK, KS>(base, ld, x0, k0, X, K, vec, tid + 768);       \
    }
#define GB_STOREK(KCF, R, s)                                                                 \
    tile_store_one<KCF, KS>(s, R##0, tid);                                                   \
    tile_store_one<KCF, KS>(s, R##1, tid + 256);                                             \
    if (KS == 32) {                                                                          \
        tile_store_one<KCF, KS>(s, R##2, tid + 512);                                         \
        tile_store_one<KCF, KS>(s, R##3, tid + 768);                                         \
    }

// The K loop of one 128x128 tile.  CHECK = false: interior tile, aligned operands, K a multiple of GB_K -- the loop
// contains no bounds logic at all (a checked load anywhere in the loop makes the compiler wait for the whole
// prefetch before the matrix-core section, which serialises load latency and MFMA time).
// SKIPZ: the B operand is the accumulated lookup gradient of the volume, which is zero outside the windows the GRU
// iterations visited (a band around the flow field: ~15-20 % of the 128x32 tiles).  The staging pass already holds a
// tile in registers, so "is any element non-zero" is one OR per thread folded into the step's barrier
// (__syncthreads_or), and a zero tile skips its 64 MFMAs.  Values are unchanged (adding exact zeros).
template <bool A_KC, bool B_KC, bool CHECK, bool SKIPZ, int KS>
__device__ __forceinline__ void gemm_tile_loop(const float* __restrict__ Ab, const float* __restrict__ Bb, float* lds,
                                               f32x16 (&acc)[2][2], int M, int N, int K, int64_t lda, int64_t ldb, int m0,
                                               int n0, bool vec_a, bool vec_b, int tid, int wm, int wn, int k_begin = 0,
                                               int k_end = 1 << 30) {
    constexpr int LDA = OperandTile<A_KC>::LD, LDB = OperandTile<B_KC>::LD;
    k_end = min(k_end, K);                          // [k_begin, k_end): this workgroup's share of K (split-K adjoint)
    if (k_begin >= k_end) return;
    float* const sA = lds;                          // two buffers of [GB_K][LDA]
    float* const sB = lds + 2 * KS * LDA;           // two buffers of [KS][LDB]
    const int lane = tid & 63;
    const int fk = lane >> 5, fm = lane & 31;
    float4 ra0, ra1, ra2 = make_float4(0.f, 0.f, 0.f, 0.f), ra3 = ra2, rb0, rb1, rb2 = ra2, rb3 = ra2;
    GB_LOADK(A_KC, ra, Ab, lda, m0, k_begin, M, K, vec_a);
    GB_LOADK(B_KC, rb, Bb, ldb, n0, k_begin, N, K, vec_b);
    GB_STOREK(A_KC, ra, sA);
    GB_STOREK(B_KC, rb, sB);
#define GB_NONZERO(R) (((R##0).x != 0.f) | ((R##0).y != 0.f) | ((R##0).z != 0.f) | ((R##0).w != 0.f) | ((R##1).x != 0.f) | ((R##1).y != 0.f) | \
                       ((R##1).z != 0.f) | ((R##1).w != 0.f) | ((R##2).x != 0.f) | ((R##2).y != 0.f) | ((R##2).z != 0.f) | ((R##2).w != 0.f) | \
                       ((R##3).x != 0.f) | ((R##3).y != 0.f) | ((R##3).z != 0.f) | ((R##3).w != 0.f))
    int live = 1;
    if (SKIPZ) live = __syncthreads_or(GB_NONZERO(rb) ? 1 : 0);
    else __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += KS) {
        const bool more = k0 + KS < k_end;
        if (more) {     // next step's operands: in flight while this step runs on the matrix cores
            GB_LOADK(A_KC, ra, Ab, lda, m0, k0 + KS, M, K, vec_a);
            GB_LOADK(B_KC, rb, Bb, ldb, n0, k0 + KS, N, K, vec_b);
        }
        const float* a = sA + buf * (KS * LDA) + fk * LDA + wm + fm;
        const float* b = sB + buf * (KS * LDB) + fk * LDB + wn + fm;
        // the fragments of a half step (8 k-pairs x 4 values) are read from LDS in one batch, then 32 MFMAs run
        // back to back: the matrix pipe never waits on an LDS round trip between its own instructions
        if (!SKIPZ || live)
#pragma unroll
        for (int kh = 0; kh < KS; kh += 16) {
            float fa0[8], fa1[8], fb0[8], fb1[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                fa0[t] = a[(kh + 2 * t) * LDA]; fa1[t] = a[(kh + 2 * t) * LDA + 32];
                fb0[t] = b[(kh + 2 * t) * LDB]; fb1[t] = b[(kh + 2 * t) * LDB + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
            GB_STOREK(A_KC, ra, sA + (buf ^ 1) * (KS * LDA));     // the other buffer: its readers finished one barrier ago
            GB_STOREK(B_KC, rb, sB + (buf ^ 1) * (KS * LDB));
        }
        if (SKIPZ) live = __syncthreads_or((more && GB_NONZERO(rb)) ? 1 : 0);
        else __syncthreads();
        buf ^= 1;
    }
}

// Fragment reads / MFMA batches of half a K step (NP k pairs) of the direct-to-LDS loop below.
template <int NP, int LDA, int LDB>
__device__ __forceinline__ void frag_read(const float* a, const float* b, float (&fa0)[NP], float (&fa1)[NP],
                                          float (&fb0)[NP], float (&fb1)[NP]) {
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        fa0[t] = a[2 * t * LDA]; fa1[t] = a[2 * t * LDA + 32];
        fb0[t] = b[2 * t * LDB]; fb1[t] = b[2 * t * LDB + 32];
    }
}
template <int NP>
__device__ __forceinline__ void frag_mfma(f32x16 (&acc)[2][2], const float (&fa0)[NP], const float (&fa1)[NP],
                                          const float (&fb0)[NP], const float (&fb1)[NP]) {
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb1[t], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb0[t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc[1][1], 0, 0, 0);
    }
}

// The forward loop with DIRECT-TO-LDS operand loads (global_load_lds_dwordx4: 64 lanes x 16 bytes land at M0 + lane * 16,
// i.e. one instruction fills two 128-float k rows of a tile).  No staging registers, no ds_write, no
// "loads back -> LDS stores done" chain in front of the barrier; and since nothing is held in registers the loads of
// tile t + 2 are issued as soon as the barrier of step t has freed the buffer tile t was read from -- a full step of
// flight time out of two buffers.  LDS rows are unpadded (128 floats): the fragment reads take the two half-waves of a
// ds_read in separate passes, so row k and row k + 1 on the same banks do not collide.  Both operands m-contiguous,
// 16-byte aligned, K a multiple of 16 (nothing can zero-fill a k tail here); M / N edges by clamped addresses.
__device__ __forceinline__ void lds_dma16(const float* src, float* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}
template <int NBUF>
__device__ __forceinline__ void gemm_tile_loop_dma(const float* __restrict__ Ab, const float* __restrict__ Bb, float* lds,
                                                   f32x16 (&acc)[2][2], int M, int N, int K, int64_t lda, int64_t ldb, int m0,
                                                   int n0, int tid, int wm, int wn) {
    constexpr int KS = 16, H = 8, NP = 4, LD = 128;
    float* const sA = lds;                          // NBUF buffers of [16][128]
    float* const sB = lds + NBUF * KS * LD;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fk = lane >> 5, fm = lane & 31;
    // this lane's source column and first k row; instruction j of a step covers k rows 4 * wave + 2 * j + {0, 1}
    const float* ga = Ab + (int64_t)(4 * wave + fk) * lda + min(m0 + 4 * fm, M - 4);
    const float* gb = Bb + (int64_t)(4 * wave + fk) * ldb + min(n0 + 4 * fm, N - 4);
    float* const da = sA + (4 * wave) * LD;         // wave-uniform LDS destinations (buffer 0)
    float* const db = sB + (4 * wave) * LD;
    auto issue = [&](int step, int buf) {
        const float* pa = ga + (int64_t)step * KS * lda;
        const float* pb = gb + (int64_t)step * KS * ldb;
        lds_dma16(pa, da + buf * (KS * LD));
        lds_dma16(pa + 2 * lda, da + buf * (KS * LD) + 2 * LD);
        lds_dma16(pb, db + buf * (KS * LD));
        lds_dma16(pb + 2 * ldb, db + buf * (KS * LD) + 2 * LD);
    };
    float pa0[NP], pa1[NP], pb0[NP], pb1[NP];
    float qa0[NP], qa1[NP], qb0[NP], qb1[NP];
    const int steps = K / KS;
    // NBUF - 1 steps in flight beside the one on the matrix cores (the loads of a wave return in order: "all but the
    // youngest 4 * (NBUF - 2)" = the next step has landed)
    issue(0, 0);
#pragma unroll
    for (int i = 1; i < NBUF - 1; ++i)
        if (i < steps) issue(i, i);
    if (NBUF == 3 && steps > 1) __builtin_amdgcn_s_waitcnt(0x0F74); else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (NBUF - 1 < steps) issue(NBUF - 1, NBUF - 1);
    const float* a = sA + fk * LD + wm + fm;
    const float* b = sB + fk * LD + wn + fm;
    frag_read<NP, LD, LD>(a, b, pa0, pa1, pb0, pb1);
    int buf = 0;
    // all steps but the last: no conditions inside (a guarded fragment read makes the compiler wait for every LDS
    // read in front of the next MFMA batch)
    for (int t = 0; t + 1 < steps; ++t) {
        frag_read<NP, LD, LD>(a + (buf * KS + H) * LD, b + (buf * KS + H) * LD, qa0, qa1, qb0, qb1);
        __builtin_amdgcn_sched_barrier(0);
        frag_mfma<NP>(acc, pa0, pa1, pb0, pb1);
        __builtin_amdgcn_sched_barrier(0);
        // step t + 1 has landed (this wave's share; the barrier covers the rest)
        if (NBUF == 3 && t + 2 < steps) __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4): step t + 2 may still fly
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0)
        __syncthreads();
        if (t + NBUF < steps) issue(t + NBUF, buf);       // the buffer every wave has just finished reading
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        frag_read<NP, LD, LD>(a + buf * KS * LD, b + buf * KS * LD, pa0, pa1, pb0, pb1);
        __builtin_amdgcn_sched_barrier(0);
        frag_mfma<NP>(acc, qa0, qa1, qb0, qb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    frag_read<NP, LD, LD>(a + (buf * KS + H) * LD, b + (buf * KS + H) * LD, qa0, qa1, qb0, qb1);
    __builtin_amdgcn_sched_barrier(0);
    frag_mfma<NP>(acc, pa0, pa1, pb0, pb1);
    frag_mfma<NP>(acc, qa0, qa1, qb0, qb1);
}

// The same K loop driven by visit marks instead of tests on loaded data.  The lookup adjoint records which 32x32
// blocks (32 source pixels x 32 target pixels) of a gradient level it ever wrote (camli_allpairs_lookup_bwd_marked);
// a K step whose B tile holds no marked block is never loaded, staged or multiplied.  ~80 % of the level-0 tiles are
// untouched, and with the data-driven skip each of them still cost one exposed load latency (only one step is
// prefetched ahead) plus its 16 KB of HBM traffic.  mode 1: B = gV [N = source][K = target] (tile rows n0.. are 4
// source blocks, step s is target block s); mode 2: B = gV [K = source][N = target] (step s is source block s, the
// tile's columns n0.. are 4 target blocks).  marks: this sample's [src_blocks][tb] bytes.  K <= 8192 (256 steps).
template <bool A_KC, bool B_KC, bool CHECK>
__device__ __forceinline__ void gemm_tile_loop_marked(const float* __restrict__ Ab, const float* __restrict__ Bb, float* lds,
                                                      f32x16 (&acc)[2][2], int M, int N, int K, int64_t lda, int64_t ldb,
                                                      int m0, int n0, bool vec_a, bool vec_b, int tid, int wm, int wn,
                                                      const unsigned char* __restrict__ marks, int mode, int tb, int src_blocks,
                                                      int ks_begin = 0, int ks_end = 1 << 24) {
    constexpr int LDA = OperandTile<A_KC>::LD, LDB = OperandTile<B_KC>::LD;
    __shared__ unsigned long long s_live[4];        // step s has a marked block (bit s & 63 of word s >> 6)
    float* const sA = lds;
    float* const sB = lds + 2 * GB_K * LDA;
    const int lane = tid & 63, wave = tid >> 6;
    const int fk = lane >> 5, fm = lane & 31;
    {
        const int steps = (K + GB_K - 1) / GB_K;
        const int nb = n0 >> 5;
        bool live = false;
        if (tid < steps && tid >= ks_begin && tid < ks_end) {      // [ks_begin, ks_end): this workgroup's share of the steps
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (mode == 1) { if (nb + i < src_blocks) live |= marks[(size_t)(nb + i) * tb + tid] != 0; }
                else           { if (nb + i < tb) live |= marks[(size_t)tid * tb + nb + i] != 0; }
            }
        }
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_live[wave] = m;
        __syncthreads();
    }
    // smallest live step > after, or -1 (block-uniform)
    auto next_live = [&](int after) -> int {
        const int from = after + 1;
        int found = -1;
#pragma unroll
        for (int w = 3; w >= 0; --w) {
            unsigned long long m = s_live[w];
            if (w == (from >> 6)) m &= ~0ull << (from & 63);
            if (w < (from >> 6)) m = 0;
            if (m) found = w * 64 + __builtin_ctzll(m);
        }
        return __builtin_amdgcn_readfirstlane(found);
    };
    int cur = next_live(-1);
    if (cur < 0) return;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    GB_LOAD4(A_KC, ra, Ab, lda, m0, cur * GB_K, M, K, vec_a);
    GB_LOAD4(B_KC, rb, Bb, ldb, n0, cur * GB_K, N, K, vec_b);
    GB_STORE4(A_KC, ra, sA);
    GB_STORE4(B_KC, rb, sB);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int nxt = next_live(cur);
        if (nxt >= 0) {
            GB_LOAD4(A_KC, ra, Ab, lda, m0, nxt * GB_K, M, K, vec_a);
            GB_LOAD4(B_KC, rb, Bb, ldb, n0, nxt * GB_K, N, K, vec_b);
        }
        const float* a = sA + buf * (GB_K * LDA) + fk * LDA + wm + fm;
        const float* b = sB + buf * (GB_K * LDB) + fk * LDB + wn + fm;
        // (leaving out the MFMAs of a 32-column block whose gradient block is unmarked was measured: no gain -- a live
        // step is bound by its loads, LDS stores and barrier, not by the matrix pipe)
#pragma unroll
        for (int kh = 0; kh < GB_K; kh += 16) {
            float fa0[8], fa1[8], fb0[8], fb1[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                fa0[t] = a[(kh + 2 * t) * LDA]; fa1[t] = a[(kh + 2 * t) * LDA + 32];
                fb0[t] = b[(kh + 2 * t) * LDB]; fb1[t] = b[(kh + 2 * t) * LDB + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc[1][1], 0, 0, 0);
            }
        }
        if (nxt >= 0) {
            GB_STORE4(A_KC, ra, sA + (buf ^ 1) * (GB_K * LDA));
            GB_STORE4(B_KC, rb, sB + (buf ^ 1) * (GB_K * LDB));
        }
        __syncthreads();
        if (nxt < 0) break;
        buf ^= 1;
        cur = nxt;
    }
}

// One wave's 64 x 64 share of a 128 x 128 output tile: C = alpha * acc (+ C when ACC).
template <bool ACC>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[2][2], float* __restrict__ Cb, int M, int N, int64_t ldc, float alpha,
                                           int m0, int n0, int wm, int wn, int lane) {
    const int fk = lane >> 5, fm = lane & 31;
    // C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (m0 + GB_T <= M && n0 + GB_T <= N) {
        // interior tile (block-uniform): every row address is a wave-uniform 64-bit pointer (scalar arithmetic) plus
        // one per-lane 32-bit offset computed once -- no per-store 64-bit multiply, compare or exec masking as in the
        // checked form below (fewer instructions; the kernel time did not move with it: the tail is bound by the
        // writes themselves)
        const int uwm = __builtin_amdgcn_readfirstlane(wm), uwn = __builtin_amdgcn_readfirstlane(wn);
        const int lane_off = 4 * fk * (int)ldc + fm;            // ldc is a pixel count < 2^28 (build_args_ok)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* urow = Cb + (int64_t)(m0 + uwm + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + (n0 + uwn + j * 32);
                    const float v = alpha * acc[i][j][r];
                    urow[lane_off] = ACC ? (urow[lane_off] + v) : v;
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + j * 32 + fm;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < M && col < N) {
                    float* c = Cb + (int64_t)row * ldc + col;
                    const float v = alpha * acc[i][j][r];
                    *c = ACC ? (*c + v) : v;
                }
            }
        }
}

// C[b][m][n] (row-major, ldc)  =  alpha * sum_k A(b; m, k) * B(b; k, n)   (+ C when ACC)
// grid (ceil(N/128), ceil(M/128), batch), block 256
// One 128x128 output tile (m0, n0) of one batch entry: Ab / Bb / Cb / mk already point at that entry.
template <bool A_KC, bool B_KC, bool ACC, bool SKIPZ, int KS = GB_K, int DMA = 0>
__device__ __forceinline__ void gemm_block(const float* __restrict__ Ab, const float* __restrict__ Bb, float* __restrict__ Cb,
                                           int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, float alpha, int vec_a,
                                           int vec_b, const unsigned char* __restrict__ mk, int mark_mode, int mark_tb,
                                           int mark_src_blocks, int m0, int n0, float* lds, int ks_begin = 0,
                                           int ks_end = 1 << 24) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // block-uniform: aligned operands take the unchecked loops for every tile (see tile_load_one)
    const bool fast = vec_a && vec_b && (A_KC ? (K % 4 == 0 && K >= 4) : (M % 4 == 0 && M >= 4)) &&
                      (B_KC ? (K % 4 == 0 && K >= 4) : (N % 4 == 0 && N >= 4));
    if (SKIPZ && mk) {
        if (fast)
            gemm_tile_loop_marked<A_KC, B_KC, false>(Ab, Bb, lds, acc, M, N, K, lda, ldb, m0, n0, true, true, tid, wm, wn, mk,
                                                     mark_mode, mark_tb, mark_src_blocks, ks_begin, ks_end);
        else
            gemm_tile_loop_marked<A_KC, B_KC, true>(Ab, Bb, lds, acc, M, N, K, lda, ldb, m0, n0, vec_a != 0, vec_b != 0, tid, wm,
                                                    wn, mk, mark_mode, mark_tb, mark_src_blocks, ks_begin, ks_end);
    } else if (fast) {
        if constexpr (DMA != 0 && !SKIPZ && !A_KC && !B_KC)
            gemm_tile_loop_dma<DMA>(Ab, Bb, lds, acc, M, N, K, lda, ldb, m0, n0, tid, wm, wn);      // launch_gemm checks K % 16 == 0
        else
            gemm_tile_loop<A_KC, B_KC, false, SKIPZ, KS>(Ab, Bb, lds, acc, M, N, K, lda, ldb, m0, n0, true, true, tid, wm, wn,
                                                         ks_begin * GB_K, min(ks_end, 1 << 24) * GB_K);
    } else
        gemm_tile_loop<A_KC, B_KC, true, SKIPZ, KS>(Ab, Bb, lds, acc, M, N, K, lda, ldb, m0, n0, vec_a != 0, vec_b != 0, tid, wm, wn,
                                                    ks_begin * GB_K, min(ks_end, 1 << 24) * GB_K);

    store_tile<ACC>(acc, Cb, M, N, ldc, alpha, m0, n0, wm, wn, lane);
}

template <bool A_KC, bool B_KC, bool ACC, bool SKIPZ, int KS = GB_K>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                             float* __restrict__ C, int M, int N, int K, int64_t lda,
                                                             int64_t ldb, int64_t ldc, int64_t sa, int64_t sb, int64_t sc,
                                                             float alpha, int vec_a, int vec_b,
                                                             const unsigned char* __restrict__ marks, int mark_mode,
                                                             int mark_tb, int mark_src_blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // 2 * GB_K * (LDA + LDB) floats (> 64 KB: dynamic)
    const unsigned char* mk = marks ? marks + (size_t)blockIdx.z * mark_src_blocks * mark_tb : nullptr;
    gemm_block<A_KC, B_KC, ACC, SKIPZ, KS>(A + (int64_t)blockIdx.z * sa, Bm + (int64_t)blockIdx.z * sb, C + (int64_t)blockIdx.z * sc,
                                       M, N, K, lda, ldb, ldc, alpha, vec_a, vec_b, mk, mark_mode, mark_tb, mark_src_blocks,
                                       blockIdx.y * GB_T, blockIdx.x * GB_T, lds);
}

// The forward build's kernel: m-contiguous aligned operands, K % 16 == 0, direct-to-LDS loop (32 KB of LDS, <= 168
// registers: three workgroups per CU).  Measured on top of it and not kept, all within 0.5 % of it (0.67 of the fp32 MFMA
// peak; profiles/r04_gemm_epilogue_experiments.txt): a fourth wave per SIMD, three LDS buffers (two steps in flight), a
// 256 x 128 tile per workgroup (128 x 64 per wave: -25 % operand loads and fragment reads per MFMA), and a persistent form
// (768 workgroups walking the tiles, K steps of consecutive tiles in one pipeline).
template <bool ACC, int NBUF = 2>
__global__ __launch_bounds__(256) void gemm_fwd_dma_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                            float* __restrict__ C, int M, int N, int K, int64_t lda, int64_t ldb,
                                                            int64_t ldc, int64_t sa, int64_t sb, int64_t sc, float alpha) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // 2 x 16 x (128 + 128) floats
    gemm_block<false, false, ACC, false, 16, NBUF>(A + (int64_t)blockIdx.z * sa, Bm + (int64_t)blockIdx.z * sb,
                                                   C + (int64_t)blockIdx.z * sc, M, N, K, lda, ldb, ldc, alpha, 1, 1, nullptr, 0, 0,
                                                   0, blockIdx.y * GB_T, blockIdx.x * GB_T, lds);
}

// The g_f2_l GEMMs of ALL pyramid levels in one launch.  Per level they are M = C (2 row tiles), N = P_l, K = P: the
// coarse levels have 16 / 4 / 1 column tiles, i.e. 256 / 64 / 16 workgroups walking a 255-step K loop -- launched one
// after the other they leave the chip mostly idle for ~1.5 ms.  grid.x enumerates the column tiles of the levels, coarse
// levels FIRST (their long loops start at once, the level-0 tiles fill the remaining CUs).  A = f1 is shared.
// Split K: the coarse levels are few tiles with the longest and densest loops (measured on a training step: 40 live steps of
// 255 per level-0 tile, 83 / 170 / 255 on levels 1 / 2 / 3), i.e. the launch lasted as long as ONE level-3 workgroup walking
// 255 steps behind its barriers.  A group is therefore a (level, range of K steps) pair writing its own partial result;
// reduce_parts_kernel adds the parts of a level in a fixed order (deterministic).
constexpr int GG_MAX = 48;
struct GemmGroup {
    const float* B[GG_MAX];
    float* C[GG_MAX];
    const unsigned char* marks[GG_MAX];
    int N[GG_MAX];
    int ks_begin[GG_MAX], ks_end[GG_MAX];      // K steps (of 32) of this group
    int tile_begin[GG_MAX + 1];        // in launch order
    int groups;
};
__global__ __launch_bounds__(256) void gemm_gf2_grouped_kernel(const float* __restrict__ A, GemmGroup g, int M, int K,
                                                                int64_t lda, int64_t sa, float alpha, int vec_a, int vec_ok,
                                                                int mark_src_blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int l = 0;
    while (l + 1 < g.groups && (int)blockIdx.x >= g.tile_begin[l + 1]) ++l;
    const int N = g.N[l];
    const int tb = (N + 31) / 32;
    const int n0 = ((int)blockIdx.x - g.tile_begin[l]) * GB_T;
    const unsigned char* mk = g.marks[l] ? g.marks[l] + (size_t)blockIdx.z * mark_src_blocks * tb : nullptr;
    const int vec_b = vec_ok && (N % 4 == 0);
    gemm_block<true, false, false, true>(A + (int64_t)blockIdx.z * sa, g.B[l] + (int64_t)blockIdx.z * K * N,
                                         g.C[l] + (int64_t)blockIdx.z * M * N, M, N, K, lda, N, N, alpha, vec_a, vec_b, mk, 2, tb,
                                         mark_src_blocks, blockIdx.y * GB_T, n0, lds, g.ks_begin[l], g.ks_end[l]);
}

// out[i] = part_0[i] + part_1[i] + ... (left to right) for every split level; grid (blocks, levels)
constexpr int RG_MAX = 8;
struct ReduceGroup {
    float* out[RG_MAX];
    const float* part[RG_MAX];         // parts of a level lie n floats apart
    int parts[RG_MAX];
    int64_t n[RG_MAX];
};
__global__ __launch_bounds__(256) void reduce_parts_kernel(ReduceGroup g) {
    const int l = blockIdx.y;
    const int64_t n = g.n[l];
    const float* p = g.part[l];
    float* out = g.out[l];
    const int parts = g.parts[l];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float acc = p[i];
        for (int s = 1; s < parts; ++s) acc += p[(int64_t)s * n + i];
        out[i] = acc;
    }
}

// parts of level l in the split-K adjoint: 1, 2, 4, 8, 8, ... with at least 16 K steps each
int gf2_parts(int level, int steps) {
    int parts = level >= 3 ? 8 : (1 << level);
    while (parts > 1 && steps / parts < 16) parts >>= 1;
    return parts;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// The persistent 256 x 256-tile kernel of gemm_w128.h.  Returns false (nothing launched) when the shape is outside its range:
// operands must be 16-byte addressable row by row (direct-to-LDS loads of 4 floats), K a multiple of 16 with at least three
// steps (the pipeline's depth), the output slab of one batch entry below 2 GB (32-bit offsets behind a buffer descriptor), and
// N wide enough that a 256-column tile is not mostly padding.  CAMLI_GEMM_W128=0 keeps every GEMM on the older kernels (A/B).
bool launch_w128(const float* A, const float* Bm, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                 int64_t sa, int64_t sb, int64_t sc, int batch, float alpha, bool vec, hipStream_t stream) {
    constexpr int KS = 16, NBUF = 3;
    static const bool enabled = []() { const char* e = getenv("CAMLI_GEMM_W128"); return !e || atoi(e) != 0; }();
    if (!enabled || !vec || M % 4 || N % 4 || M < 4 || N < 192 || K % KS || K < NBUF * KS) return false;
    if (ldc % 4 || sc % 4 || !aligned16(C) || (int64_t)(M + 256) * ldc * 4 >= (int64_t)0x7FF00000) return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        cus = n >= 8 ? n / 8 * 8 : 8;
    }
    constexpr size_t lds = (size_t)NBUF * KS * 512 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&w128::gemm_w128_kernel<KS, NBUF, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        attr_set = true;
    }
    w128::Problem p;
    p.A = A; p.B = Bm; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sa = sa; p.sb = sb; p.sc = sc;
    p.alpha = alpha;
    p.tiles_m = camli_divup(M, w128::MT);
    p.tiles_n = camli_divup(N, w128::MT);
    const int64_t tiles = (int64_t)batch * p.tiles_m * p.tiles_n;
    if (tiles >= (int64_t)1 << 30) return false;
    p.tiles = (int)tiles;
    hipLaunchKernelGGL((w128::gemm_w128_kernel<KS, NBUF, 0>), dim3(cus), dim3(256), lds, stream, p);
    return true;
}

template <bool A_KC, bool B_KC, bool SKIPZ>
void launch_gemm(const float* A, const float* Bm, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                 int64_t sa, int64_t sb, int64_t sc, int batch, float alpha, bool accumulate, hipStream_t stream,
                 const unsigned char* marks = nullptr, int mark_mode = 0, int mark_tb = 0, int mark_src_blocks = 0) {
    if (K > 256 * GB_K) marks = nullptr;      // the live-step masks hold 256 steps; longer loops test the data instead
    const int vec_a = aligned16(A) && (lda % 4 == 0) && (sa % 4 == 0);
    const int vec_b = aligned16(Bm) && (ldb % 4 == 0) && (sb % 4 == 0);
    dim3 grid(camli_divup(N, GB_T), camli_divup(M, GB_T), batch);
    if constexpr (!SKIPZ) {
        // unmarked loops (the forward build).  Aligned m-contiguous operands with K % 16 == 0 (every level of the build)
        // take the direct-to-LDS kernel; CAMLI_GEMM_DMA=0 keeps them on the register-staged two-phase loop for A/B runs.
        // Otherwise: K step 16 -- 33 KB of LDS and <= 168 registers, three workgroups per CU (CAMLI_GEMM_KS=32 restores the
        // 67 KB / two-workgroup form).
        static const bool ks32 = []() { const char* e = getenv("CAMLI_GEMM_KS"); return e && atoi(e) == 32; }();
        static const bool dma = []() { const char* e = getenv("CAMLI_GEMM_DMA"); return !e || atoi(e) != 0; }();
        if constexpr (!A_KC && !B_KC) {
            // r5: the 128 x 128-per-wave persistent kernel (gemm_w128.h; 0.85 of the fp32 MFMA peak on the level-0 build where
            // the 128 x 128-per-workgroup kernel below reaches 0.67) for every level wide enough to fill its 256-column tiles
            if (!accumulate && launch_w128(A, Bm, C, M, N, K, lda, ldb, ldc, sa, sb, sc, batch, alpha, vec_a && vec_b, stream)) return;
            constexpr size_t ldsd = (size_t)2 * 16 * (128 + 128) * sizeof(float);
            if (dma && K % 16 == 0 && vec_a && vec_b && M % 4 == 0 && N % 4 == 0 && M >= 4 && N >= 4) {
                if (accumulate)
                    hipLaunchKernelGGL((gemm_fwd_dma_kernel<true>), grid, dim3(256), ldsd, stream, A, Bm, C, M, N, K, lda, ldb, ldc,
                                       sa, sb, sc, alpha);
                else
                    hipLaunchKernelGGL((gemm_fwd_dma_kernel<false>), grid, dim3(256), ldsd, stream, A, Bm, C, M, N, K, lda, ldb,
                                       ldc, sa, sb, sc, alpha);
                return;
            }
        }
        if (!ks32 && K % 16 == 0) {
            constexpr size_t lds16 = (size_t)2 * 16 * (OperandTile<A_KC>::LD + OperandTile<B_KC>::LD) * sizeof(float);
            if (accumulate)
                hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KC, B_KC, true, false, 16>), grid, dim3(256), lds16, stream, A, Bm, C,
                                   M, N, K, lda, ldb, ldc, sa, sb, sc, alpha, vec_a, vec_b, nullptr, 0, 0, 0);
            else
                hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KC, B_KC, false, false, 16>), grid, dim3(256), lds16, stream, A, Bm, C,
                                   M, N, K, lda, ldb, ldc, sa, sb, sc, alpha, vec_a, vec_b, nullptr, 0, 0, 0);
            return;
        }
    }
    constexpr size_t lds = (size_t)2 * GB_K * (OperandTile<A_KC>::LD + OperandTile<B_KC>::LD) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<A_KC, B_KC, true, SKIPZ>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<A_KC, B_KC, false, SKIPZ>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (accumulate)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KC, B_KC, true, SKIPZ>), grid, dim3(256), lds, stream, A, Bm, C, M, N, K, lda,
                           ldb, ldc, sa, sb, sc, alpha, vec_a, vec_b, marks, mark_mode, mark_tb, mark_src_blocks);
    else
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KC, B_KC, false, SKIPZ>), grid, dim3(256), lds, stream, A, Bm, C, M, N, K, lda,
                           ldb, ldc, sa, sb, sc, alpha, vec_a, vec_b, marks, mark_mode, mark_tb, mark_src_blocks);
}

int build_args_ok(const char* what, const void* f1, const void* a, const void* b, const int* p_levels, int L, int B, int C,
                  int P) {
    if (!f1 || !a || !b || !p_levels) { camli_set_error("%s: null pointer", what); return 0; }
    if (L < 1 || L > 8 || B < 0 || C < 1 || P < 1 || B > 65535 || P >= (1 << 28) || C >= (1 << 28)) {
        camli_set_error("%s: bad shape L=%d B=%d C=%d P=%d", what, L, B, C, P);
        return 0;
    }
    for (int l = 0; l < L; ++l)
        if (p_levels[l] < 1 || p_levels[l] >= (1 << 28)) { camli_set_error("%s: level %d has %d target pixels", what, l, p_levels[l]); return 0; }
    return 1;
}

}  // namespace

// V_l[b, p, q] = scale * sum_c f1[b, c, p] * f2_l[b, c, q]       f1 [B,C,P], f2_l [B,C,P_l], V_l [B,P,P_l]
extern "C" int camli_allpairs_build_fwd(const float* f1, const float* const* f2_levels, float* const* vol_levels,
                                        const int* p_levels, int L, int B, int C, int P, float scale, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!build_args_ok("camli_allpairs_build_fwd", f1, f2_levels, vol_levels, p_levels, L, B, C, P)) return CAMLI_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int l = 0; l < L; ++l) {
        if (!f2_levels[l] || !vol_levels[l]) { camli_set_error("camli_allpairs_build_fwd: null level pointer"); return CAMLI_EINVAL; }
        const int Pl = p_levels[l];
        // M = P (source pixel), N = P_l (target pixel), K = C;  A = f1 [K][M], B = f2_l [K][N]
        launch_gemm<false, false, false>(f1, f2_levels[l], vol_levels[l], P, Pl, C, P, Pl, Pl, (int64_t)C * P, (int64_t)C * Pl,
                                  (int64_t)P * Pl, B, scale, false, s);
    }
    return camli_check_launch("camli_allpairs_build_fwd");
}

// g_f1[b, c, p]   = scale * sum_l sum_q gV_l[b, p, q] * f2_l[b, c, q]         (fully written)
// g_f2_l[b, c, q] = scale * sum_p f1[b, c, p] * gV_l[b, p, q]                 (fully written, per level)
static size_t gf2_workspace_floats(const int* p_levels, int L, int B, int C, int P) {
    const int steps = camli_divup(P, GB_K);
    size_t n = 0;
    for (int l = 0; l < L; ++l) {
        const int parts = gf2_parts(l, steps);
        if (parts > 1) n += (size_t)parts * B * C * p_levels[l];
    }
    return n;
}

static int allpairs_build_bwd_impl(const float* f1, const float* const* f2_levels, const float* const* gvol_levels,
                                   const int* p_levels, int L, float* g_f1, float* const* g_f2_levels, int B, int C, int P,
                                   float scale, const unsigned char* const* marks, void* stream, const char* what,
                                   float* workspace = nullptr, size_t workspace_bytes = 0) {
    if (B == 0) return CAMLI_OK;
    if (!build_args_ok(what, f1, f2_levels, gvol_levels, p_levels, L, B, C, P)) return CAMLI_EINVAL;
    if (!g_f1 || !g_f2_levels) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int sb = camli_divup(P, 32);
    for (int l = 0; l < L; ++l)
        if (!f2_levels[l] || !gvol_levels[l] || !g_f2_levels[l] || (marks && !marks[l])) {
            camli_set_error("%s: null level pointer", what);
            return CAMLI_EINVAL;
        }
    // g_f2_l for every level in one launch: M = C, N = P_l, K = P;  A = f1 [M][K] (k contiguous), B = gV_l [K][N] (n contiguous)
    {
        const bool split = workspace && workspace_bytes >= gf2_workspace_floats(p_levels, L, B, C, P) * sizeof(float);
        GemmGroup g;
        ReduceGroup rg;
        int ng = 0, nr = 0, tiles = 0, vec_ok = aligned16(f1) ? 1 : 0;
        float* ws = workspace;
        int64_t max_n = 0;
        for (int i = 0; i < L; ++i) {
            const int l = L - 1 - i;            // coarse levels first
            const int parts = split ? gf2_parts(l, sb) : 1;
            const int64_t n = (int64_t)B * C * p_levels[l];
            if (parts > 1) {
                rg.out[nr] = g_f2_levels[l];
                rg.part[nr] = ws;
                rg.parts[nr] = parts;
                rg.n[nr] = n;
                max_n = n > max_n ? n : max_n;
                ++nr;
            }
            for (int part = 0; part < parts; ++part, ++ng) {
                g.B[ng] = gvol_levels[l];
                g.C[ng] = parts > 1 ? ws + (int64_t)part * n : g_f2_levels[l];
                g.marks[ng] = (marks && P <= 256 * GB_K) ? marks[l] : nullptr;
                g.N[ng] = p_levels[l];
                g.ks_begin[ng] = (int)((int64_t)sb * part / parts);
                g.ks_end[ng] = (int)((int64_t)sb * (part + 1) / parts);
                g.tile_begin[ng] = tiles;
                tiles += camli_divup(p_levels[l], GB_T);
            }
            if (parts > 1) ws += (int64_t)parts * n;
            vec_ok = vec_ok && aligned16(gvol_levels[l]);
        }
        g.groups = ng;
        g.tile_begin[ng] = tiles;
        const int vec_a = aligned16(f1) && (P % 4 == 0) && (((int64_t)C * P) % 4 == 0);
        constexpr size_t lds = (size_t)2 * GB_K * (OperandTile<true>::LD + OperandTile<false>::LD) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_gf2_grouped_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_gf2_grouped_kernel, dim3(tiles, camli_divup(C, GB_T), B), dim3(256), lds, s, f1, g, C, P,
                           (int64_t)P, (int64_t)C * P, scale, vec_a, vec_ok, sb);
        if (nr > 0) {
            const int blocks = (int)((max_n + 1023) / 1024 < 2048 ? (max_n + 1023) / 1024 : 2048);
            hipLaunchKernelGGL(reduce_parts_kernel, dim3(blocks, nr), dim3(256), 0, s, rg);
        }
    }
    for (int l = 0; l < L; ++l) {
        const int Pl = p_levels[l];
        const unsigned char* mk = marks ? marks[l] : nullptr;
        const int tb = camli_divup(Pl, 32);
        // g_f1: M = C, N = P, K = P_l;  A = f2_l [M][K] (k contiguous), B = gV_l [N][K] (k contiguous); accumulate over levels
        launch_gemm<true, true, true>(f2_levels[l], gvol_levels[l], g_f1, C, P, Pl, Pl, Pl, P, (int64_t)C * Pl, (int64_t)P * Pl,
                                (int64_t)C * P, B, scale, l > 0, s, mk, 1, tb, sb);
    }
    return camli_check_launch(what);
}

extern "C" int camli_allpairs_build_bwd(const float* f1, const float* const* f2_levels, const float* const* gvol_levels,
                                        const int* p_levels, int L, float* g_f1, float* const* g_f2_levels, int B, int C,
                                        int P, float scale, void* stream) {
    return allpairs_build_bwd_impl(f1, f2_levels, gvol_levels, p_levels, L, g_f1, g_f2_levels, B, C, P, scale, nullptr, stream,
                                   "camli_allpairs_build_bwd");
}

// Workspace of the split-K form below (bytes; 0 when no level would be split).
extern "C" int64_t camli_allpairs_build_bwd_workspace_bytes(const int* p_levels, int L, int B, int C, int P) {
    if (!p_levels || L < 1 || L > 8 || B < 1 || C < 1 || P < 1) return 0;
    return (int64_t)(gf2_workspace_floats(p_levels, L, B, C, P) * sizeof(float));
}

// camli_allpairs_build_bwd_marked (marks != NULL) or camli_allpairs_build_bwd (marks == NULL) with the g_f2 GEMMs of the
// coarse levels split over K into 2 / 4 / 8 workgroups per tile; the parts of a level are written to the workspace and added
// in a fixed order.  A NULL or too small workspace runs the unsplit form.
extern "C" int camli_allpairs_build_bwd_splitk(const float* f1, const float* const* f2_levels, const float* const* gvol_levels,
                                               const int* p_levels, int L, float* g_f1, float* const* g_f2_levels, int B, int C,
                                               int P, float scale, const unsigned char* const* marks, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) {
        camli_set_error("camli_allpairs_build_bwd_splitk: workspace must be 16-byte aligned");
        return CAMLI_EINVAL;
    }
    return allpairs_build_bwd_impl(f1, f2_levels, gvol_levels, p_levels, L, g_f1, g_f2_levels, B, C, P, scale, marks, stream,
                                   "camli_allpairs_build_bwd_splitk", static_cast<float*>(workspace),
                                   workspace_bytes > 0 ? (size_t)workspace_bytes : 0);
}

// Same adjoint, skipping every K step whose gradient tile was never written: marks[l] = [B][ceil(P/32)][ceil(P_l/32)]
// bytes, non-zero where camli_allpairs_lookup_bwd_marked added a window into that 32x32 block of level l (a superset of
// the non-zero blocks is enough; an all-zero gradient volume with all-zero marks yields zero gradients).
extern "C" int camli_allpairs_build_bwd_marked(const float* f1, const float* const* f2_levels,
                                               const float* const* gvol_levels, const int* p_levels, int L, float* g_f1,
                                               float* const* g_f2_levels, int B, int C, int P, float scale,
                                               const unsigned char* const* marks, void* stream) {
    if (!marks && B > 0) { camli_set_error("camli_allpairs_build_bwd_marked: null marks"); return CAMLI_EINVAL; }
    return allpairs_build_bwd_impl(f1, f2_levels, gvol_levels, p_levels, L, g_f1, g_f2_levels, B, C, P, scale, marks, stream,
                                   "camli_allpairs_build_bwd_marked");
}

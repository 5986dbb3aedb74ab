// The Winograd-domain weight gradient's contraction (winograd.h): for every plane t and K split s
//
//     part[s][t][i][j] = sum_{k in split s} X[t][i][k] * Wt[t][j][k]            k = tiles
//
// = the k-contiguous machinery of convcl.h (one tap, no image borders) on one slice of operands per (plane, split): 256 x 32 NTW
// tile per workgroup, 64-byte LDS rows filled direct-to-LDS, asm MFMA rows on pinned accumulators.  The operands come in the
// chunk-major layout [P][NT / 16][rows][16] (winograd.h): row r of K chunk k is 64 bytes at (k * rows + r) * 64, so one
// direct-to-LDS instruction (16 rows x 64 bytes) reads 1 KB contiguous -- with row-major [rows][NT] operands the same
// instruction touched 16 half cache lines 4 NT bytes apart and the contraction ran at 0.5 of the matrix rate.  No atomics:
// the parts are summed in a fixed order by wino::wrw_reduce_kernel.
#pragma once
#include "convcl.h"

namespace wino {

struct WrwBatch {
    ccl::Problem base;            // x = X (row side), w = Wt (column side), y = parts; B = H = 1, W = rows of X; T = 1
    int planes;                   // P = (M + 2)^2 transform-domain planes
    int q_chunks;                 // 16-float K chunks per split
    int total_chunks;             // NT / 16
    int64_t x_plane, w_plane;     // floats between the planes of X / Wt
    int64_t y_part;               // floats per part (rows x ldy)
};

template <int NTW, int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wrw_planes_kernel(WrwBatch wb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ccl::Problem& p = wb.base;
    const int z = blockIdx.y, s = z / wb.planes, t = z - s * wb.planes;
    const int c0 = s * wb.q_chunks;
    const int chunks = min(wb.q_chunks, wb.total_chunks - c0);
    const float* x = p.x + (int64_t)t * wb.x_plane + (int64_t)c0 * p.xk;
    const float* w = p.w + (int64_t)t * wb.w_plane + (int64_t)c0 * p.wk;
    float* y = p.y + (int64_t)z * wb.y_part;
    const ccl::Operands o = {x, x, w, y, y, chunks * 16, chunks * 16};
    ccl::convcl_body<NTW, NBUF, ccl::EPI_PLAIN>(p, o, lds);
}

}  // namespace wino

// Batched 2-D transposes between channel-first and channels-last maps, gfx950.
//
// The library runs the update block's wide convolutions (GRU2D 1x5 / 5x1, raft_core.py:112-139) on NHWC implicit-GEMM
// kernels and wraps each call on NCHW tensors in layout transposes of its own -- six volume-sized passes per layer and
// training step (x, y, gy twice, gx, x again for the weight gradient).  The cores feed those convolutions channels-last
// operands instead (cores/blocks.py:_CatConvCL): the concatenation that builds the input writes straight into a
// channels-last buffer that is kept for the weight gradient, and the output gradient is transposed once for both adjoints.
// These are the passes that remain:
//     dst[b][j][i] = src[b][i][j]      i < rows, j < cols;  element (b, i, j) of src at b * src_bs + i * src_rs + j,
//                                      element (b, j, i) of dst at b * dst_bs + j * dst_rs + i
// so a channel SLICE of a wider channels-last map is a dst / src with row stride = the wide channel count.
// HBM-bound: 8 bytes per element.  64x64 tiles through LDS (row stride 65: the transposing accesses of a wave spread over
// all banks), 16-byte global accesses on both sides when strides and pointers allow, scalar otherwise.
#include "camli_common.h"

namespace {

constexpr int TT = 64;

// grid (ceil(cols/64), ceil(rows/64), B), block 256
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_planes_kernel(const float* __restrict__ src, int64_t src_bs, int64_t src_rs,
                                                                float* __restrict__ dst, int64_t dst_bs, int64_t dst_rs, int rows,
                                                                int cols) {
    __shared__ float tile[TT][TT + 1];
    const int i0 = blockIdx.y * TT, j0 = blockIdx.x * TT;
    const float* __restrict__ s = src + (int64_t)blockIdx.z * src_bs;
    float* __restrict__ d = dst + (int64_t)blockIdx.z * dst_bs;
    const int tid = threadIdx.x;
    if (VEC) {
        const int q = tid & 15, r = tid >> 4;            // 16 float4 per tile row, 16 rows per sweep
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = i0 + r + 16 * it, j = j0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < rows && j < cols) v = *reinterpret_cast<const float4*>(s + (int64_t)i * src_rs + j);      // cols % 4 == 0
            tile[4 * q + 0][r + 16 * it] = v.x;
            tile[4 * q + 1][r + 16 * it] = v.y;
            tile[4 * q + 2][r + 16 * it] = v.z;
            tile[4 * q + 3][r + 16 * it] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int jj = r + 16 * it, i = i0 + 4 * q, j = j0 + jj;
            if (j < cols && i < rows) {                                                                       // rows % 4 == 0
                const float4 v = make_float4(tile[jj][4 * q], tile[jj][4 * q + 1], tile[jj][4 * q + 2], tile[jj][4 * q + 3]);
                *reinterpret_cast<float4*>(d + (int64_t)j * dst_rs + i) = v;
            }
        }
    } else {
        const int c = tid & 63, r = tid >> 6;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int i = i0 + r + 4 * it, j = j0 + c;
            tile[c][r + 4 * it] = (i < rows && j < cols) ? s[(int64_t)i * src_rs + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int jj = r + 4 * it, i = i0 + c, j = j0 + jj;
            if (j < cols && i < rows) d[(int64_t)j * dst_rs + i] = tile[jj][c];
        }
    }
}

}  // namespace

extern "C" int camli_transpose_planes(const float* src, int64_t src_batch_stride, int64_t src_row_stride, float* dst,
                                      int64_t dst_batch_stride, int64_t dst_row_stride, int B, int rows, int cols, void* stream) {
    if (B == 0 || rows == 0 || cols == 0) return CAMLI_OK;
    if (!src || !dst) { camli_set_error("camli_transpose_planes: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || B > 65535 || rows < 0 || cols < 0 || src_row_stride < cols || dst_row_stride < rows || src_batch_stride < 0 ||
        dst_batch_stride < 0) {
        camli_set_error("camli_transpose_planes: bad shape B=%d rows=%d cols=%d strides %lld %lld", B, rows, cols,
                        (long long)src_row_stride, (long long)dst_row_stride);
        return CAMLI_EINVAL;
    }
    if (camli_divup(rows, TT) > 65535) { camli_set_error("camli_transpose_planes: rows=%d too large", rows); return CAMLI_EINVAL; }
    const bool vec = (rows % 4 == 0) && (cols % 4 == 0) && (src_row_stride % 4 == 0) && (dst_row_stride % 4 == 0) &&
                     (src_batch_stride % 4 == 0) && (dst_batch_stride % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    dim3 grid(camli_divup(cols, TT), camli_divup(rows, TT), B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec)
        hipLaunchKernelGGL((transpose_planes_kernel<true>), grid, dim3(256), 0, s, src, src_batch_stride, src_row_stride, dst,
                           dst_batch_stride, dst_row_stride, rows, cols);
    else
        hipLaunchKernelGGL((transpose_planes_kernel<false>), grid, dim3(256), 0, s, src, src_batch_stride, src_row_stride, dst,
                           dst_batch_stride, dst_row_stride, rows, cols);
    return camli_check_launch("camli_transpose_planes");
}

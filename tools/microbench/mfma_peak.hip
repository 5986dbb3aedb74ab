// Calibration: what does the fp32 matrix pipe of gfx950 deliver to (a) a pure chain of v_mfma_f32_32x32x2_f32 on four
// independent accumulators, (b) the same chain fed from LDS the way gemm_f32_mfma_kernel feeds it (one ds_read_b32 per
// operand fragment, k-major tiles)?   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o tools/microbench/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    __shared__ float tile[2 * 32 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 32 * 132; i += 256) tile[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int fk = lane >> 5, fm = lane & 31;
    const int wm = ((threadIdx.x >> 6) & 1) * 64, wn = ((threadIdx.x >> 7) & 1) * 64;
    float a0 = 1.0f + lane, a1 = 2.0f, b0 = 0.5f, b1 = 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            if (LDS) {
                const float* sa = tile + (k + fk) * 132;
                const float* sb = tile + 32 * 132 + (k + fk) * 132;
                a0 = sa[wm + fm]; a1 = sa[wm + 32 + fm];
                b0 = sb[wn + fm]; b1 = sb[wn + 32 + fm];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool LDS>
static void run(const char* name, int blocks, int iters) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(mfma_loop<LDS>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(mfma_loop<LDS>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 64 /*mfma per iter*/ * 4096.0;
    printf("%-28s blocks %5d iters %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, iters, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        run<false>("pure mfma chain", 256 * per_cu, 2000);
        run<true>("mfma fed from LDS (b32)", 256 * per_cu, 2000);
    }
    return 0;
}

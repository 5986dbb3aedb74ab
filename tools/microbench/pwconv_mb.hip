// Stand-alone driver of the 128 -> 128 pointwise convolution (tools/microbench/pwconv.h: an experiment, not on the product path -- profiles/r06_experiments.txt 17): against a one-thread-per-output
// fp64 reference, then timing at the shapes of the step (batch 8: 8,160 image positions, 2,048 points).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/microbench tools/microbench/pwconv_mb.hip -o tools/microbench/bin/pwconv_mb
#include "pwconv.h"
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = ((float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale;
    }
}

__global__ void ref_kernel(const float* x, const float* w, const float* bias, float* y, int P, int act) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (p >= P) return;
    double acc = bias[o];
    for (int c = 0; c < 128; ++c) acc += (double)w[o * 128 + c] * (double)x[((size_t)b * 128 + c) * P + p];
    float v = (float)acc;
    if (act == 1) v = v > 0.f ? v : 0.f;
    if (act == 2) v = v > 0.f ? v : 0.1f * v;
    y[((size_t)b * 128 + o) * P + p] = v;
}

__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}

static void run(int B, int P, int reps) {
    const size_t n = (size_t)B * 128 * P;
    float *x, *w, *bias, *y, *yr, *d;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&yr, n * 4)); CK(hipMalloc(&w, 128 * 128 * 4)); CK(hipMalloc(&bias, 128 * 4));
    CK(hipMalloc(&d, 4));
    fill_kernel<<<1024, 256>>>(x, n, 1u, 1.0f);
    fill_kernel<<<64, 256>>>(w, 128 * 128, 2u, 0.09f);
    fill_kernel<<<1, 128>>>(bias, 128, 3u, 0.5f);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pwc::pwconv128_fwd_kernel<pwc::ACT_LEAKY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwc::LDS_BYTES));
    const int tiles = (P + 127) / 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pwc::pwconv128_pipe_kernel<pwc::ACT_LEAKY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwc::LDS2_BYTES));
    const int tiles64 = (P + 63) / 64, total = B * tiles64;
    const char* ge = getenv("MB_GRID");
    const int cap = ge ? atoi(ge) : 512;
    const bool pipe = !getenv("MB_PLAIN");
    auto launch = [&]() {
        if (pipe) hipLaunchKernelGGL((pwc::pwconv128_pipe_kernel<pwc::ACT_LEAKY>), dim3(total < cap ? total : cap), dim3(256), pwc::LDS2_BYTES, 0, x, (int64_t)128 * P, w, bias, y, (int64_t)128 * P, P, tiles64, total);
        else hipLaunchKernelGGL((pwc::pwconv128_fwd_kernel<pwc::ACT_LEAKY>), dim3(B * tiles), dim3(256), pwc::LDS_BYTES, 0, x, (int64_t)128 * P, w, bias, y, (int64_t)128 * P, P, tiles);
    };
    launch();
    ref_kernel<<<dim3((P + 255) / 256, 128, B), 256>>>(x, w, bias, yr, P, 2);
    CK(hipMemset(d, 0, 4));
    maxdiff_kernel<<<1024, 256>>>(y, yr, n, d);
    float md = 0.f;
    CK(hipMemcpy(&md, d, 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flop = 2.0 * B * 128.0 * 128.0 * P, bytes = 2.0 * n * 4;
    printf("B=%d P=%d: max |own - fp64 reference| %.3g;  %.2f us per launch = %.1f TFLOP/s (%.2f of the fp32 matrix peak), %.2f TB/s of algorithmic bytes\n", B, P, md, us,
           flop / us / 1e6, flop / us / 1e6 / 157.3, bytes / us / 1e6);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(yr)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(d));
}

int main() {
    run(8, 8160, 50);
    run(8, 2048, 50);
    run(8, 8192, 50);
    run(4, 8160, 50);
    run(1, 7332, 50);
    run(2, 204, 50);
    return 0;
}

// Stand-alone driver of the 128 x 128-per-wave fp32 GEMM core (camliflow_amd/csrc/hip/gemm_w128.h): bit-exactness against
// a one-thread-per-output fmaf chain, then timing of the level-0 / level-1 shapes of the all-pairs build at batch 8.
//   hipcc --offload-arch=gfx950 -O3 -I camliflow_amd/csrc/hip tools/microbench/gemm_w128_mb.hip -o tools/microbench/gemm_w128_mb
//   (-DMB_KS=16 -DMB_NBUF=4 select the instantiation; -DMB_ABL=1|2 the timing-only ablations)
#include "gemm_w128.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#ifndef MB_KS
#define MB_KS 16
#endif
#ifndef MB_NBUF
#define MB_NBUF 4
#endif
#ifndef MB_ABL
#define MB_ABL 0
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;      // uniform [-1, 1)
    }
}

__global__ void ref_kernel(const float* A, const float* B, float* C, int M, int N, int K, float alpha) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y, b = blockIdx.z;
    if (n >= N) return;
    const float* a = A + (size_t)b * K * M + m;
    const float* bb = B + (size_t)b * K * N + n;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(a[(size_t)k * M], bb[(size_t)k * N], acc);
    C[((size_t)b * M + m) * N + n] = alpha * acc;
}

__global__ void diff_kernel(const float* x, const float* y, size_t n, unsigned long long* bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__float_as_uint(x[i]) != __float_as_uint(y[i])) atomicAdd(bad, 1ull);
}

static w128::Problem make_problem(const float* A, const float* B, float* C, int M, int N, int K, int batch, float alpha) {
    w128::Problem p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = M; p.ldb = N; p.ldc = N;
    p.sa = (int64_t)K * M; p.sb = (int64_t)K * N; p.sc = (int64_t)M * N;
    p.alpha = alpha;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
    p.tiles = batch * p.tiles_m * p.tiles_n;
    return p;
}

static void launch(const w128::Problem& p, int nwg, hipStream_t s) {
    constexpr size_t lds = (size_t)MB_NBUF * MB_KS * 512 * sizeof(float);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&w128::gemm_w128_kernel<MB_KS, MB_NBUF, MB_ABL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set = true;
    }
    hipLaunchKernelGGL((w128::gemm_w128_kernel<MB_KS, MB_NBUF, MB_ABL>), dim3(nwg), dim3(256), lds, s, p);
}

static void run(int M, int N, int K, int batch, bool check, int reps) {
    float *A, *B, *C, *R = nullptr;
    const size_t na = (size_t)batch * K * M, nb = (size_t)batch * K * N, nc = (size_t)batch * M * N;
    CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, A, na, 12345u);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, B, nb, 777u);
    CK(hipMemset(C, 0xFF, nc * 4));
    const float alpha = 0.0625f;
    w128::Problem p = make_problem(A, B, C, M, N, K, batch, alpha);
    const int nwg = 256;
    launch(p, nwg, 0);
    CK(hipDeviceSynchronize());
    if (check && MB_ABL == 0) {
        CK(hipMalloc(&R, nc * 4));
        hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, M, batch), dim3(256), 0, 0, A, B, R, M, N, K, alpha);
        unsigned long long* bad;
        CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, C, R, nc, bad);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("check M=%d N=%d K=%d batch=%d: %llu of %zu outputs differ %s\n", M, N, K, batch, h, nc, h ? "FAIL" : "bit-exact");
        if (h) {
            std::vector<float> hc(256), hr(256);
            CK(hipMemcpy(hc.data(), C, 1024, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), R, 1024, hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; ++i) printf("  [%d] got %g want %g\n", i, hc[i], hr[i]);
        }
        CK(hipFree(R)); CK(hipFree(bad));
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch(p, nwg, 0);
        float best = 1e30f, sum = 0.f;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0));
            launch(p, nwg, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        const double flop = 2.0 * batch * (double)M * N * K;
        printf("KS=%d NBUF=%d ABL=%d  M=%d N=%d K=%d batch=%d: avg %.1f us (%.1f TFLOP/s, %.3f of 157.3)  best %.1f us (%.3f)\n", MB_KS,
               MB_NBUF, MB_ABL, M, N, K, batch, sum / reps * 1e3, flop / (sum / reps) / 1e9, flop / (sum / reps) / 1e9 / 157.3,
               best * 1e3, flop / best / 1e9 / 157.3);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && atoi(argv[1]) == 1;
    // edge shapes: M, N not multiples of 256, a single tile, a narrow level
    run(1000, 520, 256, 3, true, 0);
    run(256, 256, 64, 1, true, 0);
    run(8160, 120, 256, 2, true, 0);
    run(520, 2040, 128, 2, true, 0);
    if (quick) return 0;
    run(8160, 8160, 256, 8, true, 20);      // level 0 of the build at batch 8, 68 x 120
    run(8160, 2040, 256, 8, false, 20);     // level 1
    return 0;
}

// Stand-alone driver of the Winograd F(2x2,3x3) convolution (camliflow_amd/csrc/hip/winograd.h + gemm_w128.h): error against
// an fp64 direct convolution (and the fp32 direct form's own error beside it), then the three launches timed one by one and
// together at the update block's shapes (batch 8, 68 x 120).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I camliflow_amd/csrc/hip tools/microbench/winograd_mb.hip -o tools/microbench/bin/winograd_mb
#include "gemm_w128.h"
#include "winograd.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = ((float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale;
    }
}

// direct convolution, one thread per output: fp64 accumulation (T = double) or the fp32 fmaf chain (T = float)
template <typename T>
__global__ void ref_conv_kernel(const float* x, const float* w, const float* bias, float* y32, double* y64, int B, int C, int N, int H, int W, int relu) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N * H * W) return;
    const int ox = i % W, oy = (i / W) % H, n = (i / ((size_t)W * H)) % N, b = i / ((size_t)W * H * N);
    T acc = 0;
    for (int c = 0; c < C; ++c)
        for (int a = 0; a < 3; ++a)
            for (int bb = 0; bb < 3; ++bb) {
                const int yy = oy + a - 1, xx = ox + bb - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    acc += (T)x[(((size_t)b * C + c) * H + yy) * W + xx] * (T)w[((size_t)n * C + c) * 9 + a * 3 + bb];
            }
    acc += (T)bias[n];
    if (relu && acc < 0) acc = 0;
    if (y64) y64[i] = (double)acc; else y32[i] = (float)acc;
}

__global__ void err_kernel(const float* y, const double* r, size_t n, double* out /* max abs err, max abs ref, sum sq err, sum sq ref */) {
    double me = 0, mr = 0, se = 0, sr = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double e = fabs((double)y[i] - r[i]);
        me = fmax(me, e); mr = fmax(mr, fabs(r[i])); se += e * e; sr += r[i] * r[i];
    }
    // (coarse: atomics on doubles)
    atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(me));
    atomicMax(reinterpret_cast<unsigned long long*>(out + 1), (unsigned long long)__double_as_longlong(mr));
    atomicAdd(out + 2, se); atomicAdd(out + 3, sr);
}

template <int GA, int GB, int WM>
#ifndef MB_TILE
#define MB_TILE 2
#endif
#ifndef MB_NBUF
#define MB_NBUF 3
#endif
static void launch_gemm(const float* U, const float* V, float* Mo, int Mp, int NT, int K, hipStream_t s) {
    constexpr int KS = 16, NBUF = MB_NBUF;
    constexpr size_t lds = (size_t)NBUF * KS * 512 * sizeof(float);
    auto kern = &w128::gemm_w128_kernel<KS, NBUF, 0, GA, GB, WM>;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    w128::Problem p;
    p.A = U; p.B = V; p.C = Mo; p.M = Mp; p.N = NT; p.K = K;
    p.lda = Mp; p.ldb = NT; p.ldc = NT;
    p.sa = (int64_t)K * Mp; p.sb = (int64_t)K * NT; p.sc = (int64_t)Mp * NT;
    p.alpha = 1.0f;
    p.tiles_m = (Mp + w128::tile_m<GA, WM>() - 1) / w128::tile_m<GA, WM>();
    p.tiles_n = (NT + w128::tile_n<GB, WM>() - 1) / w128::tile_n<GB, WM>();
    p.tiles = (MB_TILE + 2) * (MB_TILE + 2) * p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, s, p);
}

static void gemm(const float* U, const float* V, float* Mo, int Mp, int NT, int K, hipStream_t s) {
    const int rem = Mp % 256 == 0 ? 256 : Mp % 256;
    if (Mp <= 128) launch_gemm<2, 1, 1>(U, V, Mo, Mp, NT, K, s);
    else if (Mp <= 192) launch_gemm<3, 1, 1>(U, V, Mo, Mp, NT, K, s);
    else launch_gemm<2, 2, 2>(U, V, Mo, Mp, NT, K, s);
    (void)rem;
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <typename F> float us(F&& f, int reps) {
        for (int i = 0; i < 3; ++i) f();
        float sum = 0;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
        }
        return sum / reps * 1e3f;
    }
};

static void run(int B, int C, int N, int H, int W, bool check, int reps, int relu) {
    const wino::Geometry g = wino::make_geometry(B, H, W, MB_TILE);
    constexpr int PL = (MB_TILE + 2) * (MB_TILE + 2), TPT = 8 / MB_TILE;
    const int Mp = (N + 3) & ~3;
    const size_t nx = (size_t)B * C * H * W, ny = (size_t)B * N * H * W, nw = (size_t)N * C * 9;
    float *x, *w, *bias, *y, *U, *V, *Mo;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&y, ny * 4));
    CK(hipMalloc(&U, (size_t)PL * C * Mp * 4)); CK(hipMalloc(&V, (size_t)PL * C * g.NT * 4)); CK(hipMalloc(&Mo, (size_t)PL * Mp * g.NT * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, x, nx, 12345u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, w, nw, 777u, 1.0f / sqrtf(9.0f * C));
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, 0, bias, (size_t)N, 99u, 0.1f);
    CK(hipMemset(y, 0xFF, ny * 4));
    const bool vec = W % 4 == 0;
    auto wt = [&]() { hipLaunchKernelGGL(wino::weight_transform_kernel<MB_TILE>, dim3((C * Mp + 255) / 256), dim3(256), 0, 0, w, U, N, C, C, Mp, 0); };
    auto it = [&]() {
        dim3 grid((g.NT / TPT + 255) / 256, C);
        if (vec) hipLaunchKernelGGL((wino::input_transform_kernel<MB_TILE, true, false>), grid, dim3(256), 0, 0, x, (int64_t)C * H * W, (int64_t)H * W, nullptr, V, C, C, g);
        else hipLaunchKernelGGL((wino::input_transform_kernel<MB_TILE, false, false>), grid, dim3(256), 0, 0, x, (int64_t)C * H * W, (int64_t)H * W, nullptr, V, C, C, g);
    };
    auto gm = [&]() { gemm(U, V, Mo, Mp, g.NT, C, 0); };
    auto ot = [&]() {
        dim3 grid((g.NT / TPT + 255) / 256, N);
        if (vec) hipLaunchKernelGGL((wino::output_transform_kernel<MB_TILE, true>), grid, dim3(256), 0, 0, Mo, Mp, bias, y, (int64_t)N * H * W, (int64_t)H * W, relu, 0, nullptr, N, g);
        else hipLaunchKernelGGL((wino::output_transform_kernel<MB_TILE, false>), grid, dim3(256), 0, 0, Mo, Mp, bias, y, (int64_t)N * H * W, (int64_t)H * W, relu, 0, nullptr, N, g);
    };
    wt(); it(); gm(); ot();
    CK(hipDeviceSynchronize());
    if (check) {
        double* r64; float* r32; double* stats;
        CK(hipMalloc(&r64, ny * 8)); CK(hipMalloc(&r32, ny * 4)); CK(hipMalloc(&stats, 64));
        const int blocks = (int)((ny + 255) / 256);
        hipLaunchKernelGGL(ref_conv_kernel<double>, dim3(blocks), dim3(256), 0, 0, x, w, bias, nullptr, r64, B, C, N, H, W, relu);
        hipLaunchKernelGGL(ref_conv_kernel<float>, dim3(blocks), dim3(256), 0, 0, x, w, bias, r32, nullptr, B, C, N, H, W, relu);
        double h[4];
        for (int which = 0; which < 2; ++which) {
            CK(hipMemset(stats, 0, 64));
            hipLaunchKernelGGL(err_kernel, dim3(1024), dim3(256), 0, 0, which ? r32 : y, r64, ny, stats);
            CK(hipMemcpy(h, stats, 32, hipMemcpyDeviceToHost));
            printf("  %-22s vs fp64 direct: max abs err %.3e (max |ref| %.3f), relative L2 %.3e\n", which ? "fp32 direct (fmaf chain)" : (MB_TILE == 2 ? "winograd F(2x2,3x3)" : "winograd F(4x4,3x3)"),
                   h[0], h[1], sqrt(h[2] / h[3]));
        }
        CK(hipFree(r64)); CK(hipFree(r32)); CK(hipFree(stats));
    }
    if (reps > 0) {
        Timer t;
        const float t_in = t.us(it, reps), t_g = t.us(gm, reps), t_out = t.us(ot, reps), t_w = t.us(wt, reps);
        const float t_all = t.us([&]() { it(); gm(); ot(); }, reps);
        const double direct = 2.0 * B * H * W * (double)C * N * 9, wflop = 2.0 * PL * (double)Mp * g.NT * C;
        printf("B=%d %d->%d %dx%d (tiles %d, Mp %d): weights %.1f us | input %.1f us (%.2f TB/s) | gemm %.1f us (%.3f of 157.3 TF) | output %.1f us (%.2f TB/s) | "
               "all three %.1f us = %.1f TF/s direct-equivalent (%.3f of the fp32 matrix peak)\n",
               B, C, N, H, W, g.NT, Mp, t_w, t_in, (nx * 4.0 + (double)PL * C * g.NT * 4) / t_in / 1e6, t_g, wflop / t_g / 1e6 / 157.3, t_out,
               ((double)PL * Mp * g.NT * 4 + ny * 4.0) / t_out / 1e6, t_all, direct / t_all / 1e6, direct / t_all / 1e6 / 157.3);
    }
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(y)); CK(hipFree(U)); CK(hipFree(V)); CK(hipFree(Mo));
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && atoi(argv[1]) == 1;
    printf("-- correctness, small / odd shapes\n");
    run(2, 48, 52, 13, 21, true, 0, 0);
    run(1, 64, 192, 16, 24, true, 0, 1);
    run(3, 128, 126, 9, 40, true, 0, 1);
    run(1, 256, 256, 47, 156, true, 0, 0);
    if (quick) return 0;
    printf("-- the update block's convolutions, batch 8, 68 x 120\n");
    run(8, 256, 192, 68, 120, true, 20, 1);      // MotionEncoder2D.conv_c2 (library forward: 518 us)
    run(8, 256, 126, 68, 120, false, 20, 1);     // MotionEncoder2D.conv    (348 us)
    run(8, 128, 256, 68, 120, false, 20, 1);     // FlowHead2D.conv1 / mask head (342 us)
    run(8, 128, 512, 68, 120, false, 20, 1);     // both heads as one convolution (655 us)
    run(8, 192, 256, 68, 120, false, 20, 0);     // conv_c2's data gradient (495 us)
    run(8, 128, 64, 68, 120, false, 20, 1);      // conv_f2 (95 us)
    printf("-- smaller batches (configs[3] per-rank batch 4, KITTI batch 1)\n");
    run(4, 256, 192, 68, 120, false, 20, 1);
    run(1, 256, 192, 47, 156, false, 20, 1);
    return 0;
}

// Stand-alone driver of the channels-last tap convolution (camliflow_amd/csrc/hip/convcl.h): bit-exactness against a
// one-thread-per-output fmaf chain in the kernel's own summation order, then timing of GRU2D's convolutions at batch 8, 68 x 120.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I camliflow_amd/csrc/hip tools/microbench/convcl_mb.hip -o tools/microbench/bin/convcl_mb
#include "convcl.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#ifndef MB_NBUF
#define MB_NBUF 3
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
}

__global__ void ref_kernel(ccl::Problem p, float* y) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int pix = blockIdx.y;
    if (n >= p.Cout) return;
    const int xx = pix % p.W, yy = (pix / p.W) % p.H;
    float acc = 0.f;
    for (int c0 = 0; c0 < p.Cin; c0 += 16)
        for (int t = 0; t < p.T; ++t) {
            const bool ok = (unsigned)(xx + p.dx[t]) < (unsigned)p.W && (unsigned)(yy + p.dy[t]) < (unsigned)p.H;
            const float* xs = p.x + (int64_t)(pix + p.dy[t] * p.W + p.dx[t]) * p.ldx + c0;
            const float* ws = p.w + ((int64_t)n * p.T + t) * p.Cin + c0;
            for (int j = 0; j < 4; ++j)
                for (int q = 0; q < 4; ++q) {
                    const int k = 4 * q + j;
                    acc = __builtin_fmaf(ws[k], ok ? xs[k] : 0.f, acc);
                }
        }
    y[(int64_t)pix * p.ldy + n] = acc;
}

__global__ void diff_kernel(const float* x, const float* y, size_t n, unsigned long long* bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__float_as_uint(x[i]) != __float_as_uint(y[i])) atomicAdd(bad, 1ull);
}

template <int NTW>
static void launch(const ccl::Problem& p, hipStream_t s) {
    constexpr size_t lds = (size_t)MB_NBUF * (256 + 32 * NTW) * 16 * sizeof(float);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ccl::convcl_kernel<NTW, MB_NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set = true;
    }
    const int tiles = p.tiles_p * p.tiles_n;
    hipLaunchKernelGGL((ccl::convcl_kernel<NTW, MB_NBUF>), dim3(tiles < 256 ? tiles : 256), dim3(256), lds, s, p);
}

static void run(int B, int H, int W, int Cin, int Cout, bool vertical, bool check, int reps) {
    const int T = 5, P = B * H * W;
    float *x, *w, *y, *r = nullptr;
    const size_t nx = (size_t)P * Cin, nw = (size_t)Cout * T * Cin, ny = (size_t)P * Cout;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x, nx, 4242u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, w, nw, 99u);
    CK(hipMemset(y, 0xFF, ny * 4));
    ccl::Problem p;
    p.x = x; p.x1 = x; p.w = w; p.y = y; p.y1 = y; p.C0 = Cin; p.N0 = Cout; p.ldx1 = Cin; p.ldy1 = Cout; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.T = T; p.ldx = Cin; p.ldy = Cout;
    p.add = p.h = p.z = x; p.y2 = y; p.ld_add = p.ld_h = p.ld_z = p.ldy2 = Cout; p.acc0 = p.acc1 = p.sanitize = 0;
    const int NT = Cout % 256 == 0 ? 256 : 128;
    p.tiles_p = (P + 255) / 256; p.tiles_n = Cout / NT; p.ldw = T * Cin; p.xk = p.wk = 16; p.xrec = p.x1rec = (uint32_t)((size_t)P * Cin * 4); p.wrec = (uint32_t)(nw * 4);
    for (int t = 0; t < T; ++t) { p.dy[t] = vertical ? t - 2 : 0; p.dx[t] = vertical ? 0 : t - 2; }
    auto go = [&]() { if (NT == 256) launch<8>(p, 0); else launch<4>(p, 0); };
    go();
    CK(hipDeviceSynchronize());
    if (check) {
        CK(hipMalloc(&r, ny * 4));
        hipLaunchKernelGGL(ref_kernel, dim3((Cout + 63) / 64, P), dim3(64), 0, 0, p, r);
        unsigned long long* bad;
        CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, y, r, ny, bad);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("check B=%d %dx%d Cin=%d Cout=%d %s: %llu of %zu outputs differ %s\n", B, H, W, Cin, Cout, vertical ? "5x1" : "1x5", h, ny,
               h ? "FAIL" : "bit-exact");
        if (h) {
            std::vector<float> hc(ny < 4096 ? ny : 4096), hr(hc.size());
            CK(hipMemcpy(hc.data(), y, hc.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), r, hr.size() * 4, hipMemcpyDeviceToHost));
            int shown = 0;
            for (size_t i = 0; i < hc.size() && shown < 8; ++i)
                if (hc[i] != hr[i]) { printf("  [%zu] got %g want %g\n", i, hc[i], hr[i]); ++shown; }
        }
        CK(hipFree(r)); CK(hipFree(bad));
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) go();
        float best = 1e30f, sum = 0.f;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0));
            go();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        const double flop = 2.0 * P * (double)Cout * Cin * T;
        printf("NBUF=%d  B=%d %dx%d Cin=%d Cout=%d %s: avg %.1f us (%.1f TFLOP/s, %.3f of 157.3)  best %.1f us (%.3f)\n", MB_NBUF, B, H, W,
               Cin, Cout, vertical ? "5x1" : "1x5", sum / reps * 1e3, flop / (sum / reps) / 1e9, flop / (sum / reps) / 1e9 / 157.3,
               best * 1e3, flop / best / 1e9 / 157.3);
    }
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y));
}

int main(int argc, char** argv) {
    run(1, 7, 9, 32, 256, false, true, 0);        // one partial tile, every pixel near a border
    run(2, 20, 30, 64, 128, true, true, 0);
    run(3, 17, 33, 48, 256, true, true, 0);
    run(2, 16, 40, 32, 128, false, true, 0);
    if (argc > 1 && atoi(argv[1]) == 1) return 0;
    run(8, 68, 120, 256, 256, false, true, 20);   // z|r 1x5
    run(8, 68, 120, 256, 256, true, false, 20);   // z|r 5x1
    run(8, 68, 120, 256, 128, false, true, 20);   // q 1x5
    run(8, 68, 120, 256, 128, true, false, 20);   // q 5x1
    return 0;
}

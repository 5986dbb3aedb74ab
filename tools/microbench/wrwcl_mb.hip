// Stand-alone driver of the weight-gradient kernel (camliflow_amd/csrc/hip/wrwcl.h): bit-exactness against a one-thread-per-
// output fmaf chain in the kernel's own summation order (pixels ascending inside a part, parts added in order), then timing
// at GRU2D's shapes (batch 8, 68 x 120).
#include "wrwcl.h"
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

// gw[(n * Cin + c) * T + t]
__global__ void ref_kernel(wrw::Problem p, float* gw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, t = blockIdx.z;
    if (n >= p.Cout) return;
    const int P = p.B * p.H * p.W;
    float total = 0.f;
    for (int s = 0; s < p.S; ++s) {
        float acc = 0.f;
        const int k1 = min(P, (s + 1) * p.ksplit);
        for (int pix = s * p.ksplit; pix < k1; ++pix) {
            const int xx = pix % p.W, yy = (pix / p.W) % p.H;
            const bool ok = (unsigned)(xx + p.dx[t]) < (unsigned)p.W && (unsigned)(yy + p.dy[t]) < (unsigned)p.H;
            const float xv = ok ? p.x[(int64_t)(pix + p.dy[t] * p.W + p.dx[t]) * p.ldx + c] : 0.f;
            acc = __builtin_fmaf(xv, p.gy[(int64_t)pix * p.ldg + n], acc);
        }
        total = s == 0 ? acc : total + acc;
    }
    gw[((size_t)n * p.Cin + c) * p.T + t] = total;
}

__global__ void diff_kernel(const float* x, const float* y, size_t n, unsigned long long* bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__float_as_uint(x[i]) != __float_as_uint(y[i])) atomicAdd(bad, 1ull);
}

template <int TBN>
static void launch(const wrw::Problem& p, float* gw, hipStream_t s) {
    constexpr size_t lds = (size_t)MB_NBUF * 16 * (256 + 32 * TBN) * sizeof(float);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wrw::wrw_kernel<TBN, MB_NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set = true;
    }
    const int units = p.S * p.T * p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((wrw::wrw_kernel<TBN, MB_NBUF>), dim3(units), dim3(256), lds, s, p);
    const size_t n_el = (size_t)p.T * p.Cin * p.Cout;
    hipLaunchKernelGGL(wrw::wrw_reduce_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, p.part, gw, p.S, p.T, p.Cin, p.Cout, 0, 0);
}

static void run(int B, int H, int W, int Cin, int Cout, bool vertical, int S, bool check, int reps) {
    const int T = 5, P = B * H * W;
    float *x, *gy, *part, *gw, *r = nullptr, *zero;
    CK(hipMalloc(&zero, 1024)); CK(hipMemset(zero, 0, 1024));
    const size_t nx = (size_t)P * Cin, ng = (size_t)P * Cout, nw = (size_t)Cout * T * Cin;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&gy, ng * 4)); CK(hipMalloc(&gw, nw * 4)); CK(hipMalloc(&part, nw * 4 * S));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x, nx, 4242u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, gy, ng, 99u);
    CK(hipMemset(gw, 0xFF, nw * 4));
    CK(hipMemset(part, 0xFF, nw * 4 * S));
    wrw::Problem p;
    p.x = x; p.x1 = x; p.C0 = Cin; p.ldx1 = Cin; p.zero = zero; p.gy = gy; p.part = part; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.T = T; p.ldx = Cin; p.ldg = Cout;
    p.S = S; p.ksplit = ((P + S - 1) / S + 15) / 16 * 16;
    p.S = (P + p.ksplit - 1) / p.ksplit;
    const int NB = Cout % 256 == 0 ? 256 : 128;
    p.tiles_m = Cin / 256; p.tiles_n = Cout / NB;
    for (int t = 0; t < T; ++t) { p.dy[t] = vertical ? t - 2 : 0; p.dx[t] = vertical ? 0 : t - 2; }
    auto go = [&]() { if (NB == 256) launch<8>(p, gw, 0); else launch<4>(p, gw, 0); };
    go();
    CK(hipDeviceSynchronize());
    if (check) {
        CK(hipMalloc(&r, nw * 4));
        hipLaunchKernelGGL(ref_kernel, dim3((Cout + 63) / 64, Cin, T), dim3(64), 0, 0, p, r);
        unsigned long long* bad;
        CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, gw, r, nw, bad);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("check B=%d %dx%d Cin=%d Cout=%d %s S=%d: %llu of %zu outputs differ %s\n", B, H, W, Cin, Cout, vertical ? "5x1" : "1x5",
               p.S, h, nw, h ? "FAIL" : "bit-exact");
        if (h) {
            std::vector<float> hc(nw < 4096 ? nw : 4096), hr(hc.size());
            CK(hipMemcpy(hc.data(), gw, hc.size() * 4, hipMemcpyDeviceToHost));
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
        printf("NBUF=%d  B=%d %dx%d Cin=%d Cout=%d %s S=%d (kernel + reduce): avg %.1f us (%.1f TFLOP/s, %.3f of 157.3)  best %.1f us (%.3f)\n",
               MB_NBUF, B, H, W, Cin, Cout, vertical ? "5x1" : "1x5", p.S, sum / reps * 1e3, flop / (sum / reps) / 1e9,
               flop / (sum / reps) / 1e9 / 157.3, best * 1e3, flop / best / 1e9 / 157.3);
    }
    CK(hipFree(x)); CK(hipFree(gy)); CK(hipFree(gw)); CK(hipFree(part));
}

int main(int argc, char** argv) {
    run(1, 7, 9, 256, 256, false, 2, true, 0);
    run(2, 20, 30, 256, 128, true, 3, true, 0);
    run(3, 17, 33, 256, 256, true, 5, true, 0);
    run(2, 16, 40, 512, 128, false, 4, true, 0);
    if (argc > 1 && atoi(argv[1]) == 1) return 0;
    run(8, 68, 120, 256, 256, false, 51, true, 20);
    run(8, 68, 120, 256, 256, true, 51, false, 20);
    run(8, 68, 120, 256, 128, false, 51, true, 20);
    run(8, 68, 120, 256, 128, true, 51, false, 20);
    run(8, 68, 120, 256, 256, false, 25, false, 20);
    return 0;
}

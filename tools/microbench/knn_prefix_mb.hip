// Stand-alone driver of the nested-prefix KNN search with the phase stamps compiled in (-DCAMLI_KNN_PROFILE).  Build (repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCAMLI_KNN_PROFILE -I camliflow_amd/csrc/hip -I include \
//         tools/microbench/knn_prefix_mb.hip -o tools/microbench/bin/knn_prefix_mb      (and without the define: _plain)
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "knn.hip"

void camli_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int camli_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return CAMLI_ELAUNCH; }
    return CAMLI_OK;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const bool cube = argc > 1 && atoi(argv[1]) == 1;      // 1: uniform in [0, 10)^3 (tools/kernel_bench.py's clouds)
    const int B = 8, M = 2048, Nq = 2048, K = 16, L = 4;
    const int sizes[4] = {2048, 1024, 512, 256};
    // targets: frustum points (the bench's clouds) in arbitrary order (an FPS prefix order is arbitrary in space too);
    // queries: the same kind of cloud, displaced (the warped source points of a GRU iteration)
    std::vector<float> hi((size_t)B * M * 3), hq((size_t)B * Nq * 3);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xFFFFFF) / 16777216.0f; };
    auto fill = [&](std::vector<float>& v, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            if (cube) { v[3 * i] = 10.0f * rnd(); v[3 * i + 1] = 10.0f * rnd(); v[3 * i + 2] = 10.0f * rnd(); continue; }
            const float z = 5.0f + 30.0f * rnd(), u = rnd() * 959.0f, w = rnd() * 539.0f;
            v[3 * i] = (u - 479.5f) * z / 1050.0f; v[3 * i + 1] = (w - 269.5f) * z / 1050.0f; v[3 * i + 2] = z;
        }
    };
    fill(hi, (size_t)B * M); fill(hq, (size_t)B * Nq);
    float *in, *q; int64_t* out[4];
    CK(hipMalloc(&in, hi.size() * 4)); CK(hipMalloc(&q, hq.size() * 4));
    for (int l = 0; l < L; ++l) CK(hipMalloc(&out[l], (size_t)B * Nq * K * 8));
    CK(hipMemcpy(in, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // the clock of an idle GPU ramps up over milliseconds: 400 back-to-back launches first, then 3 x 100 timed as a train
    for (int r = 0; r < 400; ++r)
        if (camli_knn_prefixes(in, q, out, sizes, L, B, M, Nq, 3, K, nullptr) != CAMLI_OK) return 1;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 100; ++i)
            if (camli_knn_prefixes(in, q, out, sizes, L, B, M, Nq, 3, K, nullptr) != CAMLI_OK) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("camli_knn_prefixes B%d %d/%d/%d/%d Nq%d k%d: %.1f us per launch (train of 100)\n", B, sizes[0], sizes[1], sizes[2], sizes[3], Nq, K, ms * 10);
    }
    unsigned long long sum = 0;
    for (int l = 0; l < L; ++l) {
        std::vector<int64_t> ho((size_t)B * Nq * K);
        CK(hipMemcpy(ho.data(), out[l], ho.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ho.size(); ++i) sum = sum * 1000003ull + (unsigned long long)ho[i];
    }
    printf("checksum of the indices %llu\n", sum);
    // the same search one GRU iteration later: targets displaced by up to 2 % of the cloud's extent, the first call's results as prior
    {
        std::vector<float> hm(hi);
        for (size_t i = 0; i < hm.size(); ++i) hm[i] += (rnd() - 0.5f) * (cube ? 0.2f : 0.6f);
        float* in2; int64_t* out2[4]; int64_t* out3[4];
        CK(hipMalloc(&in2, hm.size() * 4));
        CK(hipMemcpy(in2, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
        for (int l = 0; l < L; ++l) { CK(hipMalloc(&out2[l], (size_t)B * Nq * K * 8)); CK(hipMalloc(&out3[l], (size_t)B * Nq * K * 8)); }
        auto train = [&](bool with_prior, int64_t** o) {
            for (int r = 0; r < 50; ++r)
                if (camli_knn_prefixes_prior(in2, q, o, with_prior ? out : nullptr, sizes, L, B, M, Nq, 3, K, nullptr) != CAMLI_OK) exit(1);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 100; ++i)
                if (camli_knn_prefixes_prior(in2, q, o, with_prior ? out : nullptr, sizes, L, B, M, Nq, 3, K, nullptr) != CAMLI_OK) exit(1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long cs = 0;
            for (int l = 0; l < L; ++l) {
                std::vector<int64_t> ho((size_t)B * Nq * K);
                CK(hipMemcpy(ho.data(), o[l], ho.size() * 8, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < ho.size(); ++i) cs = cs * 1000003ull + (unsigned long long)ho[i];
            }
            printf("moved targets, %s: %.1f us per launch, checksum %llu\n", with_prior ? "with the earlier result as prior" : "no prior", ms * 10, cs);
        };
        train(false, out2);
        train(true, out3);
    }
#ifdef CAMLI_KNN_PROFILE
    unsigned long long prof[16][8];
    CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(camli_knn_prof), sizeof(prof)));
    printf("workgroup (0,0): s_memtime ticks per wave: set-up | scan of the own chunk | merge rounds | snapshots ; rescanned queries\n");
    for (int w = 0; w < 8; ++w)
        printf(" wave %d: %6llu | %6llu | %6llu | %6llu ; %llu\n", w, prof[w][0], prof[w][1], prof[w][2], prof[w][3], prof[w][4]);
    static unsigned wg[64][64][4];
    CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(camli_knn_wg_ticks), sizeof(wg)));
    printf("per workgroup, wave 0: scan / merge / snapshot kiloticks, rescanned queries (rows: batch element, 32 workgroups each)\n");
    for (int b = 0; b < B; ++b) {
        printf(" b%d:", b);
        for (int x = 0; x < 32; ++x) printf(" %u/%u/%u/%u", wg[b][x][0] / 1000, wg[b][x][1] / 1000, wg[b][x][2] / 1000, wg[b][x][3]);
        printf("\n");
    }
#endif
    return 0;
}

// Stand-alone driver of the bucket-pruned FPS kernel with the phase stamps compiled in (-DCAMLI_FPS_PROFILE): where the
// 0.86 us of a dependent step go.  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCAMLI_FPS_PROFILE -I camliflow_amd/csrc/hip -I include \
//         tools/microbench/fps_mb.hip -o tools/microbench/bin/fps_mb
// and once more without -DCAMLI_FPS_PROFILE (bin/fps_mb_plain) for the undisturbed time.
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "fps.hip"

void camli_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int camli_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return CAMLI_ELAUNCH; }
    return CAMLI_OK;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
    const int B = 16, N = 8192, S = 4096;
    // the clouds of the bench (bench.synthetic_batch): pixels of a 960 x 540 image lifted to depths 5 .. 35
    std::vector<float> h((size_t)B * N * 3);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xFFFFFF) / 16777216.0f; };
    for (size_t i = 0; i < (size_t)B * N; ++i) {
        const float z = 5.0f + 30.0f * rnd(), u = rnd() * 959.0f, v = rnd() * 539.0f;
        h[3 * i] = (u - 479.5f) * z / 1050.0f; h[3 * i + 1] = (v - 269.5f) * z / 1050.0f; h[3 * i + 2] = z;
    }
    float* x; int64_t* out;
    CK(hipMalloc(&x, h.size() * 4)); CK(hipMalloc(&out, (size_t)B * S * 8));
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        if (camli_fps(x, out, B, N, S, nullptr) != CAMLI_OK) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("camli_fps B%d %d->%d: %.3f ms = %.3f us per step\n", B, N, S, ms, ms * 1e3 / S);
    }
    std::vector<int64_t> ho((size_t)B * S);
    CK(hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long sum = 0; for (size_t i = 0; i < ho.size(); ++i) sum = sum * 1000003ull + (unsigned long long)ho[i];
    printf("checksum of the picks %llu\n", sum);
#ifdef CAMLI_FPS_PROFILE
    unsigned long long prof[16][8];
    CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(camli_fps_prof), sizeof(prof)));
    const char* names[5] = {"box test", "bucket updates", "wave arg-max", "slot + barrier", "slot reduce"};
    printf("workgroup 0: s_memtime ticks per step, per wave\n");
    for (int w = 0; w < 8; ++w) {
        const double steps = (double)prof[w][7];
        printf(" wave %d:", w);
        double tot = 0;
        for (int k = 0; k < 5; ++k) { printf("  %s %.0f", names[k], prof[w][k] / steps); tot += prof[w][k] / steps; }
        printf("  | total %.0f  buckets/step %.2f  steps with work %.2f  ticks per updated bucket %.0f\n", tot, prof[w][5] / steps,
               prof[w][6] / steps, prof[w][5] ? (double)prof[w][1] / prof[w][5] : 0.0);
    }
    unsigned touch[16][32];
    CK(hipMemcpyFromSymbol(touch, HIP_SYMBOL(camli_fps_touch), sizeof(touch)));
    printf("updates per bucket over the 4 runs (rows: wave, columns: register slot)\n");
    for (int w = 0; w < 8; ++w) {
        printf(" wave %d:", w);
        for (int j = 0; j < 16; ++j) printf(" %6u", touch[w][j]);
        printf("\n");
    }
#endif
    return 0;
}

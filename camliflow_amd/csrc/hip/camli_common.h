// Shared helpers for the gfx950 kernels.  This translation unit set is compiled with
// -ffp-contract=off: every a*b+c below is two roundings unless it is spelled __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAMLI_WAVE 64

// error codes returned by every C-ABI entry point (include/camli_hip.h)
#define CAMLI_OK 0
#define CAMLI_EINVAL (-22)
#define CAMLI_ELAUNCH (-5)
#define CAMLI_ENOTSUP (-95)

void camli_set_error(const char* fmt, ...);
int camli_check_launch(const char* what);

static inline int camli_divup(int a, int b) { return (a + b - 1) / b; }
// 1 KB of zeros on the current device (convcl.hip): the source of padded rows of wrw::wrw_kernel
const float* camli_zero_page();

// Let kernel `fn` use `bytes` of dynamic LDS on the CURRENT device; `done` = the call site's static bit mask of devices
// already served (one process per GPU is the rule here, but a process that drives several must not skip the others).
static inline bool camli_reserve_lds(const void* fn, size_t bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done |= bit;
    return true;
}

// lexicographic arg-max on (value, lowest index) across the 64 lanes of a wave
__device__ __forceinline__ void wave_argmax_lowidx(float& v, int& i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(v, off, CAMLI_WAVE);
        int oi = __shfl_xor(i, off, CAMLI_WAVE);
        bool take = (ov > v) || (ov == v && oi < i);
        v = take ? ov : v;
        i = take ? oi : i;
    }
}

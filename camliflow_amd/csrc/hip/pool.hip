// 3x3 / stride 2 / padding 1 max pooling of the ResNet stem (mmdet ResNet: nn.MaxPool2d(3, 2, 1)) and its adjoint, gfx950.
//
// The stem's pooling runs on the largest activation of the model ([16,64,272,480] = 535 MB for the two frames of a batch of
// 8).  torch's kernels take 404 us forward and 992 us backward on it (max_pool_backward_nchw walks, per INPUT element, every
// output window that could contain it and compares 64-bit flat indices).  Here
//   fwd  lane = output pixel: 9 taps, first maximum in row-major window order wins (a NaN wins too: torch's
//        `val > max || isnan(val)`), the winner's window-local position 0..8 is kept as one byte
//   bwd  lane = INPUT pixel: at most 4 windows cover it (2 per axis when its coordinate is odd); gx = sum of the gradients
//        of those whose byte points at it -- a gather, fully written, no atomics, no zero-fill
// HBM-bound: 4*(H*W + Ho*Wo) + Ho*Wo bytes per plane forward, 4*H*W + 5*Ho*Wo backward.
#include "camli_common.h"

namespace {

// grid (ceil(Wo/256), Ho, planes), block 256
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                unsigned char* __restrict__ arg, int H, int W, int Ho,
                                                                int Wo) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= Wo) return;
    const size_t plane = blockIdx.z;
    const float* __restrict__ xp = x + plane * (size_t)H * W;
    float best = -INFINITY;
    int code = -1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= H) continue;                 // wave-uniform
        const float* __restrict__ row = xp + (size_t)iy * W;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            const float v = row[ix];
            if (code < 0) code = ky * 3 + kx;            // torch: maxidx starts at the window's first valid position
            if (v > best || v != v) {                    // ... and moves on a strictly larger value or a NaN
                best = v;
                code = ky * 3 + kx;
            }
        }
    }
    const size_t o = plane * (size_t)Ho * Wo + (size_t)oy * Wo + ox;
    y[o] = best;
    arg[o] = (unsigned char)(code < 0 ? 0 : code);
}

// grid (ceil(W/256), H, planes), block 256
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ arg,
                                                                float* __restrict__ gx, int H, int W, int Ho, int Wo) {
    const int ix = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y;
    if (ix >= W) return;
    const size_t plane = blockIdx.z;
    const float* __restrict__ gp = gy + plane * (size_t)Ho * Wo;
    const unsigned char* __restrict__ ap = arg + plane * (size_t)Ho * Wo;
    // windows covering (iy, ix): oy in {iy/2} for even iy, {(iy-1)/2, (iy+1)/2} for odd iy; same along x
    const int oy0 = iy >> 1, oy1 = (iy & 1) ? oy0 + 1 : -1;
    const int ox0 = ix >> 1, ox1 = (ix & 1) ? ox0 + 1 : -1;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a == 0 ? oy0 : oy1;
        if (oy < 0 || oy >= Ho) continue;
        const int ky = iy - (2 * oy - 1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = b == 0 ? ox0 : ox1;
            if (ox < 0 || ox >= Wo) continue;
            const int kx = ix - (2 * ox - 1);
            const size_t o = (size_t)oy * Wo + ox;
            if (ap[o] == ky * 3 + kx) acc += gp[o];
        }
    }
    gx[plane * (size_t)H * W + (size_t)iy * W + ix] = acc;
}

bool pool_args_ok(const char* what, long long planes, int H, int W, int Ho, int Wo) {
    if (planes < 0 || H < 1 || W < 1 || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1 ||
        H > 65535 || Ho > 65535 || planes > 2147483647LL) {
        camli_set_error("%s: bad shape planes=%lld H=%d W=%d Ho=%d Wo=%d", what, planes, H, W, Ho, Wo);
        return false;
    }
    return true;
}

}  // namespace

extern "C" int camli_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* arg, int planes, int H, int W, int Ho, int Wo,
                                      void* stream) {
    if (planes == 0) return CAMLI_OK;
    if (!x || !y || !arg) { camli_set_error("camli_maxpool3x3s2_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!pool_args_ok("camli_maxpool3x3s2_fwd", planes, H, W, Ho, Wo)) return CAMLI_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int p0 = 0; p0 < planes; p0 += 65535) {       // grid.z limit
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(camli_divup(Wo, 256), Ho, np), dim3(256), 0, s, x + (size_t)p0 * H * W,
                           y + (size_t)p0 * Ho * Wo, arg + (size_t)p0 * Ho * Wo, H, W, Ho, Wo);
    }
    return camli_check_launch("camli_maxpool3x3s2_fwd");
}

extern "C" int camli_maxpool3x3s2_bwd(const float* gy, const unsigned char* arg, float* gx, int planes, int H, int W, int Ho,
                                      int Wo, void* stream) {
    if (planes == 0) return CAMLI_OK;
    if (!gy || !arg || !gx) { camli_set_error("camli_maxpool3x3s2_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!pool_args_ok("camli_maxpool3x3s2_bwd", planes, H, W, Ho, Wo)) return CAMLI_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(camli_divup(W, 256), H, np), dim3(256), 0, s, gy + (size_t)p0 * Ho * Wo,
                           arg + (size_t)p0 * Ho * Wo, gx + (size_t)p0 * H * W, H, W, Ho, Wo);
    }
    return camli_check_launch("camli_maxpool3x3s2_bwd");
}

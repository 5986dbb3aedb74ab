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

// Forward, two output pixels per thread (W % 4 == 0, Wo even, 16-byte aligned rows): outputs 2q, 2q+1 of row oy read the
// input columns 4q-1 .. 4q+3 = one 16-byte load + the left neighbour per input row.  Same tap order and the same
// "strictly larger or NaN" rule as the one-pixel kernel, so values and arg-max bytes are identical.
// grid (ceil(Ho * Wo/2 / 256), 1, planes), block 256
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 unsigned char* __restrict__ arg, int H, int W, int Ho,
                                                                 int Wo) {
    const int W2 = Wo >> 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Ho * W2) return;
    const int oy = t / W2, q = t - oy * W2;
    const size_t plane = blockIdx.z;
    const float* __restrict__ xp = x + plane * (size_t)H * W;
    float best0 = -INFINITY, best1 = -INFINITY;
    int code0 = -1, code1 = -1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= H) continue;
        const float* __restrict__ row = xp + (size_t)iy * W + 4 * q;
        const float4 v = *reinterpret_cast<const float4*>(row);          // columns 4q .. 4q+3
        const float left = q > 0 ? row[-1] : 0.0f;                       // column 4q-1 (outside the image for q = 0)
        const float tap0[3] = {left, v.x, v.y}, tap1[3] = {v.y, v.z, v.w};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if (!(q == 0 && kx == 0)) {                                   // output 2q: columns 4q-1, 4q, 4q+1
                const float u = tap0[kx];
                if (code0 < 0) code0 = ky * 3 + kx;
                if (u > best0 || u != u) { best0 = u; code0 = ky * 3 + kx; }
            }
            {                                                             // output 2q+1: columns 4q+1 .. 4q+3 (< W: W % 4 == 0)
                const float u = tap1[kx];
                if (code1 < 0) code1 = ky * 3 + kx;
                if (u > best1 || u != u) { best1 = u; code1 = ky * 3 + kx; }
            }
        }
    }
    const size_t o = plane * (size_t)Ho * Wo + (size_t)oy * Wo + 2 * q;
    *reinterpret_cast<float2*>(y + o) = make_float2(best0, best1);
    *reinterpret_cast<uchar2*>(arg + o) = make_uchar2((unsigned char)(code0 < 0 ? 0 : code0), (unsigned char)(code1 < 0 ? 0 : code1));
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

// Adjoint, four input pixels per thread (W % 4 == 0, 16-byte aligned rows): the one-pixel form is bound by workgroup
// dispatch -- 553 k workgroups of one 4-byte store per thread take 561 us on the stem's 535 MB gradient (1.25 TB/s).
// Pixels 4q .. 4q+3 of row iy are covered by the window columns 2q, 2q+1, 2q+2 (x = 4q: {2q}; 4q+1: {2q, 2q+1};
// 4q+2: {2q+1}; 4q+3: {2q+1, 2q+2}) of the window rows iy/2 (and iy/2 + 1 when iy is odd): at most 6 (gradient, byte)
// pairs, one 16-byte store.  Same sums in the same order as the one-pixel kernel (row a, then column b).
// grid (ceil(H * W/4 / 256), 1, planes), block 256
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd4_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ arg,
                                                                 float* __restrict__ gx, int H, int W, int Ho, int Wo) {
    const int W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W4) return;
    const int iy = t / W4, q = t - iy * W4;
    const size_t plane = blockIdx.z;
    const float* __restrict__ gp = gy + plane * (size_t)Ho * Wo;
    const unsigned char* __restrict__ ap = arg + plane * (size_t)Ho * Wo;
    const int oy0 = iy >> 1, oy1 = (iy & 1) ? oy0 + 1 : -1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a == 0 ? oy0 : oy1;
        if (oy < 0 || oy >= Ho) continue;
        const int ky3 = 3 * (iy - (2 * oy - 1));
        const size_t o = (size_t)oy * Wo + 2 * q;
        // window column ox covers input columns 2 ox - 1 .. 2 ox + 1, i.e. kx = ix - 2 ox + 1
        const bool c1 = 2 * q + 1 < Wo, c2 = 2 * q + 2 < Wo;
        const float g0 = gp[o], g1 = c1 ? gp[o + 1] : 0.0f, g2 = c2 ? gp[o + 2] : 0.0f;
        const int a0 = ap[o], a1 = c1 ? ap[o + 1] : 255, a2 = c2 ? ap[o + 2] : 255;
        if (a0 == ky3 + 1) acc.x += g0;                 // x = 4q     in window 2q     at kx = 1
        if (a0 == ky3 + 2) acc.y += g0;                 // x = 4q + 1 in window 2q     at kx = 2
        if (a1 == ky3 + 0) acc.y += g1;                 // x = 4q + 1 in window 2q + 1 at kx = 0
        if (a1 == ky3 + 1) acc.z += g1;                 // x = 4q + 2 in window 2q + 1 at kx = 1
        if (a1 == ky3 + 2) acc.w += g1;                 // x = 4q + 3 in window 2q + 1 at kx = 2
        if (a2 == ky3 + 0) acc.w += g2;                 // x = 4q + 3 in window 2q + 2 at kx = 0
    }
    *reinterpret_cast<float4*>(gx + plane * (size_t)H * W + (size_t)iy * W + 4 * q) = acc;
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
    // W % 4 == 0 makes Wo = W / 2 even and every row / plane offset a multiple of 16 (x), 8 (y) and 2 (arg) bytes
    const bool two = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(arg) & 1) == 0;
    for (int p0 = 0; p0 < planes; p0 += 65535) {       // grid.z limit
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        if (two)
            hipLaunchKernelGGL(maxpool3x3s2_fwd2_kernel, dim3(camli_divup(Ho * (Wo / 2), 256), 1, np), dim3(256), 0, s,
                               x + (size_t)p0 * H * W, y + (size_t)p0 * Ho * Wo, arg + (size_t)p0 * Ho * Wo, H, W, Ho, Wo);
        else
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
    const bool four = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0;
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        if (four)
            hipLaunchKernelGGL(maxpool3x3s2_bwd4_kernel, dim3(camli_divup(H * (W / 4), 256), 1, np), dim3(256), 0, s,
                               gy + (size_t)p0 * Ho * Wo, arg + (size_t)p0 * Ho * Wo, gx + (size_t)p0 * H * W, H, W, Ho, Wo);
        else
            hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(camli_divup(W, 256), H, np), dim3(256), 0, s, gy + (size_t)p0 * Ho * Wo,
                               arg + (size_t)p0 * Ho * Wo, gx + (size_t)p0 * H * W, H, W, Ho, Wo);
    }
    return camli_check_launch("camli_maxpool3x3s2_bwd");
}

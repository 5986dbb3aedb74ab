// Bilinear sampling of an image feature map at projected point positions (the 2-D -> 3-D half of the
// CLFM fusion), gfx950.
//
// Replaces grid_sample_wrapper (models/utils.py:262-269): normalise uv to [-1,1], build a [B,N,1,2]
// grid, F.grid_sample(bilinear, align_corners=True, zeros padding), drop the trailing axis.  The
// library kernel walks the channels per thread with 64-bit strided indexing (280 us per call at
// [8,128,68,120] x 2048 points); its result is detached on this path (models/clfm.py:187-190), so only
// the forward exists here.
//
//   out[b,c,n] = sum over the 4 corners of feat[b,c,yi,xi] * w,  (x,y) = un-normalise(normalise(uv[b,:,n]))
// with the normalise / un-normalise round trip kept (fp32) so the sample positions are the reference's.
//
// HBM/L2-bound gather: lanes run along the points (coalesced uv reads and output rows); the four
// corner offsets and weights are computed once per point and reused for every channel; the four waves
// of a workgroup split the channels, four channels in flight per lane.
#include "camli_common.h"

namespace {

// grid (ceil(N/64), B, CS), block 256: wave w of channel slice z handles channels 4z + w, 4z + w + 4 CS, ...
// (CS slices so that the launch carries ~4 waves per SIMD: at [8,128,68,120] x 2048 points one slice is 1,024 waves,
// each a serial chain of 32 x 4 dependent gathers)
__global__ __launch_bounds__(256) void bilinear_sample_kernel(const float* __restrict__ feat,
                                                              const float* __restrict__ uv, float* __restrict__ out,
                                                              int C, int H, int W, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane, b = blockIdx.y;
    if (n >= N) return;
    const float u = uv[((size_t)b * 2 + 0) * N + n], v = uv[((size_t)b * 2 + 1) * N + n];
    // utils.py:265-266 then grid_sampler's align_corners=True un-normalisation
    const float gx = 2.0f * u / (float)(W - 1) - 1.0f;
    const float gy = 2.0f * v / (float)(H - 1) - 1.0f;
    const float x = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float y = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx1 = x - x0f, wy1 = y - y0f;                   // weight of the +1 neighbour
    const float wx0 = (x0f + 1.0f) - x, wy0 = (y0f + 1.0f) - y;
    const float lim = 1.0e6f;                                   // keeps the int conversion defined; such taps are outside
    const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
    const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
    const float w_nw = (okx0 && oky0) ? wx0 * wy0 : 0.0f, w_ne = (okx1 && oky0) ? wx1 * wy0 : 0.0f;
    const float w_sw = (okx0 && oky1) ? wx0 * wy1 : 0.0f, w_se = (okx1 && oky1) ? wx1 * wy1 : 0.0f;
    // clamped offsets: an out-of-image tap reads a valid address and is multiplied by an exact 0
    // (a non-finite value there would give NaN where the reference skips the tap)
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
    const int o_nw = cy0 * W + cx0, o_ne = cy0 * W + cx1, o_sw = cy1 * W + cx0, o_se = cy1 * W + cx1;
    const size_t plane = (size_t)H * W;
    const float* __restrict__ src = feat + (size_t)b * C * plane;
    float* __restrict__ dst = out + (size_t)b * C * N + n;
#pragma unroll 4
    for (int c = wave + 4 * (int)blockIdx.z; c < C; c += 4 * (int)gridDim.z) {
        const float* __restrict__ p = src + (size_t)c * plane;
        float acc = p[o_nw] * w_nw;
        acc += p[o_ne] * w_ne;
        acc += p[o_sw] * w_sw;
        acc += p[o_se] * w_se;
        dst[(size_t)c * N] = acc;
    }
}

}  // namespace

extern "C" int camli_bilinear_sample_fwd(const float* feat, const float* uv, float* out, int B, int C, int H, int W,
                                         int N, void* stream) {
    if (B == 0 || N == 0 || C == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!feat || !uv || !out) {
        camli_set_error("camli_bilinear_sample_fwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C < 0 || H < 2 || W < 2 || N < 0 || B > 65535 || (long long)H * W > 0x7fffffffLL) {
        camli_set_error("camli_bilinear_sample_fwd: bad shape B=%d C=%d H=%d W=%d N=%d", B, C, H, W, N);
        return CAMLI_EINVAL;
    }
    int cs = 1;
    while (cs * 2 * 4 <= C && (long long)camli_divup(N, 64) * B * 4 * cs < 4096) cs *= 2;
    hipLaunchKernelGGL(bilinear_sample_kernel, dim3(camli_divup(N, 64), B, cs), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), feat, uv, out, C, H, W, N);
    return camli_check_launch("camli_bilinear_sample_fwd");
}

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

struct BilinearTaps {
    int o_nw, o_ne, o_sw, o_se;
    float w_nw, w_ne, w_sw, w_se;
};

// sample position + the four clamped corner offsets and (zeroed when outside) weights of one point
__device__ __forceinline__ BilinearTaps bilinear_taps(float u, float v, int H, int W) {
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
    BilinearTaps t;
    t.w_nw = (okx0 && oky0) ? wx0 * wy0 : 0.0f;
    t.w_ne = (okx1 && oky0) ? wx1 * wy0 : 0.0f;
    t.w_sw = (okx0 && oky1) ? wx0 * wy1 : 0.0f;
    t.w_se = (okx1 && oky1) ? wx1 * wy1 : 0.0f;
    // clamped offsets: an out-of-image tap reads a valid address and is multiplied by an exact 0
    // (a non-finite value there would give NaN where the reference skips the tap)
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
    t.o_nw = cy0 * W + cx0;
    t.o_ne = cy0 * W + cx1;
    t.o_sw = cy1 * W + cx0;
    t.o_se = cy1 * W + cx1;
    return t;
}

// grid (ceil(N/64), B, CS), block 256: wave w of channel slice z handles channels 4z + w, 4z + w + 4 CS, ...
// (CS slices so that the launch carries ~4 waves per SIMD: at [8,128,68,120] x 2048 points one slice is 1,024 waves,
// each a serial chain of 32 x 4 dependent gathers)
__global__ __launch_bounds__(256) void bilinear_sample_kernel(const float* __restrict__ feat,
                                                              const float* __restrict__ uv, float* __restrict__ out,
                                                              int C, int H, int W, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane, b = blockIdx.y;
    if (n >= N) return;
    const BilinearTaps t = bilinear_taps(uv[((size_t)b * 2 + 0) * N + n], uv[((size_t)b * 2 + 1) * N + n], H, W);
    const size_t plane = (size_t)H * W;
    const float* __restrict__ src = feat + (size_t)b * C * plane;
    float* __restrict__ dst = out + (size_t)b * C * N + n;
#pragma unroll 4
    for (int c = wave + 4 * (int)blockIdx.z; c < C; c += 4 * (int)gridDim.z) {
        const float* __restrict__ p = src + (size_t)c * plane;
        float acc = p[t.o_nw] * t.w_nw;
        acc += p[t.o_ne] * t.w_ne;
        acc += p[t.o_sw] * t.w_sw;
        acc += p[t.o_se] * t.w_se;
        dst[(size_t)c * N] = acc;
    }
}

// Plane-resident form (round 3): one workgroup per (b, c) copies the H x W plane into LDS with coalesced 16-byte loads
// -- the feature map is read from memory exactly once -- and its threads then take the four taps of every point from
// LDS.  The gather form above issues 4 scattered 4-byte global loads per (point, channel): 8.4 M of them at
// [8,128,68,120] x 2048 points, 68 us per call, 27 calls per step.  Same arithmetic, same results.
// grid B*C, block 256, dynamic LDS H*W floats (planes up to 64 KB).
__global__ __launch_bounds__(256) void bilinear_sample_plane_kernel(const float* __restrict__ feat,
                                                                    const float* __restrict__ uv, float* __restrict__ out,
                                                                    int C, int H, int W, int N) {
    extern __shared__ __attribute__((aligned(16))) float plane_s[];
    const int bc = blockIdx.x, b = bc / C;
    const int hw = H * W;
    const float* __restrict__ src = feat + (size_t)bc * hw;
    if ((hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        for (int e = threadIdx.x; e < (hw >> 2); e += 256)
            reinterpret_cast<float4*>(plane_s)[e] = reinterpret_cast<const float4*>(src)[e];
    } else {
        for (int e = threadIdx.x; e < hw; e += 256) plane_s[e] = src[e];
    }
    __syncthreads();
    float* __restrict__ dst = out + (size_t)bc * N;
    for (int n = threadIdx.x; n < N; n += 256) {
        const BilinearTaps t = bilinear_taps(uv[((size_t)b * 2 + 0) * N + n], uv[((size_t)b * 2 + 1) * N + n], H, W);
        float acc = plane_s[t.o_nw] * t.w_nw;
        acc += plane_s[t.o_ne] * t.w_ne;
        acc += plane_s[t.o_sw] * t.w_sw;
        acc += plane_s[t.o_se] * t.w_se;
        dst[n] = acc;
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
    const size_t plane_bytes = (size_t)H * W * sizeof(float);
    if (plane_bytes <= 64 * 1024 && (long long)B * C <= 0x7fffffffLL && N >= 256) {
        hipLaunchKernelGGL(bilinear_sample_plane_kernel, dim3(B * C), dim3(256), plane_bytes,
                           reinterpret_cast<hipStream_t>(stream), feat, uv, out, C, H, W, N);
        return camli_check_launch("camli_bilinear_sample_fwd");
    }
    int cs = 1;
    while (cs * 2 * 4 <= C && (long long)camli_divup(N, 64) * B * 4 * cs < 4096) cs *= 2;
    hipLaunchKernelGGL(bilinear_sample_kernel, dim3(camli_divup(N, 64), B, cs), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), feat, uv, out, C, H, W, N);
    return camli_check_launch("camli_bilinear_sample_fwd");
}

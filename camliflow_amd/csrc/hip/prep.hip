// Input side of the hot path, gfx950: what sits between the data loader and furthest-point
// sampling / the image encoders.
//
//   persp2paral   models/ids.py:4-33 -- perspective camera -> low-resolution parallel camera with log depth
//                 ("inverse depth scaling"): 3 divisions, a log and ~12 pointwise ops per point, composed in torch as
//                 ~20 launches per cloud; here one launch for BOTH clouds ([B,6,N] -> 2 x [B,3,N]).
//   pad_normalize models/camliraft.py:38-46 + utils.py:7-15 -- replicate-pad both frames to a multiple of 8
//                 (width split left/right, height at the bottom) and apply the ImageNet mean / std: two pads and four
//                 elementwise passes over [B,3,H,W] in torch; here one pass that reads [B,6,H,W] once and writes both
//                 padded, normalised frames.
//   project_pc2image models/utils.py:234-259 followed by the feature-grid rescale every caller applies
//                 (camliraft_core.py:51-56, camlipwc_core.py:112-114): [B,3,N] -> pixel coordinates [B,2,N] of a
//                 parallel or perspective camera, times (grid - 1) / (sensor - 1): ~8 launches in torch, here one.
// Operation ORDER follows the reference expression by expression (this file is built with -ffp-contract=off), so
// the results equal the torch composition on the same device bit for bit: FPS is a chain of 4096 arg-max decisions
// on these coordinates and must see identical inputs.
#include "camli_common.h"

namespace {

// grid (ceil(N/256), 2, B): y = which cloud
__global__ __launch_bounds__(256) void persp2paral_kernel(const float* __restrict__ pcs /*[B,6,N]*/,
                                                           const float* __restrict__ intr /*[B,3] = f, cx, cy*/,
                                                           float* __restrict__ out1, float* __restrict__ out2, int N,
                                                           float rw, float rh, float rmin, float aw, float ah) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int cloud = blockIdx.y, b = blockIdx.z;
    if (n >= N) return;
    const float f = intr[b * 3 + 0], cx = intr[b * 3 + 1], cy = intr[b * 3 + 2];
    const float* __restrict__ src = pcs + ((size_t)b * 6 + 3 * cloud) * N;
    const float x = src[n], y = src[(size_t)N + n], z = src[2 * (size_t)N + n];
    const float u = cx + (f / z) * x;                 // ids.py:16
    const float v = cy + (f / z) * y;                 // ids.py:17
    const float d = f * logf(z) + 1.0f;               // ids.py:18
    float* __restrict__ dst = (cloud == 0 ? out1 : out2) + (size_t)b * 3 * N;
    dst[n] = u * rw - aw;                             // ids.py:27-31
    dst[(size_t)N + n] = v * rh - ah;
    dst[2 * (size_t)N + n] = d * rmin;
}

// grid (ceil(Wp/256), Hp, B*6): one output row of one (frame, channel) plane per (y, z)
__global__ __launch_bounds__(256) void pad_normalize_kernel(const float* __restrict__ images /*[B,6,H,W]*/,
                                                             float* __restrict__ out1, float* __restrict__ out2, int H,
                                                             int W, int Hp, int Wp, int left, float m0, float m1, float m2,
                                                             float s0, float s1, float s2) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z / 6, c6 = blockIdx.z % 6;
    if (x >= Wp) return;
    const int c = c6 % 3;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float std = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const int sy = min(y, H - 1);                                   // bottom rows replicate the last row
    const int sx = min(max(x - left, 0), W - 1);                    // left / right columns replicate the edge
    const float v = images[(((size_t)b * 6 + c6) * H + sy) * W + sx];
    float* __restrict__ dst = (c6 < 3 ? out1 : out2) + (((size_t)b * 3 + c) * Hp + y) * Wp + x;
    *dst = (v - mean) / std;                                        // camliraft.py:45-46
}

// grid (ceil(N/256), B)
template <bool PERSPECTIVE>
__global__ __launch_bounds__(256) void project_pc2image_kernel(const float* __restrict__ pc /*[B,3,N]*/,
                                                                const float* __restrict__ intr /*[B,3] = f, cx, cy*/,
                                                                float* __restrict__ uv /*[B,2,N]*/, int N, float cx,
                                                                float cy, float sx, float sy) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const float* __restrict__ src = pc + (size_t)b * 3 * N;
    const float x = src[n], y = src[(size_t)N + n];
    float u, v;
    if (PERSPECTIVE) {
        const float f = intr[b * 3 + 0], cxb = intr[b * 3 + 1], cyb = intr[b * 3 + 2];
        const float z = src[2 * (size_t)N + n];
        u = cxb + (f / z) * x;                        // utils.py:247
        v = cyb + (f / z) * y;                        // utils.py:248
    } else {
        u = x + cx;                                   // utils.py:250-251
        v = y + cy;
    }
    float* __restrict__ dst = uv + (size_t)b * 2 * N;
    dst[n] = u * sx;                                  // camliraft_core.py:54-55
    dst[(size_t)N + n] = v * sy;
}

}  // namespace

extern "C" int camli_project_pc2image(const float* pc, const float* intrinsics, float* uv, int B, int N, int perspective,
                                      float cx, float cy, float scale_x, float scale_y, void* stream) {
    if (B == 0 || N == 0) return CAMLI_OK;
    if (!pc || !uv || (perspective && !intrinsics)) { camli_set_error("camli_project_pc2image: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || N < 0 || B > 65535) { camli_set_error("camli_project_pc2image: bad shape B=%d N=%d", B, N); return CAMLI_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (perspective)
        hipLaunchKernelGGL(project_pc2image_kernel<true>, dim3(camli_divup(N, 256), B), dim3(256), 0, s, pc, intrinsics, uv, N,
                           cx, cy, scale_x, scale_y);
    else
        hipLaunchKernelGGL(project_pc2image_kernel<false>, dim3(camli_divup(N, 256), B), dim3(256), 0, s, pc, intrinsics, uv, N,
                           cx, cy, scale_x, scale_y);
    return camli_check_launch("camli_project_pc2image");
}

extern "C" int camli_persp2paral(const float* pcs, const float* intrinsics, float* out1, float* out2, int B, int N,
                                 float ratio_w, float ratio_h, float ratio_min, float half_w, float half_h, void* stream) {
    if (B == 0 || N == 0) return CAMLI_OK;
    if (!pcs || !intrinsics || !out1 || !out2) { camli_set_error("camli_persp2paral: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || N < 0 || B > 65535) { camli_set_error("camli_persp2paral: bad shape B=%d N=%d", B, N); return CAMLI_EINVAL; }
    hipLaunchKernelGGL(persp2paral_kernel, dim3(camli_divup(N, 256), 2, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       pcs, intrinsics, out1, out2, N, ratio_w, ratio_h, ratio_min, half_w, half_h);
    return camli_check_launch("camli_persp2paral");
}

extern "C" int camli_pad_normalize(const float* images, float* out1, float* out2, int B, int H, int W, int Hp, int Wp,
                                   int pad_left, const float* mean3, const float* std3, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!images || !out1 || !out2 || !mean3 || !std3) { camli_set_error("camli_pad_normalize: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || H < 1 || W < 1 || Hp < H || Wp < W || pad_left < 0 || pad_left > Wp - W || Hp > 65535 || (long long)B * 6 > 65535) {
        camli_set_error("camli_pad_normalize: bad shape B=%d H=%d W=%d Hp=%d Wp=%d left=%d", B, H, W, Hp, Wp, pad_left);
        return CAMLI_EINVAL;
    }
    hipLaunchKernelGGL(pad_normalize_kernel, dim3(camli_divup(Wp, 256), Hp, B * 6), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), images, out1, out2, H, W, Hp, Wp, pad_left, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2]);
    return camli_check_launch("camli_pad_normalize");
}

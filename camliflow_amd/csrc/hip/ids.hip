// Scene-flow read-out in the perspective camera (inverse depth scaling), gfx950.
//
// The models predict 3-D flow in the low-resolution parallel camera; every one of the n_iters flow
// iterates is mapped back with  flow_persp = paral2persp(pc1 + flow) - paral2persp(pc1)
// (models/camliraft.py:108-110, models/ids.py:36-67).  In torch that is ~22 pointwise launches per iterate
// and twice as many in the backward, on [B,3,N] tensors of under a megabyte.  One kernel each way:
//
//   p = pc1 + flow;  u = (p0 + aw) / rw;  v = (p1 + ah) / rh;  d = p2 / rm;  z = exp((d - 1) / f)
//   out = ((u - cx) z / f, (v - cy) z / f, z) - origin
//   gflow0 = gx z / (f rw);  gflow1 = gy z / (f rh);  gflow2 = (gx (u-cx)/f + gy (v-cy)/f + gz) z / (f rm)
//
// with the operation order of ids.py kept (fp32, no contraction) so the values are the reference's.
#include "camli_common.h"

namespace {

struct IdsCam { float rw, rh, rm, aw, ah; };

template <bool BACKWARD>
__global__ __launch_bounds__(256) void ids_flow_kernel(const float* __restrict__ pc1, const float* __restrict__ flow,
                                                       const float* __restrict__ origin_or_gout,
                                                       const float* __restrict__ f_all, const float* __restrict__ cx_all,
                                                       const float* __restrict__ cy_all, float* __restrict__ out,
                                                       IdsCam cam, int N) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float f = f_all[b], cx = cx_all[b], cy = cy_all[b];
    const size_t base = (size_t)b * 3 * N + n;
    const float p0 = pc1[base] + flow[base], p1 = pc1[base + N] + flow[base + N], p2 = pc1[base + 2 * (size_t)N] + flow[base + 2 * (size_t)N];
    const float u = (p0 + cam.aw) / cam.rw, v = (p1 + cam.ah) / cam.rh, d = p2 / cam.rm;
    const float z = expf((d - 1.0f) / f);
    if (!BACKWARD) {
        out[base] = (u - cx) * z / f - origin_or_gout[base];
        out[base + N] = (v - cy) * z / f - origin_or_gout[base + N];
        out[base + 2 * (size_t)N] = z - origin_or_gout[base + 2 * (size_t)N];
    } else {
        const float gx = origin_or_gout[base], gy = origin_or_gout[base + N], gz = origin_or_gout[base + 2 * (size_t)N];
        const float zf = z / f;
        out[base] = gx * zf / cam.rw;
        out[base + N] = gy * zf / cam.rh;
        out[base + 2 * (size_t)N] = (gx * ((u - cx) / f) + gy * ((v - cy) / f) + gz) * zf / cam.rm;
    }
}

template <bool BACKWARD>
int ids_launch(const char* what, const float* pc1, const float* flow, const float* third, const float* f, const float* cx,
               const float* cy, float* out, float rw, float rh, float rm, float aw, float ah, int B, int N, void* stream) {
    if (B == 0 || N == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!pc1 || !flow || !third || !f || !cx || !cy || !out) {
        camli_set_error("%s: null pointer", what);
        return CAMLI_EINVAL;
    }
    if (B < 0 || N < 0 || B > 65535 || !(rw > 0.0f) || !(rh > 0.0f) || !(rm > 0.0f)) {
        camli_set_error("%s: bad arguments B=%d N=%d rw=%g rh=%g rm=%g", what, B, N, rw, rh, rm);
        return CAMLI_EINVAL;
    }
    IdsCam cam{rw, rh, rm, aw, ah};
    hipLaunchKernelGGL((ids_flow_kernel<BACKWARD>), dim3(camli_divup(N, 256), B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), pc1, flow, third, f, cx, cy, out, cam, N);
    return camli_check_launch(what);
}

}  // namespace

extern "C" int camli_ids_flow_fwd(const float* pc1, const float* flow, const float* origin, const float* f,
                                  const float* cx, const float* cy, float* out, float ratio_w, float ratio_h,
                                  float ratio_min, float half_w, float half_h, int B, int N, void* stream) {
    return ids_launch<false>("camli_ids_flow_fwd", pc1, flow, origin, f, cx, cy, out, ratio_w, ratio_h, ratio_min, half_w,
                             half_h, B, N, stream);
}

extern "C" int camli_ids_flow_bwd(const float* pc1, const float* flow, const float* gout, const float* f,
                                  const float* cx, const float* cy, float* gflow, float ratio_w, float ratio_h,
                                  float ratio_min, float half_w, float half_h, int B, int N, void* stream) {
    return ids_launch<true>("camli_ids_flow_bwd", pc1, flow, gout, f, cx, cy, gflow, ratio_w, ratio_h, ratio_min, half_w,
                            half_h, B, N, stream);
}

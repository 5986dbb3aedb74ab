// Elementwise halves of the separable convolutional GRU (RAFT "GRU2D"), gfx950.
//
// The reference writes one half-step as ~9 elementwise torch kernels forward and ~15 backward
// (models/raft_core.py:123-139: cat, sigmoid x2, r*h, cat, tanh, (1-z)*h + z*q).  With the context
// term hoisted (cores/raft2d.GRU2D.prepare/step) a half-step is
//     zr   = sigmoid(conv_zr([h | motion]) + ctx_zr)          z = zr[:, :C], r = zr[:, C:]
//     q    = tanh(conv_q([r*h | motion]) + ctx_q)
//     h'   = (1 - z) * h + z * q
// and the elementwise work collapses into two streaming kernels each way:
//   gates : (pre_zr, ctx_zr, h)        -> z, r*h          (r is kept for the adjoint)
//   blend : (pre_q, ctx_q, z, h)       -> q, h'
// HBM-bound, 16-byte accesses, one pass over each operand.
#include "camli_common.h"

#include <stdint.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// pre_zr, ctx_zr: [B, 2C, P];  h, z, r, rh: [B, C, P];  n4 = B*C*P/4 float4 groups; cp4 = C*P/4
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float4* __restrict__ pre, const float4* __restrict__ ctx,
                                                             const float4* __restrict__ h, float4* __restrict__ z,
                                                             float4* __restrict__ r, float4* __restrict__ rh, size_t n4,
                                                             size_t cp4) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / cp4, off = e - b * cp4;
        const size_t iz = b * 2 * cp4 + off, ir = iz + cp4;
        const float4 pz = pre[iz], cz = ctx[iz], pr = pre[ir], cr = ctx[ir], hv = h[e];
        float4 zo, ro, rho;
        zo.x = sigmoidf_(pz.x + cz.x); zo.y = sigmoidf_(pz.y + cz.y); zo.z = sigmoidf_(pz.z + cz.z); zo.w = sigmoidf_(pz.w + cz.w);
        ro.x = sigmoidf_(pr.x + cr.x); ro.y = sigmoidf_(pr.y + cr.y); ro.z = sigmoidf_(pr.z + cr.z); ro.w = sigmoidf_(pr.w + cr.w);
        rho.x = ro.x * hv.x; rho.y = ro.y * hv.y; rho.z = ro.z * hv.z; rho.w = ro.w * hv.w;
        z[e] = zo;
        r[e] = ro;
        rh[e] = rho;
    }
}

// adjoint of gates: (gz, grh, z, r, h) -> gpre [B,2C,P] (also the gradient of ctx_zr), gh [B,C,P]
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float4* __restrict__ gz, const float4* __restrict__ grh,
                                                             const float4* __restrict__ z, const float4* __restrict__ r,
                                                             const float4* __restrict__ h, float4* __restrict__ gpre,
                                                             float4* __restrict__ gh, size_t n4, size_t cp4, size_t gz_bs4,
                                                             size_t grh_bs4, int into, float4* __restrict__ gacc) {
    // gz / grh may be channel slices of wider gradients (what the adjoint of cat([rh, x]) hands over): batch strides in float4
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / cp4, off = e - b * cp4;
        const size_t iz = b * 2 * cp4 + off, ir = iz + cp4;
        const float4 g1 = gz[b * gz_bs4 + off], g2 = grh[b * grh_bs4 + off], zv = z[e], rv = r[e], hv = h[e];
        float4 a, c, d;
        a.x = g1.x * zv.x * (1.0f - zv.x); a.y = g1.y * zv.y * (1.0f - zv.y); a.z = g1.z * zv.z * (1.0f - zv.z); a.w = g1.w * zv.w * (1.0f - zv.w);
        c.x = g2.x * hv.x * rv.x * (1.0f - rv.x); c.y = g2.y * hv.y * rv.y * (1.0f - rv.y);
        c.z = g2.z * hv.z * rv.z * (1.0f - rv.z); c.w = g2.w * hv.w * rv.w * (1.0f - rv.w);
        d.x = g2.x * rv.x; d.y = g2.y * rv.y; d.z = g2.z * rv.z; d.w = g2.w * rv.w;
        gpre[iz] = a;
        gpre[ir] = c;
        if (gacc) {         // running total of gpre over the GRU iterations of a pass (= the gradient of the hoisted context term)
            float4 t = gacc[iz], u = gacc[ir];
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
            u.x += c.x; u.y += c.y; u.z += c.z; u.w += c.w;
            gacc[iz] = t;
            gacc[ir] = u;
        }
        if (into) { const float4 o = gh[e]; d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w; }
        gh[e] = d;
    }
}

// torch.nan_to_num: NaN -> 0, +-inf -> +-FLT_MAX
__device__ __forceinline__ float nan_to_num_f(float v) {
    return v != v ? 0.0f : fminf(fmaxf(v, -3.402823466e+38f), 3.402823466e+38f);
}
__device__ __forceinline__ bool finite_f(float v) { return fabsf(v) <= 3.402823466e+38f; }

// blend: q = tanh(pre_q + ctx_q); h' = (1 - z) h + z q     all [B,C,P].  SANITIZE: h' = nan_to_num(h') (raft_core.py:138,
// the GRU's last statement) folded in; the adjoint then zeroes the gradient where the un-sanitised h' was not finite.
template <bool SANITIZE>
__global__ __launch_bounds__(256) void gru_blend_fwd_kernel(const float4* __restrict__ pre, const float4* __restrict__ ctx,
                                                             const float4* __restrict__ z, const float4* __restrict__ h,
                                                             float4* __restrict__ q, float4* __restrict__ hn, size_t n4) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const float4 p = pre[e], c = ctx[e], zv = z[e], hv = h[e];
        float4 qo, ho;
        qo.x = tanhf(p.x + c.x); qo.y = tanhf(p.y + c.y); qo.z = tanhf(p.z + c.z); qo.w = tanhf(p.w + c.w);
        ho.x = (1.0f - zv.x) * hv.x + zv.x * qo.x; ho.y = (1.0f - zv.y) * hv.y + zv.y * qo.y;
        ho.z = (1.0f - zv.z) * hv.z + zv.z * qo.z; ho.w = (1.0f - zv.w) * hv.w + zv.w * qo.w;
        if (SANITIZE) { ho.x = nan_to_num_f(ho.x); ho.y = nan_to_num_f(ho.y); ho.z = nan_to_num_f(ho.z); ho.w = nan_to_num_f(ho.w); }
        q[e] = qo;
        hn[e] = ho;
    }
}

// adjoint of blend: (g, z, h, q) -> gpre (= gradient of ctx_q), gz, gh
template <bool SANITIZE>
__global__ __launch_bounds__(256) void gru_blend_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ z,
                                                             const float4* __restrict__ h, const float4* __restrict__ q,
                                                             float4* __restrict__ gpre, float4* __restrict__ gz,
                                                             float4* __restrict__ gh, size_t n4, float4* __restrict__ gacc) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        float4 gv = g[e];
        const float4 zv = z[e], hv = h[e], qv = q[e];
        if (SANITIZE) {      // nan_to_num's adjoint: no gradient where its input (recomputed here) was not finite
            gv.x = finite_f((1.0f - zv.x) * hv.x + zv.x * qv.x) ? gv.x : 0.0f;
            gv.y = finite_f((1.0f - zv.y) * hv.y + zv.y * qv.y) ? gv.y : 0.0f;
            gv.z = finite_f((1.0f - zv.z) * hv.z + zv.z * qv.z) ? gv.z : 0.0f;
            gv.w = finite_f((1.0f - zv.w) * hv.w + zv.w * qv.w) ? gv.w : 0.0f;
        }
        float4 a, b, c;
        a.x = gv.x * zv.x * (1.0f - qv.x * qv.x); a.y = gv.y * zv.y * (1.0f - qv.y * qv.y);
        a.z = gv.z * zv.z * (1.0f - qv.z * qv.z); a.w = gv.w * zv.w * (1.0f - qv.w * qv.w);
        b.x = gv.x * (qv.x - hv.x); b.y = gv.y * (qv.y - hv.y); b.z = gv.z * (qv.z - hv.z); b.w = gv.w * (qv.w - hv.w);
        c.x = gv.x * (1.0f - zv.x); c.y = gv.y * (1.0f - zv.y); c.z = gv.z * (1.0f - zv.z); c.w = gv.w * (1.0f - zv.w);
        gpre[e] = a;
        if (gacc) { float4 t = gacc[e]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; gacc[e] = t; }
        gz[e] = b;
        gh[e] = c;
    }
}

int gru_shape_ok(const char* what, int B, int C, int P) {
    if (B < 0 || C < 1 || P < 1 || ((size_t)C * P) % 4 != 0) {
        camli_set_error("%s: bad shape B=%d C=%d P=%d (C*P must be a multiple of 4)", what, B, C, P);
        return 0;
    }
    return 1;
}

int gru_blocks(size_t n4) { return (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192); }

}  // namespace

#define F4(p) reinterpret_cast<const float4*>(p)
#define F4W(p) reinterpret_cast<float4*>(p)

extern "C" int camli_gru_gates_fwd(const float* pre_zr, const float* ctx_zr, const float* h, float* z, float* r, float* rh,
                                   int B, int C, int P, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!pre_zr || !ctx_zr || !h || !z || !r || !rh) { camli_set_error("camli_gru_gates_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!gru_shape_ok("camli_gru_gates_fwd", B, C, P)) return CAMLI_EINVAL;
    const size_t cp4 = (size_t)C * P / 4, n4 = cp4 * B;
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(gru_blocks(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       F4(pre_zr), F4(ctx_zr), F4(h), F4W(z), F4W(r), F4W(rh), n4, cp4);
    return camli_check_launch("camli_gru_gates_fwd");
}

static int gates_bwd_impl(const float* gz, int64_t gz_batch_stride, const float* grh, int64_t grh_batch_stride, const float* z,
                          const float* r, const float* h, float* gpre_zr, float* gh, int B, int C, int P, int into, float* gacc,
                          void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gz || !grh || !z || !r || !h || !gpre_zr || !gh) { camli_set_error("camli_gru_gates_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!gru_shape_ok("camli_gru_gates_bwd", B, C, P)) return CAMLI_EINVAL;
    const int64_t cp = (int64_t)C * P;
    if (gz_batch_stride < cp || grh_batch_stride < cp || (gz_batch_stride & 3) || (grh_batch_stride & 3) ||
        (reinterpret_cast<uintptr_t>(gz) & 15) || (reinterpret_cast<uintptr_t>(grh) & 15)) {
        camli_set_error("camli_gru_gates_bwd: batch strides %lld / %lld must be multiples of 4 floats >= C*P = %lld, pointers 16-byte aligned",
                        (long long)gz_batch_stride, (long long)grh_batch_stride, (long long)cp);
        return CAMLI_EINVAL;
    }
    const size_t cp4 = (size_t)cp / 4, n4 = cp4 * B;
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(gru_blocks(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       F4(gz), F4(grh), F4(z), F4(r), F4(h), F4W(gpre_zr), F4W(gh), n4, cp4, (size_t)gz_batch_stride / 4,
                       (size_t)grh_batch_stride / 4, into, F4W(gacc));
    return camli_check_launch("camli_gru_gates_bwd");
}

extern "C" int camli_gru_gates_bwd_strided(const float* gz, int64_t gz_batch_stride, const float* grh, int64_t grh_batch_stride,
                                           const float* z, const float* r, const float* h, float* gpre_zr, float* gh, int B,
                                           int C, int P, void* stream) {
    return gates_bwd_impl(gz, gz_batch_stride, grh, grh_batch_stride, z, r, h, gpre_zr, gh, B, C, P, 0, nullptr, stream);
}

// gh += (see include/camli_hip.h)
extern "C" int camli_gru_gates_bwd_into(const float* gz, int64_t gz_batch_stride, const float* grh, int64_t grh_batch_stride,
                                        const float* z, const float* r, const float* h, float* gpre_zr, float* gh, float* gpre_acc,
                                        int B, int C, int P, void* stream) {
    if (gpre_acc && (reinterpret_cast<uintptr_t>(gpre_acc) & 15)) { camli_set_error("camli_gru_gates_bwd_into: gpre_acc must be 16-byte aligned"); return CAMLI_EINVAL; }
    return gates_bwd_impl(gz, gz_batch_stride, grh, grh_batch_stride, z, r, h, gpre_zr, gh, B, C, P, 1, gpre_acc, stream);
}

extern "C" int camli_gru_gates_bwd(const float* gz, const float* grh, const float* z, const float* r, const float* h,
                                   float* gpre_zr, float* gh, int B, int C, int P, void* stream) {
    return camli_gru_gates_bwd_strided(gz, (int64_t)C * P, grh, (int64_t)C * P, z, r, h, gpre_zr, gh, B, C, P, stream);
}

extern "C" int camli_gru_blend_fwd(const float* pre_q, const float* ctx_q, const float* z, const float* h, float* q,
                                   float* h_new, int B, int C, int P, int nan_to_num, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!pre_q || !ctx_q || !z || !h || !q || !h_new) { camli_set_error("camli_gru_blend_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!gru_shape_ok("camli_gru_blend_fwd", B, C, P)) return CAMLI_EINVAL;
    const size_t n4 = (size_t)B * C * P / 4;
    if (nan_to_num)
        hipLaunchKernelGGL(gru_blend_fwd_kernel<true>, dim3(gru_blocks(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           F4(pre_q), F4(ctx_q), F4(z), F4(h), F4W(q), F4W(h_new), n4);
    else
        hipLaunchKernelGGL(gru_blend_fwd_kernel<false>, dim3(gru_blocks(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           F4(pre_q), F4(ctx_q), F4(z), F4(h), F4W(q), F4W(h_new), n4);
    return camli_check_launch("camli_gru_blend_fwd");
}

static int blend_bwd_impl(const float* g, const float* z, const float* h, const float* q, float* gpre_q, float* gz, float* gh,
                          float* gacc, int B, int C, int P, int nan_to_num, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!g || !z || !h || !q || !gpre_q || !gz || !gh) { camli_set_error("camli_gru_blend_bwd: null pointer"); return CAMLI_EINVAL; }
    if (!gru_shape_ok("camli_gru_blend_bwd", B, C, P)) return CAMLI_EINVAL;
    const size_t n4 = (size_t)B * C * P / 4;
    if (nan_to_num)
        hipLaunchKernelGGL(gru_blend_bwd_kernel<true>, dim3(gru_blocks(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           F4(g), F4(z), F4(h), F4(q), F4W(gpre_q), F4W(gz), F4W(gh), n4, F4W(gacc));
    else
        hipLaunchKernelGGL(gru_blend_bwd_kernel<false>, dim3(gru_blocks(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           F4(g), F4(z), F4(h), F4(q), F4W(gpre_q), F4W(gz), F4W(gh), n4, F4W(gacc));
    return camli_check_launch("camli_gru_blend_bwd");
}

extern "C" int camli_gru_blend_bwd(const float* g, const float* z, const float* h, const float* q, float* gpre_q, float* gz,
                                   float* gh, int B, int C, int P, int nan_to_num, void* stream) {
    return blend_bwd_impl(g, z, h, q, gpre_q, gz, gh, nullptr, B, C, P, nan_to_num, stream);
}

// the same with gpre_acc += gpre_q (running total over the GRU iterations of a pass; see camli_gru_gates_bwd_into)
extern "C" int camli_gru_blend_bwd_acc(const float* g, const float* z, const float* h, const float* q, float* gpre_q, float* gz,
                                       float* gh, float* gpre_acc, int B, int C, int P, int nan_to_num, void* stream) {
    if (gpre_acc && (reinterpret_cast<uintptr_t>(gpre_acc) & 15)) { camli_set_error("camli_gru_blend_bwd_acc: gpre_acc must be 16-byte aligned"); return CAMLI_EINVAL; }
    return blend_bwd_impl(g, z, h, q, gpre_q, gz, gh, gpre_acc, B, C, P, nan_to_num, stream);
}

// Masked end-point-error sums of the sequence losses (models/losses.py:64-119, order 'l2-norm'), gfx950.
//
// Per flow iterate the reference runs  diff = pred - target[:, :C];  err = ||diff||_2 over the channels;
// loss_i = err[mask].mean()  -- nine pointwise / reduction launches per iterate and a dozen in the
// backward, over full-resolution tensors (2-D: [B,2,540,960]).  Here one kernel per iterate writes
//   sum_out[slot] += sum_{b,p : mask} sqrt(sum_c (pred - target)^2)          (mask = target[:, C] > 0 if present)
// and one kernel gives its adjoint  gpred = coef * diff / err  (0 where masked out or err == 0, as
// torch.linalg.norm's backward), with coef read from device memory (no host synchronisation).
#include "camli_common.h"

namespace {

// grid (ceil(P/1024), B), block 256, 4 positions per thread
template <int C, bool BACKWARD>
__global__ __launch_bounds__(256) void masked_l2_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        int target_channels, float* __restrict__ out /* sum slot | gpred */,
                                                        const float* __restrict__ coef, int P, int vec) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* __restrict__ pb = pred + (size_t)b * C * P;
    const float* __restrict__ tb = target + (size_t)b * target_channels * P;
    const bool has_mask = target_channels > C;
    const float k = BACKWARD ? coef[0] : 0.0f;
    float acc = 0.0f;
    // the forward walks several 1024-position chunks per workgroup: its sum ends in ONE float atomic per workgroup on a
    // single address, and 4,050 of those in a row cost more (50 us) than streaming the tensors (15 us)
    for (int chunk = blockIdx.x; chunk * 1024 < P; chunk += gridDim.x) {
    const int p0 = (chunk * 256 + threadIdx.x) * 4;
    if (vec && p0 < P) {
        // four consecutive positions per thread as 16-byte accesses (the scalar form below issues four 4-byte loads
        // per channel whose lanes are 16 bytes apart: every cache line is touched by four instructions)
        float dv[C][4], sq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(pb + (size_t)c * P + p0);
            const float4 t = *reinterpret_cast<const float4*>(tb + (size_t)c * P + p0);
            dv[c][0] = a.x - t.x; dv[c][1] = a.y - t.y; dv[c][2] = a.z - t.z; dv[c][3] = a.w - t.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) sq[j] += dv[c][j] * dv[c][j];
        }
        float mk[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (has_mask) {
            const float4 m = *reinterpret_cast<const float4*>(tb + (size_t)C * P + p0);
            mk[0] = m.x; mk[1] = m.y; mk[2] = m.z; mk[3] = m.w;
        }
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = mk[j] > 0.0f;
            const float err = sqrtf(sq[j]);
            if (!BACKWARD) acc += on ? err : 0.0f;
            sc[j] = (on && err > 0.0f) ? k / err : 0.0f;
        }
        if (BACKWARD) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                *reinterpret_cast<float4*>(out + ((size_t)b * C + c) * P + p0) =
                    make_float4(sc[0] * dv[c][0], sc[1] * dv[c][1], sc[2] * dv[c][2], sc[3] * dv[c][3]);
        }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + j;
        if (p >= P) break;
        float d[C], sq = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            d[c] = pb[(size_t)c * P + p] - tb[(size_t)c * P + p];
            sq += d[c] * d[c];
        }
        const bool on = !has_mask || tb[(size_t)C * P + p] > 0.0f;
        const float err = sqrtf(sq);
        if (!BACKWARD) {
            acc += on ? err : 0.0f;
        } else {
            const float s = (on && err > 0.0f) ? k / err : 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) out[((size_t)b * C + c) * P + p] = s * d[c];
        }
    }
    }
    }
    if (!BACKWARD) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
    }
}

template <bool BACKWARD>
int masked_l2_launch(const char* what, const float* pred, const float* target, int target_channels, float* out,
                     const float* coef, int B, int C, int P, void* stream) {
    if (B == 0 || P == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!pred || !target || !out || (BACKWARD && !coef)) {
        camli_set_error("%s: null pointer", what);
        return CAMLI_EINVAL;
    }
    if (B < 0 || P < 0 || B > 65535 || (C != 2 && C != 3) || (target_channels != C && target_channels != C + 1)) {
        camli_set_error("%s: bad shape B=%d C=%d (2 or 3) P=%d target_channels=%d", what, B, C, P, target_channels);
        return CAMLI_EINVAL;
    }
    const int chunks = camli_divup(P, 1024);
    dim3 grid(BACKWARD ? chunks : (chunks < 64 ? chunks : 64), B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 16-byte accesses need P % 4 == 0 (every channel plane then starts 16-byte aligned relative to its base) and aligned bases
    const int vec = (P & 3) == 0 && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) |
                                      (BACKWARD ? reinterpret_cast<uintptr_t>(out) : 0)) & 15) == 0;
    if (C == 2)
        hipLaunchKernelGGL((masked_l2_kernel<2, BACKWARD>), grid, dim3(256), 0, s, pred, target, target_channels, out, coef, P, vec);
    else
        hipLaunchKernelGGL((masked_l2_kernel<3, BACKWARD>), grid, dim3(256), 0, s, pred, target, target_channels, out, coef, P, vec);
    return camli_check_launch(what);
}

}  // namespace

extern "C" int camli_masked_l2_fwd(const float* pred, const float* target, int target_channels, float* sum_out, int B,
                                   int C, int P, void* stream) {
    return masked_l2_launch<false>("camli_masked_l2_fwd", pred, target, target_channels, sum_out, nullptr, B, C, P, stream);
}

extern "C" int camli_masked_l2_bwd(const float* pred, const float* target, int target_channels, const float* coef,
                                   float* gpred, int B, int C, int P, void* stream) {
    return masked_l2_launch<true>("camli_masked_l2_bwd", pred, target, target_channels, gpred, coef, B, C, P, stream);
}

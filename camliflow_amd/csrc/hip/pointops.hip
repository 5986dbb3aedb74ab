// Point gather / scatter, k-NN inverse-distance interpolation and the point cost-volume lookup
// gather, gfx950.  All are HBM/L2-bound index ops; the reference composes each from torch.gather /
// advanced indexing plus a chain of elementwise kernels, and their backward is torch's generic
// scatter_add / index_put(accumulate) with one atomic per gathered element.
//
//   gather_cf      models/utils.py:61-83   data [B,C,M], idx [B,I]          -> out [B,C,I]
//   gather_cl      models/utils.py:85-104  data [B,M,C], idx [B,I]          -> out [B,I,C]
//   knn_interp     models/utils.py:130-146 IDW of the k nearest             -> out [B,C,Nq]
//   corr3d_gather  models/camliraft_l_core.py:62-76 (dxyz, cost entry)      -> out [B,4,N,k]
//
// Layout rule everywhere: the query / output-point index is the fastest thread axis, so outputs
// are written in full 256-byte wave rows and the index row of a point is read once per thread.
#include "camli_common.h"

#include <stdint.h>
#include <stdlib.h>

namespace {

// ---- gather along the point axis, channel-first --------------------------------------------------
__global__ __launch_bounds__(256) void gather_cf_fwd_kernel(const float* __restrict__ data,
                                                             const int64_t* __restrict__ idx,
                                                             float* __restrict__ out, int C, int M, int I) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (i >= I) return;
    const int m = (int)idx[(size_t)b * I + i];
    for (int c = blockIdx.y; c < C; c += gridDim.y)
        out[((size_t)b * C + c) * I + i] = data[((size_t)b * C + c) * M + m];
}

__global__ __launch_bounds__(256) void gather_cf_bwd_kernel(const float* __restrict__ gout,
                                                             const int64_t* __restrict__ idx,
                                                             float* __restrict__ gdata, int C, int M, int I) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (i >= I) return;
    const int m = (int)idx[(size_t)b * I + i];
    for (int c = blockIdx.y; c < C; c += gridDim.y)
        unsafeAtomicAdd(gdata + ((size_t)b * C + c) * M + m, gout[((size_t)b * C + c) * I + i]);
}

// Gather-side adjoint: gdata[b,c,m] = sum over the positions i with idx[b,i] == m of gout[b,c,i], walked through the
// inverse map (CSR over (b, m): `order` holds the GLOBAL flat positions b*I + i sorted by (b, idx), `offsets` the
// segment bounds).  No atomics (the scatter form above manages 26 G atomics/s = 0.03 of the HBM roofline), every output
// written once, fixed summation order.  Lanes run along m; the 4-byte reads of one (b, c) row stay inside I*4 bytes
// (L2-resident), the writes are coalesced.  grid (ceil(M/256), min(C,64), B).
__global__ __launch_bounds__(256) void gather_cf_bwd_sorted_kernel(const float* __restrict__ gout,
                                                                    const int32_t* __restrict__ order,
                                                                    const int32_t* __restrict__ offsets,
                                                                    float* __restrict__ gdata, int C, int M, int I) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (m >= M) return;
    const int beg = offsets[(size_t)b * M + m], end = offsets[(size_t)b * M + m + 1];
    const int base = b * I;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        const float* __restrict__ row = gout + ((size_t)b * C + c) * I - base;
        float acc = 0.0f;
        for (int e = beg; e < end; ++e) acc += row[order[e]];
        gdata[((size_t)b * C + c) * M + m] = acc;
    }
}

// Same adjoint with the gout rows staged in LDS.  The reads above are 4-byte gathers inside an I*4-byte row: every one
// of them occupies a texture-path slot for a whole cache line, and that rate -- not HBM -- bounds the kernel (0.05-0.09
// of the roofline).  Here a workgroup copies TC consecutive (b, c) rows -- one contiguous span of gout -- into LDS with
// 16-byte loads, then each thread walks the inverse list of its points m against LDS: gout is read from memory exactly
// once, fully coalesced; `order` is read once per TC channels; summation order is unchanged (e ascending).
// grid (channel groups, B); dynamic LDS TC * I floats.
template <int TC>
__global__ __launch_bounds__(1024) void gather_cf_bwd_lds_kernel(const float* __restrict__ gout,
                                                                  const int32_t* __restrict__ order,
                                                                  const int32_t* __restrict__ offsets,
                                                                  float* __restrict__ gdata, int C, int M, int I, int vec) {
    extern __shared__ __attribute__((aligned(16))) float rows[];      // [TC][I]
    const int tid = threadIdx.x, nt = blockDim.x;
    const int b = blockIdx.y;
    const int base = b * I;
    for (int c0 = blockIdx.x * TC; c0 < C; c0 += gridDim.x * TC) {
        const int nc = min(TC, C - c0);
        const float* __restrict__ src = gout + ((size_t)b * C + c0) * I;
        const int total = nc * I;
        if (vec) {
            for (int t = tid * 4; t < total; t += nt * 4)
                *reinterpret_cast<float4*>(rows + t) = *reinterpret_cast<const float4*>(src + t);
        } else {
            for (int t = tid; t < total; t += nt) rows[t] = src[t];
        }
        __syncthreads();
        for (int m = tid; m < M; m += nt) {
            const int beg = offsets[(size_t)b * M + m], end = offsets[(size_t)b * M + m + 1];
            float acc[TC];
#pragma unroll
            for (int j = 0; j < TC; ++j) acc[j] = 0.0f;
            for (int e = beg; e < end; ++e) {
                const int i = order[e] - base;
#pragma unroll
                for (int j = 0; j < TC; ++j) acc[j] += rows[(j < nc ? j : 0) * I + i];
            }
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if (j < nc) gdata[((size_t)b * C + c0 + j) * M + m] = acc[j];
        }
        __syncthreads();
    }
}

template <int TC>
void launch_gather_cf_bwd_lds(const float* gout, const int32_t* order, const int32_t* offsets, float* gdata, int B, int C,
                              int M, int I, hipStream_t stream) {
    const size_t lds = (size_t)TC * I * sizeof(float);
    const int threads = lds <= 48 * 1024 ? 512 : 1024;
    const int vec = (I % 4 == 0) && ((reinterpret_cast<uintptr_t>(gout) & 15) == 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gather_cf_bwd_lds_kernel<TC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gather_cf_bwd_lds_kernel<TC>), dim3(camli_divup(C, TC), B), dim3(threads), lds, stream, gout,
                       order, offsets, gdata, C, M, I, vec);
}

// ---- gather along the point axis, channel-last (models/utils.py:85-104) -----------------------------
// data [B,M,C] -> out [B,I,C] = data[b, idx[b,i], :]: a row copy.  One thread per V consecutive channels of one output
// row (V = 4 when rows are 16-byte multiples, else 1), channels fastest: a row is read and written in whole lines, the
// int64 index of a row is one load shared by the C/V lanes that copy it.  C = 1 is the reference's rank-2 form
// (data [B,M], models/camliraft_l_core.py:70-74): coalesced writes, 4-byte gathers.
template <int V>
__global__ __launch_bounds__(256) void gather_cl_fwd_kernel(const float* __restrict__ data, const int64_t* __restrict__ idx,
                                                             float* __restrict__ out, int CV, int M, int64_t rows_b) {
    // rows_b = I * CV work items per batch entry
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= rows_b) return;
    const int64_t i = t / CV;
    const int c = (int)(t - i * CV);
    const int I = (int)(rows_b / CV);
    const int64_t m = idx[(int64_t)b * I + i];
    if (V == 4)
        reinterpret_cast<float4*>(out)[((int64_t)b * I + i) * CV + c] = reinterpret_cast<const float4*>(data)[((int64_t)b * M + m) * CV + c];
    else
        out[((int64_t)b * I + i) * CV + c] = data[((int64_t)b * M + m) * CV + c];
}

// adjoint through the inverse map (see gather_cf_bwd_sorted_kernel): gdata[b,m,:] = sum of the gout rows whose index
// is m, in ascending flat position; every output written once, no atomics.  `order` holds global row numbers b*I + i.
template <int V>
__global__ __launch_bounds__(256) void gather_cl_bwd_sorted_kernel(const float* __restrict__ gout, const int32_t* __restrict__ order,
                                                                    const int32_t* __restrict__ offsets, float* __restrict__ gdata,
                                                                    int CV, int64_t rows_b) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= rows_b) return;
    const int64_t m = t / CV;
    const int c = (int)(t - m * CV);
    const int M = (int)(rows_b / CV);
    const int beg = offsets[(int64_t)b * M + m], end = offsets[(int64_t)b * M + m + 1];
    if (V == 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = beg; e < end; ++e) {
            const float4 g = reinterpret_cast<const float4*>(gout)[(int64_t)order[e] * CV + c];
            acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        }
        reinterpret_cast<float4*>(gdata)[((int64_t)b * M + m) * CV + c] = acc;
    } else {
        float acc = 0.0f;
        for (int e = beg; e < end; ++e) acc += gout[(int64_t)order[e] * CV + c];
        gdata[((int64_t)b * M + m) * CV + c] = acc;
    }
}

// ---- inverse-distance interpolation from precomputed k nearest neighbours -------------------------
// dist_j = max(||in_xyz[:,idx_j] - q||, 1e-8); w_j = (1/dist_j) / sum_j(1/dist_j)
constexpr int KI_MAXK = 8;

template <bool BACKWARD>
__global__ __launch_bounds__(256) void knn_interp_kernel(const float* __restrict__ in_xyz,
                                                          const float* __restrict__ q_xyz,
                                                          const int64_t* __restrict__ knn, int knn_stride,
                                                          const float* __restrict__ src /* feat | gout */,
                                                          float* __restrict__ dst /* out | gfeat */, int C, int M,
                                                          int Nq, int k) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (q >= Nq) return;
    const float qx = q_xyz[((size_t)b * 3 + 0) * Nq + q];
    const float qy = q_xyz[((size_t)b * 3 + 1) * Nq + q];
    const float qz = q_xyz[((size_t)b * 3 + 2) * Nq + q];
    int m[KI_MAXK];
    float w[KI_MAXK];
    float wsum = 0.0f;
#pragma unroll
    for (int j = 0; j < KI_MAXK; ++j) {
        if (j < k) {
            m[j] = (int)knn[((size_t)b * Nq + q) * knn_stride + j];
            const float dx = in_xyz[((size_t)b * 3 + 0) * M + m[j]] - qx;
            const float dy = in_xyz[((size_t)b * 3 + 1) * M + m[j]] - qy;
            const float dz = in_xyz[((size_t)b * 3 + 2) * M + m[j]] - qz;
            const float dist = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-8f);
            w[j] = 1.0f / dist;
            wsum += w[j];
        } else {
            m[j] = 0;
            w[j] = 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < KI_MAXK; ++j) w[j] = w[j] / wsum;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        const size_t row = ((size_t)b * C + c) * M;
        if (!BACKWARD) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < KI_MAXK; ++j)
                if (j < k) acc += src[row + m[j]] * w[j];
            dst[((size_t)b * C + c) * Nq + q] = acc;
        } else {
            const float g = src[((size_t)b * C + c) * Nq + q];
#pragma unroll
            for (int j = 0; j < KI_MAXK; ++j)
                if (j < k) unsafeAtomicAdd(dst + row + m[j], g * w[j]);
        }
    }
}

// The interpolation weights alone, w[b][q][j] (the arithmetic of knn_interp_kernel), for the sorted adjoint below: the GRU loops
// interpolate between the SAME two clouds every iteration, so they are computed once per pass.
__global__ __launch_bounds__(256) void knn_interp_weights_kernel(const float* __restrict__ in_xyz,
                                                                  const float* __restrict__ q_xyz,
                                                                  const int64_t* __restrict__ knn, int knn_stride,
                                                                  float* __restrict__ wout, int M, int Nq, int k) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (q >= Nq) return;
    const float qx = q_xyz[((size_t)b * 3 + 0) * Nq + q];
    const float qy = q_xyz[((size_t)b * 3 + 1) * Nq + q];
    const float qz = q_xyz[((size_t)b * 3 + 2) * Nq + q];
    float w[KI_MAXK];
    float wsum = 0.0f;
#pragma unroll
    for (int j = 0; j < KI_MAXK; ++j) {
        w[j] = 0.0f;
        if (j < k) {
            const int m = (int)knn[((size_t)b * Nq + q) * knn_stride + j];
            const float dx = in_xyz[((size_t)b * 3 + 0) * M + m] - qx;
            const float dy = in_xyz[((size_t)b * 3 + 1) * M + m] - qy;
            const float dz = in_xyz[((size_t)b * 3 + 2) * M + m] - qz;
            const float dist = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-8f);
            w[j] = 1.0f / dist;
            wsum += w[j];
        }
    }
#pragma unroll
    for (int j = 0; j < KI_MAXK; ++j)
        if (j < k) wout[((size_t)b * Nq + q) * k + j] = w[j] / wsum;
}

// Adjoint of the interpolation wrt the features WITHOUT atomics (round 5): gfeat[b][c][m] = sum over the (query, slot) pairs
// that name m of gout[b][c][query] * w[b][query][slot].  The pairs of every m come from the inverse map of the neighbour table
// (fused.inverse_map), already resolved per pass into q_sorted[e] (the pair's query inside its sample) and w_sorted[e] (its
// weight), both in ascending (query, slot) order inside a segment: a fixed summation order, every output written once, no
// zero-fill, and neighbouring threads stream neighbouring segments.  thread = (b, m), blockIdx.y = channel.  The float-atomic
// form took 26 us for 8 x 8192 queries x 3 channels (12 contributions per target on average, all colliding).
__global__ __launch_bounds__(256) void knn_interp_bwd_sorted_kernel(const float* __restrict__ gout,
                                                                     const float* __restrict__ w_sorted,
                                                                     const int* __restrict__ q_sorted,
                                                                     const int* __restrict__ offsets,
                                                                     float* __restrict__ gfeat, int C, int M, int Nq) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (m >= M) return;
    const int beg = offsets[b * M + m], end = offsets[b * M + m + 1];
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        const float* __restrict__ grow = gout + ((size_t)b * C + c) * Nq;
        float acc = 0.0f;
        for (int e = beg; e < end; ++e) acc += grow[q_sorted[e]] * w_sorted[e];
        gfeat[((size_t)b * C + c) * M + m] = acc;
    }
}

// Coordinate adjoint of the interpolation (models/utils.py:138-146 differentiated: norm -> clamp(1e-8) ->
// reciprocal -> normalise -> weighted sum).  With p_j = w_j / W the normalised weights, a_j = sum_c g_c f_cj and
// S = sum_c g_c out_c:   dL/dw_j = (a_j - S) / W,   dw_j/ddist_j = -w_j^2 (where the clamp is inactive),
// ddist_j/dx_j = (x_j - q) / dist_j  (0 at dist_j = 0, as torch.linalg.norm's backward defines it).
// thread = query; the k neighbour rows are re-read per channel (L2-resident, C <= ~200), g_q is written
// directly, the input-cloud gradient by one float atomic per (neighbour, axis).
__global__ __launch_bounds__(256) void knn_interp_bwd_xyz_kernel(const float* __restrict__ in_xyz,
                                                                  const float* __restrict__ feat,
                                                                  const float* __restrict__ gout,
                                                                  const float* __restrict__ q_xyz,
                                                                  const int64_t* __restrict__ knn, int knn_stride,
                                                                  float* __restrict__ g_in, float* __restrict__ g_q,
                                                                  int C, int M, int Nq, int k) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= Nq) return;
    const float qx = q_xyz[((size_t)b * 3 + 0) * Nq + q];
    const float qy = q_xyz[((size_t)b * 3 + 1) * Nq + q];
    const float qz = q_xyz[((size_t)b * 3 + 2) * Nq + q];
    int m[KI_MAXK];
    float w[KI_MAXK], raw[KI_MAXK], dx[KI_MAXK], dy[KI_MAXK], dz[KI_MAXK], a[KI_MAXK];
    float wsum = 0.0f;
#pragma unroll
    for (int j = 0; j < KI_MAXK; ++j) {
        a[j] = 0.0f;
        if (j < k) {
            m[j] = (int)knn[((size_t)b * Nq + q) * knn_stride + j];
            dx[j] = in_xyz[((size_t)b * 3 + 0) * M + m[j]] - qx;
            dy[j] = in_xyz[((size_t)b * 3 + 1) * M + m[j]] - qy;
            dz[j] = in_xyz[((size_t)b * 3 + 2) * M + m[j]] - qz;
            raw[j] = sqrtf(dx[j] * dx[j] + dy[j] * dy[j] + dz[j] * dz[j]);
            w[j] = 1.0f / fmaxf(raw[j], 1e-8f);
            wsum += w[j];
        } else {
            m[j] = 0;
            w[j] = raw[j] = dx[j] = dy[j] = dz[j] = 0.0f;
        }
    }
    float s_tot = 0.0f;
    for (int c = 0; c < C; ++c) {
        const size_t row = ((size_t)b * C + c) * M;
        const float g = gout[((size_t)b * C + c) * Nq + q];
        float o = 0.0f;
#pragma unroll
        for (int j = 0; j < KI_MAXK; ++j)
            if (j < k) {
                const float f = feat[row + m[j]];
                o += f * (w[j] / wsum);
                a[j] += g * f;
            }
        s_tot += g * o;
    }
    float gqx = 0.0f, gqy = 0.0f, gqz = 0.0f;
#pragma unroll
    for (int j = 0; j < KI_MAXK; ++j)
        if (j < k && raw[j] >= 1e-8f) {
            const float gd = -((a[j] - s_tot) / wsum) * (w[j] * w[j]);     // dL/ddist_j
            const float sx = gd * (dx[j] / raw[j]), sy = gd * (dy[j] / raw[j]), sz = gd * (dz[j] / raw[j]);
            if (g_in) {
                unsafeAtomicAdd(g_in + ((size_t)b * 3 + 0) * M + m[j], sx);
                unsafeAtomicAdd(g_in + ((size_t)b * 3 + 1) * M + m[j], sy);
                unsafeAtomicAdd(g_in + ((size_t)b * 3 + 2) * M + m[j], sz);
            }
            gqx -= sx; gqy -= sy; gqz -= sz;
        }
    if (g_q) {
        g_q[((size_t)b * 3 + 0) * Nq + q] = gqx;
        g_q[((size_t)b * 3 + 1) * Nq + q] = gqy;
        g_q[((size_t)b * 3 + 2) * Nq + q] = gqz;
    }
}

// ---- point cost-volume lookup gather ---------------------------------------------------------------
// out[b,0:3,n,j] = xyz2[b,:,idx[b,n,j]] - xyz1[b,:,n];  out[b,3,n,j] = cost[b,n,idx[b,n,j]]
// thread = (b, n, j), j fastest (k contiguous outputs per point)
template <bool BACKWARD>
__global__ __launch_bounds__(256) void corr3d_gather_kernel(const float* __restrict__ xyz1,
                                                             const float* __restrict__ xyz2,
                                                             float* __restrict__ cost /* cost | gcost */,
                                                             const int64_t* __restrict__ knn,
                                                             float* __restrict__ io /* out | gout */, int B, int N,
                                                             int M, int k) {
    const size_t total = (size_t)B * N * k;
    const size_t plane = (size_t)N * k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t bn = e / k;
        const int b = (int)(bn / N);
        const int n = (int)(bn - (size_t)b * N);
        const int m = (int)knn[e];
        const size_t o = (size_t)b * 4 * plane + (e - (size_t)b * plane);
        if (!BACKWARD) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                io[o + a * plane] = xyz2[((size_t)b * 3 + a) * M + m] - xyz1[((size_t)b * 3 + a) * N + n];
            io[o + 3 * plane] = cost[bn * M + m];
        } else {
            // gcost is zero-filled and (b, n) rows belong to one thread group of k: a neighbour index that occurs once
            // (every row when M >= k) is a plain store.  Duplicates only exist when M < k (the search pads with index
            // 0): the FIRST occurrence in the row sums all of them in j order -- no atomics, bit-reproducible.
            const int j = (int)(e - bn * k);
            const int64_t* __restrict__ row = knn + bn * k;
            bool first = true;
            for (int jj = 0; jj < j; ++jj) first = first && ((int)row[jj] != m);
            if (first) {
                float acc = io[o + 3 * plane];
                for (int jj = j + 1; jj < k; ++jj)
                    if ((int)row[jj] == m) acc += io[o + 3 * plane + (jj - j)];
                cost[bn * M + m] += acc;
            }
        }
    }
}

// All pyramid levels of the lookup in one launch (nested target prefixes: level l holds the first sizes[l] points of
// xyz2, so one coordinate tensor serves every level).  out [B,4,N,L*k]: column l*k + j = neighbour j of level l -- the
// concatenated tensor the shared cost MLP consumes, written directly (the reference runs 4 x (2 gathers + cat)).
// The adjoint ADDS into persistent per-level gradient volumes: a point's k neighbours are distinct, launches on one
// stream are ordered, so a plain read-modify-write replaces the atomics AND the per-iteration zero-filled temporaries
// that autograd would then have to sum over the GRU iterations.
struct Corr3dLevels {
    float* cost[4];            // [B,N,size[l]]  (fwd: read; bwd: gradient volume, accumulated)
    const int64_t* knn[4];     // [B,N,k]
    int size[4];
    int levels;
};

template <bool BACKWARD>
__global__ __launch_bounds__(256) void corr3d_gather_levels_kernel(const float* __restrict__ xyz1,
                                                                    const float* __restrict__ xyz2, Corr3dLevels lv,
                                                                    float* __restrict__ io, int B, int N, int M0, int k) {
    const int lk = lv.levels * k;
    const size_t plane = (size_t)N * lk;
    const size_t total = (size_t)B * plane;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(e % lk);
        const size_t bn = e / lk;
        const int b = (int)(bn / N), n = (int)(bn - (size_t)b * N);
        const int l = col / k, j = col - l * k;
        const int m = (int)lv.knn[l][bn * k + j];
        const size_t o = (size_t)b * 4 * plane + (e - (size_t)b * plane);
        if (!BACKWARD) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                io[o + a * plane] = xyz2[((size_t)b * 3 + a) * M0 + m] - xyz1[((size_t)b * 3 + a) * N + n];
            io[o + 3 * plane] = lv.cost[l][bn * lv.size[l] + m];
        } else {
            float* g = lv.cost[l] + bn * lv.size[l] + m;
            *g += io[o + 3 * plane];
        }
    }
}

int grid_y_for(int C) { return C < 64 ? C : 64; }

// out[b,c,p] = scale[b,c,p] * data[b,c,idx[b,p]]  -- the nearest-point feature of every pixel times its score
// (FusionAwareInterp with k = 1, models/clfm.py:70-76: batch_indexing, multiply, sum over the singleton k axis:
// three launches and two [B,C,HW] intermediates).  The same kernel is its own adjoint wrt `scale`
// (gscale = gout * data[idx]); `data` is detached on this path (clfm.py:186).
// grid (ceil(P/256), ceil(C/4), B), block 256: lanes along p, four channels per thread share the index.
__global__ __launch_bounds__(256) void gather_scale_kernel(const float* __restrict__ data, const float* __restrict__ scale,
                                                           const int64_t* __restrict__ idx, float* __restrict__ out,
                                                           int C, int M, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
    if (p >= P) return;
    const int m = (int)idx[(size_t)b * P + p];
    const int c0 = blockIdx.y * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;
        if (c < C) {
            const size_t row = (size_t)b * C + c;
            out[row * P + p] = scale[row * P + p] * data[row * M + m];
        }
    }
}

// Row-resident form (round 3): a workgroup owns TWO (b, c) rows of `data`, keeps them in LDS (2 * M floats) and walks all
// P pixels: the index is read once per two channels, the gathers hit LDS instead of scattered 4-byte global loads
// (8.4 M per call at [8,128] x 8160 pixels, 31.6 us per call, 54 calls per step), scale / out stream coalesced.
// grid (ceil(C/2), B, PS pixel slices), block 256, dynamic LDS 2*M floats; four consecutive pixels per thread (16-byte
// scale loads / out stores) when P % 4 == 0.
__global__ __launch_bounds__(256) void gather_scale_rows_kernel(const float* __restrict__ data, const float* __restrict__ scale,
                                                                const int64_t* __restrict__ idx, float* __restrict__ out,
                                                                int C, int M, int P) {
    extern __shared__ float rows_s[];      // [2][M]
    const int c0 = blockIdx.x * 2, b = blockIdx.y;
    const bool two = c0 + 1 < C;
    const size_t r0 = (size_t)b * C + c0, r1 = r0 + (two ? 1 : 0);
    for (int e = threadIdx.x; e < M; e += 256) {
        rows_s[e] = data[r0 * M + e];
        rows_s[M + e] = data[r1 * M + e];
    }
    __syncthreads();
    const int64_t* __restrict__ ib = idx + (size_t)b * P;
    const int per = (P + gridDim.z - 1) / gridDim.z;
    const int lo = ((blockIdx.z * per) + 3) & ~3;                         // slice bounds on multiples of 4
    const int hi = min(P, (((blockIdx.z + 1) * per) + 3) & ~3);
    if ((P & 3) == 0) {
        for (int p = lo + 4 * threadIdx.x; p < hi; p += 4 * 256) {
            const longlong2 i01 = *reinterpret_cast<const longlong2*>(ib + p), i23 = *reinterpret_cast<const longlong2*>(ib + p + 2);
            const int m0 = (int)i01.x, m1 = (int)i01.y, m2 = (int)i23.x, m3 = (int)i23.y;
            const float4 s0 = *reinterpret_cast<const float4*>(scale + r0 * P + p);
            float4 o0;
            o0.x = s0.x * rows_s[m0]; o0.y = s0.y * rows_s[m1]; o0.z = s0.z * rows_s[m2]; o0.w = s0.w * rows_s[m3];
            *reinterpret_cast<float4*>(out + r0 * P + p) = o0;
            if (two) {
                const float4 s1 = *reinterpret_cast<const float4*>(scale + r1 * P + p);
                float4 o1;
                o1.x = s1.x * rows_s[M + m0]; o1.y = s1.y * rows_s[M + m1]; o1.z = s1.z * rows_s[M + m2]; o1.w = s1.w * rows_s[M + m3];
                *reinterpret_cast<float4*>(out + r1 * P + p) = o1;
            }
        }
    } else {
        for (int p = lo + threadIdx.x; p < hi; p += 256) {
            const int m = (int)ib[p];
            out[r0 * P + p] = scale[r0 * P + p] * rows_s[m];
            if (two) out[r1 * P + p] = scale[r1 * P + p] * rows_s[M + m];
        }
    }
}

}  // namespace

extern "C" int camli_gather_cf_fwd(const float* data, const int64_t* idx, float* out, int B, int C, int M, int I,
                                   void* stream) {
    if (B == 0 || I == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!data || !idx || !out) { camli_set_error("camli_gather_cf_fwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || I < 0 || B > 65535) {
        camli_set_error("camli_gather_cf_fwd: bad shape B=%d C=%d M=%d I=%d", B, C, M, I);
        return CAMLI_EINVAL;
    }
    if (B == 0 || I == 0) return CAMLI_OK;
    hipLaunchKernelGGL(gather_cf_fwd_kernel, dim3(camli_divup(I, 256), grid_y_for(C), B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), data, idx, out, C, M, I);
    return camli_check_launch("camli_gather_cf_fwd");
}

extern "C" int camli_gather_cf_bwd(const float* gout, const int64_t* idx, float* gdata, int B, int C, int M, int I,
                                   void* stream) {
    if (B == 0 || I == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gout || !idx || !gdata) { camli_set_error("camli_gather_cf_bwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || I < 0 || B > 65535) {
        camli_set_error("camli_gather_cf_bwd: bad shape B=%d C=%d M=%d I=%d", B, C, M, I);
        return CAMLI_EINVAL;
    }
    if (B == 0 || I == 0) return CAMLI_OK;
    hipLaunchKernelGGL(gather_cf_bwd_kernel, dim3(camli_divup(I, 256), grid_y_for(C), B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gout, idx, gdata, C, M, I);
    return camli_check_launch("camli_gather_cf_bwd");
}

extern "C" int camli_gather_cf_bwd_sorted(const float* gout, const int32_t* inv_order, const int32_t* inv_offsets,
                                          float* gdata, int B, int C, int M, int I, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gout || !inv_order || !inv_offsets || !gdata) { camli_set_error("camli_gather_cf_bwd_sorted: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || I < 0 || B > 65535 || (int64_t)B * I > 2147483647LL) {
        camli_set_error("camli_gather_cf_bwd_sorted: bad shape B=%d C=%d M=%d I=%d", B, C, M, I);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // rows in LDS when they fit: as many channels per pass as stay within 48 KiB (3 workgroups per CU), at least one
    // within 128 KiB; longer rows (I > 32768) keep the direct-gather form
    static const int use_lds = [] { const char* e = getenv("CAMLI_GATHER_BWD_LDS"); return e ? atoi(e) : 1; }();
    const size_t row = (size_t)I * sizeof(float);
    if (use_lds && I > 0 && row <= 128 * 1024) {
        if (8 * row <= 48 * 1024 && C >= 8) launch_gather_cf_bwd_lds<8>(gout, inv_order, inv_offsets, gdata, B, C, M, I, s);
        else if (4 * row <= 48 * 1024 && C >= 4) launch_gather_cf_bwd_lds<4>(gout, inv_order, inv_offsets, gdata, B, C, M, I, s);
        else if (2 * row <= 48 * 1024 && C >= 2) launch_gather_cf_bwd_lds<2>(gout, inv_order, inv_offsets, gdata, B, C, M, I, s);
        else launch_gather_cf_bwd_lds<1>(gout, inv_order, inv_offsets, gdata, B, C, M, I, s);
    } else {
        hipLaunchKernelGGL(gather_cf_bwd_sorted_kernel, dim3(camli_divup(M, 256), grid_y_for(C), B), dim3(256), 0, s, gout,
                           inv_order, inv_offsets, gdata, C, M, I);
    }
    return camli_check_launch("camli_gather_cf_bwd_sorted");
}

extern "C" int camli_gather_cl_fwd(const float* data, const int64_t* idx, float* out, int B, int C, int M, int I,
                                   void* stream) {
    if (B == 0 || I == 0) return CAMLI_OK;
    if (!data || !idx || !out) { camli_set_error("camli_gather_cl_fwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || I < 0 || B > 65535 || (int64_t)I * C > ((int64_t)1 << 38)) {
        camli_set_error("camli_gather_cl_fwd: bad shape B=%d C=%d M=%d I=%d", B, C, M, I);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int cv = vec ? C / 4 : C;
    const int64_t rows_b = (int64_t)I * cv;
    const dim3 grid((unsigned)((rows_b + 255) / 256), B);
    if (vec) hipLaunchKernelGGL(gather_cl_fwd_kernel<4>, grid, dim3(256), 0, s, data, idx, out, cv, M, rows_b);
    else hipLaunchKernelGGL(gather_cl_fwd_kernel<1>, grid, dim3(256), 0, s, data, idx, out, cv, M, rows_b);
    return camli_check_launch("camli_gather_cl_fwd");
}

extern "C" int camli_gather_cl_bwd_sorted(const float* gout, const int32_t* inv_order, const int32_t* inv_offsets,
                                          float* gdata, int B, int C, int M, int I, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gout || !inv_order || !inv_offsets || !gdata) { camli_set_error("camli_gather_cl_bwd_sorted: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || I < 0 || B > 65535 || (int64_t)B * I > 2147483647LL || (int64_t)M * C > ((int64_t)1 << 38)) {
        camli_set_error("camli_gather_cl_bwd_sorted: bad shape B=%d C=%d M=%d I=%d", B, C, M, I);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(gdata)) & 15) == 0;
    const int cv = vec ? C / 4 : C;
    const int64_t rows_b = (int64_t)M * cv;
    const dim3 grid((unsigned)((rows_b + 255) / 256), B);
    if (vec) hipLaunchKernelGGL(gather_cl_bwd_sorted_kernel<4>, grid, dim3(256), 0, s, gout, inv_order, inv_offsets, gdata, cv, rows_b);
    else hipLaunchKernelGGL(gather_cl_bwd_sorted_kernel<1>, grid, dim3(256), 0, s, gout, inv_order, inv_offsets, gdata, cv, rows_b);
    return camli_check_launch("camli_gather_cl_bwd_sorted");
}

static int knn_interp_args_ok(const char* what, const void* a, const void* b, const void* c, const void* d,
                              const void* e, int B, int C, int M, int Nq, int k, int knn_stride) {
    if (!a || !b || !c || !d || !e) { camli_set_error("%s: null pointer", what); return 0; }
    if (B < 0 || C < 1 || M < 1 || Nq < 0 || k < 1 || k > KI_MAXK || knn_stride < k || B > 65535) {
        camli_set_error("%s: bad shape B=%d C=%d M=%d Nq=%d k=%d (k <= %d)", what, B, C, M, Nq, k, KI_MAXK);
        return 0;
    }
    return 1;
}

extern "C" int camli_knn_interp_fwd(const float* in_xyz, const float* feat, const float* q_xyz, const int64_t* knn,
                                    int knn_stride, float* out, int B, int C, int M, int Nq, int k, void* stream) {
    if (B == 0 || Nq == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!knn_interp_args_ok("camli_knn_interp_fwd", in_xyz, feat, q_xyz, knn, out, B, C, M, Nq, k, knn_stride))
        return CAMLI_EINVAL;
    if (B == 0 || Nq == 0) return CAMLI_OK;
    hipLaunchKernelGGL((knn_interp_kernel<false>), dim3(camli_divup(Nq, 256), C < 16 ? 1 : grid_y_for(C / 4), B),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in_xyz, q_xyz, knn, knn_stride, feat, out, C,
                       M, Nq, k);
    return camli_check_launch("camli_knn_interp_fwd");
}

extern "C" int camli_knn_interp_bwd(const float* in_xyz, const float* gout, const float* q_xyz, const int64_t* knn,
                                    int knn_stride, float* gfeat, int B, int C, int M, int Nq, int k, void* stream) {
    if (B == 0 || Nq == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!knn_interp_args_ok("camli_knn_interp_bwd", in_xyz, gout, q_xyz, knn, gfeat, B, C, M, Nq, k, knn_stride))
        return CAMLI_EINVAL;
    if (B == 0 || Nq == 0) return CAMLI_OK;
    hipLaunchKernelGGL((knn_interp_kernel<true>), dim3(camli_divup(Nq, 256), C < 16 ? 1 : grid_y_for(C / 4), B),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in_xyz, q_xyz, knn, knn_stride, gout, gfeat,
                       C, M, Nq, k);
    return camli_check_launch("camli_knn_interp_bwd");
}

extern "C" int camli_knn_interp_weights(const float* in_xyz, const float* q_xyz, const int64_t* knn, int knn_stride,
                                        float* w_out, int B, int M, int Nq, int k, void* stream) {
    if (B == 0 || Nq == 0) return CAMLI_OK;
    if (!in_xyz || !q_xyz || !knn || !w_out) { camli_set_error("camli_knn_interp_weights: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || M < 1 || Nq < 0 || k < 1 || k > KI_MAXK || knn_stride < k || B > 65535) {
        camli_set_error("camli_knn_interp_weights: bad shape B=%d M=%d Nq=%d k=%d (k <= %d)", B, M, Nq, k, KI_MAXK);
        return CAMLI_EINVAL;
    }
    hipLaunchKernelGGL(knn_interp_weights_kernel, dim3(camli_divup(Nq, 256), 1, B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in_xyz, q_xyz, knn, knn_stride, w_out, M, Nq, k);
    return camli_check_launch("camli_knn_interp_weights");
}

extern "C" int camli_knn_interp_bwd_sorted(const float* gout, const float* w_sorted, const int* q_sorted, const int* offsets,
                                           float* gfeat, int B, int C, int M, int Nq, void* stream) {
    if (B == 0 || M == 0) return CAMLI_OK;
    if (!gout || !w_sorted || !q_sorted || !offsets || !gfeat) { camli_set_error("camli_knn_interp_bwd_sorted: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || Nq < 0 || B > 65535) {
        camli_set_error("camli_knn_interp_bwd_sorted: bad shape B=%d C=%d M=%d Nq=%d", B, C, M, Nq);
        return CAMLI_EINVAL;
    }
    hipLaunchKernelGGL(knn_interp_bwd_sorted_kernel, dim3(camli_divup(M, 256), grid_y_for(C), B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gout, w_sorted, q_sorted, offsets, gfeat, C, M, Nq);
    return camli_check_launch("camli_knn_interp_bwd_sorted");
}

extern "C" int camli_knn_interp_bwd_xyz(const float* in_xyz, const float* feat, const float* gout, const float* q_xyz,
                                        const int64_t* knn, int knn_stride, float* g_in_xyz, float* g_q_xyz, int B, int C,
                                        int M, int Nq, int k, void* stream) {
    if (B == 0 || Nq == 0) return CAMLI_OK;
    if (!knn_interp_args_ok("camli_knn_interp_bwd_xyz", in_xyz, feat, q_xyz, knn, gout, B, C, M, Nq, k, knn_stride))
        return CAMLI_EINVAL;
    if (!g_in_xyz && !g_q_xyz) return CAMLI_OK;
    hipLaunchKernelGGL(knn_interp_bwd_xyz_kernel, dim3(camli_divup(Nq, 256), B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in_xyz, feat, gout, q_xyz, knn, knn_stride, g_in_xyz,
                       g_q_xyz, C, M, Nq, k);
    return camli_check_launch("camli_knn_interp_bwd_xyz");
}

extern "C" int camli_corr3d_gather_fwd(const float* xyz1, const float* xyz2, const float* cost, const int64_t* knn,
                                       float* out, int B, int N, int M, int k, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!xyz1 || !xyz2 || !cost || !knn || !out) { camli_set_error("camli_corr3d_gather_fwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || N < 1 || M < 1 || k < 1) {
        camli_set_error("camli_corr3d_gather_fwd: bad shape B=%d N=%d M=%d k=%d", B, N, M, k);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    const size_t total = (size_t)B * N * k;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((corr3d_gather_kernel<false>), dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       xyz1, xyz2, const_cast<float*>(cost), knn, out, B, N, M, k);
    return camli_check_launch("camli_corr3d_gather_fwd");
}

extern "C" int camli_corr3d_gather_bwd(const float* gout, const int64_t* knn, float* gcost, int B, int N, int M, int k,
                                       void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!gout || !knn || !gcost) { camli_set_error("camli_corr3d_gather_bwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || N < 1 || M < 1 || k < 1) {
        camli_set_error("camli_corr3d_gather_bwd: bad shape B=%d N=%d M=%d k=%d", B, N, M, k);
        return CAMLI_EINVAL;
    }
    if (B == 0) return CAMLI_OK;
    const size_t total = (size_t)B * N * k;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((corr3d_gather_kernel<true>), dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       nullptr, nullptr, gcost, knn, const_cast<float*>(gout), B, N, M, k);
    return camli_check_launch("camli_corr3d_gather_bwd");
}

static int corr3d_levels_pack(const char* what, Corr3dLevels& lv, float* const* cost, const int64_t* const* knn,
                              const int* sizes, int L, int B, int N, int M0, int k) {
    if (!cost || !knn || !sizes) { camli_set_error("%s: null pointer", what); return 0; }
    if (L < 1 || L > 4 || B < 0 || N < 1 || M0 < 1 || k < 1) {
        camli_set_error("%s: bad shape L=%d B=%d N=%d M0=%d k=%d", what, L, B, N, M0, k);
        return 0;
    }
    lv.levels = L;
    for (int l = 0; l < 4; ++l) {
        lv.cost[l] = l < L ? cost[l] : nullptr;
        lv.knn[l] = l < L ? knn[l] : nullptr;
        lv.size[l] = l < L ? sizes[l] : 0;
        if (l < L && (!cost[l] || !knn[l] || sizes[l] < 1 || sizes[l] > M0)) {
            camli_set_error("%s: level %d: null pointer or size %d outside [1, %d]", what, l, l < L ? sizes[l] : 0, M0);
            return 0;
        }
    }
    return 1;
}

extern "C" int camli_corr3d_gather_levels_fwd(const float* xyz1, const float* xyz2, const float* const* cost_levels,
                                              const int64_t* const* knn_levels, const int* sizes, int L, float* out, int B,
                                              int N, int M0, int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!xyz1 || !xyz2 || !out) { camli_set_error("camli_corr3d_gather_levels_fwd: null pointer"); return CAMLI_EINVAL; }
    Corr3dLevels lv;
    if (!corr3d_levels_pack("camli_corr3d_gather_levels_fwd", lv, const_cast<float* const*>(cost_levels), knn_levels, sizes, L, B, N,
                            M0, k))
        return CAMLI_EINVAL;
    const size_t total = (size_t)B * N * L * k;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((corr3d_gather_levels_kernel<false>), dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       xyz1, xyz2, lv, out, B, N, M0, k);
    return camli_check_launch("camli_corr3d_gather_levels_fwd");
}

extern "C" int camli_corr3d_gather_levels_bwd(const float* gout, const int64_t* const* knn_levels, float* const* gcost_levels,
                                              const int* sizes, int L, int B, int N, int M0, int k, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gout) { camli_set_error("camli_corr3d_gather_levels_bwd: null pointer"); return CAMLI_EINVAL; }
    Corr3dLevels lv;
    if (!corr3d_levels_pack("camli_corr3d_gather_levels_bwd", lv, gcost_levels, knn_levels, sizes, L, B, N, M0, k)) return CAMLI_EINVAL;
    for (int l = 0; l < L; ++l)
        if (sizes[l] < k) {      // fewer candidates than k: the unfilled slots repeat index 0 and would race
            camli_set_error("camli_corr3d_gather_levels_bwd: level %d has %d < k = %d points", l, sizes[l], k);
            return CAMLI_ENOTSUP;
        }
    const size_t total = (size_t)B * N * L * k;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((corr3d_gather_levels_kernel<true>), dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       nullptr, nullptr, lv, const_cast<float*>(gout), B, N, M0, k);
    return camli_check_launch("camli_corr3d_gather_levels_bwd");
}

extern "C" int camli_gather_scale_fwd(const float* data, const float* scale, const int64_t* idx, float* out, int B, int C,
                                      int M, int P, void* stream) {
    if (B == 0 || P == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    if (!data || !scale || !idx || !out) { camli_set_error("camli_gather_scale_fwd: null pointer"); return CAMLI_EINVAL; }
    if (B < 0 || C < 1 || M < 1 || P < 0 || B > 65535 || C > 4 * 65535) {
        camli_set_error("camli_gather_scale_fwd: bad shape B=%d C=%d M=%d P=%d", B, C, M, P);
        return CAMLI_EINVAL;
    }
    if ((size_t)2 * M * sizeof(float) <= 64 * 1024 && P >= 1024) {
        int ps = 1;
        while ((long long)camli_divup(C, 2) * B * ps < 2048 && P / (ps * 2) >= 1024) ps *= 2;
        hipLaunchKernelGGL(gather_scale_rows_kernel, dim3(camli_divup(C, 2), B, ps), dim3(256), (size_t)2 * M * sizeof(float),
                           reinterpret_cast<hipStream_t>(stream), data, scale, idx, out, C, M, P);
        return camli_check_launch("camli_gather_scale_fwd");
    }
    hipLaunchKernelGGL(gather_scale_kernel, dim3(camli_divup(P, 256), camli_divup(C, 4), B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), data, scale, idx, out, C, M, P);
    return camli_check_launch("camli_gather_scale_fwd");
}

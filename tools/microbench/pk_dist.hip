// Microbenchmark: the unfused 3-D squared distance (3 sub, 3 mul, 2 add) on 32 register-resident candidates per lane,
// scalar fp32 VALU vs packed (v_pk_add_f32 / v_pk_mul_f32 on slot pairs).  Question: do the packed forms issue at the
// scalar rate (half the instructions for the same arithmetic) when the 16 slot pairs are independent?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench/pk_dist.hip -o tools/microbench/pk_dist && tools/microbench/pk_dist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void k(const float* __restrict__ cand, const float* __restrict__ qry, float* out, int nq) {
    const int lane = threadIdx.x;
    float cx[32], cy[32], cz[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        cx[j] = cand[(j * 256 + lane) * 3];
        cy[j] = cand[(j * 256 + lane) * 3 + 1];
        cz[j] = cand[(j * 256 + lane) * 3 + 2];
    }
    float acc = 3.4e38f;
    for (int q = 0; q < nq; ++q) {
        const float ux = qry[q * 3], uy = qry[q * 3 + 1], uz = qry[q * 3 + 2];   // uniform -> scalar loads
        if (PK) {
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                float2v x = {cx[j], cx[j + 1]}, y = {cy[j], cy[j + 1]}, z = {cz[j], cz[j + 1]};
                float2v u = {ux, ux}, v = {uy, uy}, w = {uz, uz};
                float2v dx = u - x, dy = v - y, dz = w - z;
                float2v d = dx * dx + dy * dy;
                d = d + dz * dz;
                acc = fminf(acc, fminf(d.x, d.y));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float dx = ux - cx[j], dy = uy - cy[j], dz = uz - cz[j];
                float d = dx * dx + dy * dy;
                d = d + dz * dz;
                acc = fminf(acc, d);
            }
        }
    }
    out[blockIdx.x * 256 + lane] = acc;
}

int main() {
    const int nq = 2048, blocks = 256 * 8;
    std::vector<float> hc(32 * 256 * 3), hq(nq * 3);
    for (auto& v : hc) v = rand() / (float)RAND_MAX * 10;
    for (auto& v : hq) v = rand() / (float)RAND_MAX * 10;
    float *dc, *dq, *dout;
    hipMalloc(&dc, hc.size() * 4); hipMalloc(&dq, hq.size() * 4); hipMalloc(&dout, blocks * 256 * 4);
    hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> r0(blocks * 256), r1(blocks * 256);
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (pk) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, dc, dq, dout, nq);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, dc, dq, dout, nq);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double pairs = (double)blocks * 256 * 32 * nq;
            printf("%s: %.3f ms, %.1f Gpairs/s, %.2f cycles per 64 pairs per SIMD at 2.4 GHz\n", pk ? "packed" : "scalar", ms,
                   pairs / ms * 1e-6, ms * 1e-3 * 2.4e9 / (pairs / 64 / 1024));
        }
        hipMemcpy((pk ? r1 : r0).data(), dout, blocks * 256 * 4, hipMemcpyDeviceToHost);
    }
    int diff = 0;
    for (size_t i = 0; i < r0.size(); ++i) diff += r0[i] != r1[i];
    printf("mismatching results: %d\n", diff);
    return 0;
}

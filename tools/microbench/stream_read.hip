// Calibration for the set-conv forward: what does a READ-ONLY stream deliver on MI355X, as a function
// of tensor size, load width, workgroup shape and access order?
//   (a) flat: grid-stride 16-byte loads over the whole tensor
//   (b) rows: the k-major set-conv order -- workgroup (tile of 512 points, batch b, channel slice z) walks channels
//       c = z, z+CS, ...; per channel every lane reads K dword rows [j][n] spaced N floats apart, next channel prefetched
// Both sum what they read and write one float per thread at the end (so nothing is optimised away).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_read.hip -o tools/microbench/stream_read && tools/microbench/stream_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void flat_read(const float4* __restrict__ src, size_t n4, float* __restrict__ out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int K, int NWV>
__global__ __launch_bounds__(64 * NWV) void rows_read(const float* __restrict__ w, float* __restrict__ out, int C, int N) {
    const int n = blockIdx.x * 64 * NWV + threadIdx.x, b = blockIdx.y;
    const int cstep = gridDim.z;
    float cur[K], nxt[K];
    float acc = 0.0f;
    int c = blockIdx.z;
    {
        const float* __restrict__ s = w + ((size_t)b * C + c) * K * (size_t)N + n;
#pragma unroll
        for (int j = 0; j < K; ++j) cur[j] = s[(size_t)j * N];
    }
    for (; c < C; c += cstep) {
        if (c + cstep < C) {
            const float* __restrict__ s = w + ((size_t)b * C + c + cstep) * K * (size_t)N + n;
#pragma unroll
            for (int j = 0; j < K; ++j) nxt[j] = s[(size_t)j * N];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) acc += cur[j];
#pragma unroll
        for (int j = 0; j < K; ++j) cur[j] = nxt[j];
    }
    out[(((size_t)b * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 64 * NWV + threadIdx.x] = acc;
}

// the k-major set-conv forward rebuilt step by step on top of rows_read: STAGE = feature row of the channel staged per
// workgroup in a double-buffered LDS row + one barrier per channel; GATHER = the K products read the row at the lane's
// neighbour indices (max + arg-max); STORE: 0 none, 1 out only, 2 out + arg + wsel + msel
template <int K, int NWV, bool STAGE, bool GATHER, int STORE>
__global__ __launch_bounds__(64 * NWV) void dw_steps(const float* __restrict__ w, const float* __restrict__ feat,
                                                     const int* __restrict__ idx, float* __restrict__ out,
                                                     unsigned char* __restrict__ arg, float* __restrict__ wsel,
                                                     int* __restrict__ msel, int C, int N) {
    extern __shared__ float rows[];
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 64 * NWV + tid, b = blockIdx.y;
    const int cstep = gridDim.z;
    int m[K];
#pragma unroll
    for (int j = 0; j < K; ++j) m[j] = GATHER ? idx[((size_t)b * K + j) * N + n] : tid;
    float cur[K], nxt[K];
    float4 rst = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = blockIdx.z, buf = 0;
    {
        const float* __restrict__ s = w + ((size_t)b * C + c) * K * (size_t)N + n;
#pragma unroll
        for (int j = 0; j < K; ++j) cur[j] = s[(size_t)j * N];
        if (STAGE) rst = reinterpret_cast<const float4*>(feat + ((size_t)b * C + c) * N)[tid];
    }
    float total = 0.0f;
    for (; c < C; c += cstep) {
        float* rowbuf = rows + buf * N;
        if (STAGE) *reinterpret_cast<float4*>(rowbuf + tid * 4) = rst;
        if (c + cstep < C) {
            const float* __restrict__ s = w + ((size_t)b * C + c + cstep) * K * (size_t)N + n;
#pragma unroll
            for (int j = 0; j < K; ++j) nxt[j] = s[(size_t)j * N];
            if (STAGE) rst = reinterpret_cast<const float4*>(feat + ((size_t)b * C + c + cstep) * N)[tid];
        }
        if (STAGE) __syncthreads();
        float best = -1e30f;
        int barg = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float p = (STAGE ? rowbuf[GATHER ? m[j] : tid] : 1.0f) * cur[j];
            const bool gt = p > best;
            best = gt ? p : best;
            barg = gt ? j : barg;
        }
        const size_t o = ((size_t)b * C + c) * N + n;
        if (STORE >= 1) out[o] = best;
        if (STORE >= 2) {
            arg[o] = (unsigned char)barg;
            float ws = cur[0];
            int ms = m[0];
#pragma unroll
            for (int j = 1; j < K; ++j) { ws = barg == j ? cur[j] : ws; ms = barg == j ? m[j] : ms; }
            wsel[o] = ws;
            msel[o] = ms;
        }
        total += best;
#pragma unroll
        for (int j = 0; j < K; ++j) cur[j] = nxt[j];
        buf ^= 1;
    }
    if (STORE == 0) out[(((size_t)b * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 64 * NWV + tid] = total;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float* out; hipMalloc(&out, 64 << 20);
    const size_t sizes_mb[] = {32, 128, 256, 1024, 4096};
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20;
        float* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
        for (int blocks : {1024, 4096, 16384}) {
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(flat_read, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
            hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(flat_read, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            const float ms = time_ms(e0, e1) / 10;
            printf("flat_read   %5zu MB  %5d blocks  %8.1f us  %7.1f GB/s\n", mb, blocks, ms * 1e3, bytes / ms / 1e6);
        }
        hipFree(buf);
    }
    // set-conv order: B = 8, C = 128, N = 2048
    const int B = 8, C = 128, N = 2048;
#define ROWS(KK, CS)                                                                                                   \
    {                                                                                                                  \
        const size_t bytes = (size_t)B * C * KK * N * 4;                                                               \
        float* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);                                                  \
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((rows_read<KK, 8>), dim3(N / 512, B, CS), dim3(512), 0, 0, buf, out, C, N); \
        hipEventRecord(e0);                                                                                            \
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((rows_read<KK, 8>), dim3(N / 512, B, CS), dim3(512), 0, 0, buf, out, C, N); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                                   \
        const float ms = time_ms(e0, e1) / 10;                                                                         \
        printf("rows_read   k=%2d  %4zu MB  slices %2d  %8.1f us  %7.1f GB/s\n", KK, bytes >> 20, CS, ms * 1e3, bytes / ms / 1e6); \
        hipFree(buf);                                                                                                  \
    }
    ROWS(4, 32) ROWS(16, 8) ROWS(16, 16) ROWS(16, 32) ROWS(16, 64) ROWS(32, 16) ROWS(32, 32)
    {
        const int K = 16, CS = 32;
        const size_t wbytes = (size_t)B * C * K * N * 4, obytes = (size_t)B * C * N * 4;
        float *wbuf, *feat, *o1, *ws; int *idx, *ms; unsigned char* ar;
        hipMalloc(&wbuf, wbytes); hipMemset(wbuf, 0, wbytes);
        hipMalloc(&feat, obytes); hipMemset(feat, 0, obytes);
        hipMalloc(&o1, obytes); hipMalloc(&ws, obytes); hipMalloc(&ms, obytes); hipMalloc(&ar, obytes / 4);
        hipMalloc(&idx, (size_t)B * K * N * 4);
        int* hidx = (int*)malloc((size_t)B * K * N * 4);
        for (size_t i = 0; i < (size_t)B * K * N; ++i) hidx[i] = rand() % N;
        hipMemcpy(idx, hidx, (size_t)B * K * N * 4, hipMemcpyHostToDevice);
#define STEP(ST, GA, SO, LABEL)                                                                                            \
        {                                                                                                                  \
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((dw_steps<16, 8, ST, GA, SO>), dim3(N / 512, B, CS), dim3(512), 2 * N * 4, 0, wbuf, feat, idx, o1, ar, ws, ms, C, N); \
            hipEventRecord(e0);                                                                                            \
            for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((dw_steps<16, 8, ST, GA, SO>), dim3(N / 512, B, CS), dim3(512), 2 * N * 4, 0, wbuf, feat, idx, o1, ar, ws, ms, C, N); \
            hipEventRecord(e1); hipEventSynchronize(e1);                                                                   \
            printf("dw_steps k=16 slices %d  %-44s %8.1f us\n", CS, LABEL, time_ms(e0, e1) / 10 * 1e3);                     \
        }
        STEP(false, false, 0, "weights only + max")
        STEP(true, false, 0, "+ staged row, barrier per channel")
        STEP(true, true, 0, "+ gathers through the neighbour table")
        STEP(true, true, 1, "+ store out")
        STEP(true, true, 2, "+ store arg, wsel, msel")
        STEP(false, false, 2, "weights only + all four stores")
        // the same full kernel with the caches flushed between launches (a 1 GB read in between, events around each launch):
        // back-to-back repetitions above re-read a 128 MB tensor that fits the 256 MB Infinity Cache
        {
            float* big; hipMalloc(&big, (size_t)1 << 30); hipMemset(big, 0, (size_t)1 << 30);
            float tot_full = 0.f, tot_w = 0.f;
            for (int r = 0; r < 12; ++r) {
                hipLaunchKernelGGL(flat_read, dim3(4096), dim3(256), 0, 0, (const float4*)big, ((size_t)1 << 30) / 16, out);
                hipEventRecord(e0);
                hipLaunchKernelGGL((dw_steps<16, 8, true, true, 2>), dim3(N / 512, B, CS), dim3(512), 2 * N * 4, 0, wbuf, feat, idx, o1, ar, ws, ms, C, N);
                hipEventRecord(e1); hipEventSynchronize(e1);
                if (r >= 2) tot_full += time_ms(e0, e1);
                hipLaunchKernelGGL(flat_read, dim3(4096), dim3(256), 0, 0, (const float4*)big, ((size_t)1 << 30) / 16, out);
                hipEventRecord(e0);
                hipLaunchKernelGGL((dw_steps<16, 8, false, false, 0>), dim3(N / 512, B, CS), dim3(512), 2 * N * 4, 0, wbuf, feat, idx, o1, ar, ws, ms, C, N);
                hipEventRecord(e1); hipEventSynchronize(e1);
                if (r >= 2) tot_w += time_ms(e0, e1);
            }
            printf("dw_steps k=16 slices %d  full kernel, caches flushed before every launch      %8.1f us\n", CS, tot_full / 10 * 1e3);
            printf("dw_steps k=16 slices %d  weights only + max, caches flushed                   %8.1f us\n", CS, tot_w / 10 * 1e3);
        }
    }
    return 0;
}

// Stand-alone probe of stream fork / join patterns inside hipStreamBeginCapture ... hipStreamEndCapture (round-4 finding: a
// captured training step with the CLFM chain forked onto an auxiliary stream dies with SIGSEGV inside hipStreamEndCapture,
// profiles/r04_branch_capture_bisect.txt).  One pattern per process (a crash must not hide the others):
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/capture_fork_join.hip -o tools/microbench/bin/capture_fork_join
//   for v in 0 1 2 3 4 5; do tools/microbench/bin/capture_fork_join $v; echo "variant $v -> exit $?"; done
//   0  A forks B, B joins A                                   (plain fork / join: must capture)
//   1  A forks B, B forks C, C joins B, B joins A             (nested, joined in order)
//   2  A forks B, B forks C, C joins A, B joins A             (nested fork joined to the ORIGIN, not to its parent)
//   3  A forks B and C, a kernel on C waits for an event of B, C joins A, B joins A   (cross edge between two forks)
//   4  A forks B, B never joins                               (must fail with hipErrorStreamCaptureUnjoined, not crash)
//   5  A forks B, B forks C, C joins A, B never joins         (unjoined parent of a joined child)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorName(e_)); fflush(stdout); if (fatal) exit(2); } } while (0)
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }
static void fork(hipStream_t from, hipStream_t to, hipEvent_t e, bool fatal) { CK(hipEventRecord(e, from)); CK(hipStreamWaitEvent(to, e, 0)); }
int main(int argc, char** argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    bool fatal = true;
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    hipEvent_t e[6];
    for (auto& x : e) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    float* d; CK(hipMalloc(&d, 3 * 256)); CK(hipMemset(d, 0, 3 * 256));
    printf("variant %d\n", v); fflush(stdout);
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, A, d);
    fork(A, B, e[0], fatal);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, B, d + 64);
    if (v == 1 || v == 2 || v == 5) { fork(B, C, e[1], fatal); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, C, d + 128); }
    if (v == 3) { fork(A, C, e[1], fatal); fork(B, C, e[2], fatal); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, C, d + 128); }
    if (v == 1) fork(C, B, e[3], fatal);                                   // C joins its parent
    if (v == 2 || v == 3 || v == 5) fork(C, A, e[3], fatal);              // C joins the origin
    if (v != 4 && v != 5) fork(B, A, e[4], fatal);                         // B joins A
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, A, d);
    hipGraph_t g = nullptr;
    fatal = false;
    printf("  ending capture ...\n"); fflush(stdout);
    hipError_t rc = hipStreamEndCapture(A, &g);
    printf("  hipStreamEndCapture -> %s, graph %p\n", hipGetErrorName(rc), (void*)g); fflush(stdout);
    if (rc == hipSuccess && g) {
        hipGraphExec_t x; CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0)); CK(hipGraphLaunch(x, A)); CK(hipStreamSynchronize(A));
        float h[192]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("  replayed: A %.0f B %.0f C %.0f\n", h[0], h[64], h[128]);
    }
    return 0;
}

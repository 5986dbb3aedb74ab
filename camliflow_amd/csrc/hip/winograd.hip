// C-ABI of the Winograd F(2x2,3x3) convolution (winograd.h + gemm_w128.h): the 3x3 convolutions of the RAFT update block
// (models/raft_core.py:148-151 MotionEncoder2D.conv_c2 / conv, :173 FlowHead2D.conv1, :188 the mask head) forward and data
// gradient, replacing the library's fp32 Winograd / implicit-GEMM kernels (0.70-0.75 of the fp32 matrix rate counted as a
// direct convolution; this path: 1.05-1.10, profiles/r06a_winograd_microbench.txt).
#include "camli_common.h"
#include "gemm_w128.h"
#include "winograd.h"

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        cus = n >= 8 ? n / 8 * 8 : 8;
    }
    return cus;
}

constexpr int KS = 16, NBUF = 3;

// the 16 plane GEMMs Mo[t] = U[t]^T V[t]: one launch, batch = 16
template <int GA, int GB, int WM>
int launch_planes(const float* U, const float* V, float* Mo, int Mp, int NT, int K, hipStream_t s) {
    constexpr size_t lds = (size_t)NBUF * KS * 512 * sizeof(float);
    auto kern = &w128::gemm_w128_kernel<KS, NBUF, 0, GA, GB, WM>;
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            camli_set_error("camli_wino_conv3x3: cannot reserve %zu bytes of LDS", lds);
            return CAMLI_ELAUNCH;
        }
        set = true;
    }
    w128::Problem p;
    p.A = U; p.B = V; p.C = Mo; p.M = Mp; p.N = NT; p.K = K;
    p.lda = Mp; p.ldb = NT; p.ldc = NT;
    p.sa = (int64_t)K * Mp; p.sb = (int64_t)K * NT; p.sc = (int64_t)Mp * NT;
    p.alpha = 1.0f;
    p.tiles_m = camli_divup(Mp, w128::tile_m<GA, WM>());
    p.tiles_n = camli_divup(NT, w128::tile_n<GB, WM>());
    p.tiles = 16 * p.tiles_m * p.tiles_n;
    const int cus = cu_count();
    // the kernel's tile order deals 8 chunks (one per XCD) of gridDim.x / 8 consecutive tiles per round
    int nwg = p.tiles < cus ? (p.tiles + 7) / 8 * 8 : cus;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, p);
    return CAMLI_OK;
}

int mp_of(int M) { return (M + 3) & ~3; }
int kp_of(int K) { return (K + KS - 1) / KS * KS; }

}  // namespace

extern "C" int64_t camli_wino_weight_floats(int K, int M) { return K < 1 || M < 1 ? 0 : (int64_t)16 * kp_of(K) * mp_of(M); }

// U [16][Kp][Mp] from w [Cout][Cin][3][3]: flip = 0: K = Cin, M = Cout (forward); flip = 1: K = Cout, M = Cin, taps reversed
// (the data gradient's weights).  Kp = K rounded up to a multiple of 16, Mp = M to a multiple of 4, zeros beyond K / M.
extern "C" int camli_wino_weights(const float* w, float* U, int Cout, int Cin, int flip, void* stream) {
    const char* what = "camli_wino_weights";
    if (!w || !U) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (Cout < 1 || Cin < 1) { camli_set_error("%s: bad shape %d x %d", what, Cout, Cin); return CAMLI_EINVAL; }
    const int Kp = kp_of(flip ? Cout : Cin), Mp = mp_of(flip ? Cin : Cout);
    hipLaunchKernelGGL(wino::weight_transform_kernel, dim3(camli_divup(Kp * Mp, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, U,
                       Cout, Cin, Kp, Mp, flip ? 1 : 0);
    return camli_check_launch(what);
}

extern "C" int64_t camli_wino_workspace_bytes(int B, int C, int N, int H, int W) {
    if (B < 1 || C < 1 || N < 1 || H < 1 || W < 1) return 0;
    const wino::Geometry g = wino::make_geometry(B, H, W);
    return (int64_t)16 * g.NT * ((int64_t)kp_of(C) + mp_of(N)) * (int64_t)sizeof(float);
}

// y (= or +=) act(conv3x3(x) + bias).  x: B images of C planes H x W, image b at x + b * x_bs (planes dense: a channel slice
// of a wider NCHW tensor is fine); mask (optional, same geometry, image stride mask_bs): x reads as zero where mask <= 0;
// U = camli_wino_weights(...) [16][C][Mp]; y: N planes per image, image stride y_bs.
extern "C" int camli_wino_conv3x3(const float* x, int64_t x_bs, const float* mask, int64_t mask_bs, const float* U, const float* bias,
                                  float* y, int64_t y_bs, float* workspace, int64_t workspace_bytes, int B, int C, int N, int H, int W,
                                  int act, int accumulate, void* stream) {
    if (B == 0) return CAMLI_OK;
    const char* what = "camli_wino_conv3x3";
    if (!x || !U || !y || !workspace) { camli_set_error("%s: null pointer", what); return CAMLI_EINVAL; }
    if (B < 0 || H < 1 || W < 1 || C <= (NBUF - 1) * KS || N < 4 || (act != wino::OUT_PLAIN && act != wino::OUT_RELU)) {
        camli_set_error("%s: unsupported shape B=%d C=%d N=%d %dx%d act=%d (more than %d input channels; act 0 | 1)", what, B, C, N, H, W, act,
                        (NBUF - 1) * KS);
        return CAMLI_ENOTSUP;
    }
    const wino::Geometry g = wino::make_geometry(B, H, W);
    const int Mp = mp_of(N), Cp = kp_of(C);
    const int64_t need = camli_wino_workspace_bytes(B, C, N, H, W);
    if (workspace_bytes < need) { camli_set_error("%s: workspace of %lld bytes, %lld needed", what, (long long)workspace_bytes, (long long)need); return CAMLI_EINVAL; }
    if ((int64_t)(Mp + 256) * g.NT * 4 >= (int64_t)0x7FF00000 || (int64_t)g.NT * 16 >= ((int64_t)1 << 30)) {
        camli_set_error("%s: a transform-domain plane beyond 2 GB (B=%d N=%d %dx%d)", what, B, N, H, W);
        return CAMLI_ENOTSUP;
    }
    if (!aligned16(U) || !aligned16(workspace)) { camli_set_error("%s: U / workspace must be 16-byte aligned", what); return CAMLI_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* V = workspace;
    float* Mo = workspace + (size_t)16 * Cp * g.NT;
    const int64_t plane = (int64_t)H * W;
    const dim3 block(256);
    {
        const bool vec = W % 4 == 0 && aligned16(x) && x_bs % 4 == 0 && (!mask || (aligned16(mask) && mask_bs % 4 == 0));
        const dim3 grid(camli_divup(g.NT / 4, 256), Cp);
        if (vec) hipLaunchKernelGGL(wino::input_transform_kernel<true>, grid, block, 0, s, x, x_bs, plane, mask, mask_bs, plane, V, C, g);
        else hipLaunchKernelGGL(wino::input_transform_kernel<false>, grid, block, 0, s, x, x_bs, plane, mask, mask_bs, plane, V, C, g);
    }
    int rc;
    if (Mp <= 128) rc = launch_planes<2, 1, 1>(U, V, Mo, Mp, g.NT, Cp, s);
    else if (Mp <= 192) rc = launch_planes<3, 1, 1>(U, V, Mo, Mp, g.NT, Cp, s);
    else rc = launch_planes<2, 2, 2>(U, V, Mo, Mp, g.NT, Cp, s);
    if (rc != CAMLI_OK) return rc;
    {
        const bool vec = W % 4 == 0 && aligned16(y) && y_bs % 4 == 0;
        const dim3 grid(camli_divup(g.NT / 4, 256), N);
        if (vec) hipLaunchKernelGGL(wino::output_transform_kernel<true>, grid, block, 0, s, Mo, Mp, bias, y, y_bs, plane, act, accumulate ? 1 : 0, g);
        else hipLaunchKernelGGL(wino::output_transform_kernel<false>, grid, block, 0, s, Mo, Mp, bias, y, y_bs, plane, act, accumulate ? 1 : 0, g);
    }
    return camli_check_launch(what);
}

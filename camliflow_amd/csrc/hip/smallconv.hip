// Small-channel 3x3 convolution heads, gfx950 (the 2-D update block).
//
// The last convolution of every flow head maps a wide feature map to TWO channels (FlowHead2D.conv2: 256 -> 2,
// models/raft_core.py:169-181; PWC's conv_last, models/pwc_core.py).  As a library convolution that is an
// implicit GEMM with N = 2: at batch 8, 68x120 MIOpen needs 89 us forward and 182 us backward plus ~30 us of layout
// transposes for 0.6 GFLOP (6.7 TFLOP/s, profiles/r03_conv_layout_microbench.txt), although the op only has to stream
// the 67 MB input once per direction.  Three HBM-bound kernels replace it, bias included (fp32, NCHW, stride 1, pad 1):
//
//   fwd        y[b,co,y,x]  = bias[co] + sum_{ci,dy,dx} w[co,ci,dy,dx] * x[b,ci,y+dy-1,x+dx-1]
//              lane = pixel: a wave covers 62 consecutive x of one row plus one halo column on either side, so a row of
//              an input channel is ONE coalesced load and the x-1 / x+1 taps are wave shifts (v_mov_dpp wave_shr/shl);
//              the 4 waves of a workgroup split the input channels and are summed through LDS in a fixed order; the
//              weights (2*Cin*9 floats) sit in LDS and are read as broadcasts
//   bwd_data   gx[b,ci,y,x] = sum_{co,dy,dx} w[co,ci,dy,dx] * gy[b,co,y-dy+1,x-dx+1]
//              lane = pixel keeps its 18 gradient taps in registers, one coalesced store per input channel
//   bwd_weight gw[co,ci,dy,dx] = sum_{b,y,x} gy[b,co,y,x] * x[b,ci,y+dy-1,x+dx-1],  gb[co] = sum gy
//              wave = (input channel, 64-pixel column strip, batch element) walking down the rows with a sliding
//              3-row window: 18 register accumulators per lane, ONE cross-lane reduction at the end, per-wave partial
//              sums reduced by a second kernel in a fixed order (no atomics: bit-reproducible).
// Zero padding, no dilation / groups.  Algorithmic bytes: 4*B*H*W*(Cin + 2) per direction (+ the weights).
#include "camli_common.h"

namespace {

constexpr int SC_CO = 2;
constexpr int SC_TILE = 62;      // output pixels per wave: lanes 1..62; lanes 0 and 63 carry the left / right halo column

// neighbour columns without extra loads: lane i takes the value of lane i-1 / i+1 (gfx9 wave shifts); the outermost
// lanes get 0, which only the halo lanes (no output) ever consume
__device__ __forceinline__ float from_left(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x138, 0xf, 0xf, false));   // wave_shr:1
}
__device__ __forceinline__ float from_right(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x130, 0xf, 0xf, false));   // wave_shl:1
}

constexpr int SC_WPAD = 12;      // LDS weight block per (co, ci): 9 taps padded to 12 floats = three 16-byte broadcast reads

__device__ __forceinline__ void stage_weights(float* __restrict__ sw, const float* __restrict__ w, int Cin) {
    for (int e = threadIdx.x; e < SC_CO * Cin * SC_WPAD; e += 256) {
        const int blk = e / SC_WPAD, t = e - blk * SC_WPAD;
        sw[e] = t < 9 ? w[blk * 9 + t] : 0.0f;
    }
}

// A workgroup = 62 columns x TWO output rows (four input rows serve both; a 544-workgroup launch is one round on 256
// CUs -- one row per workgroup was 1088 x 4 waves = 1.06 rounds, i.e. twice the wave lifetime).  Loads use a uniform
// 64-bit base (advanced per input channel on the scalar unit) plus a fixed 32-bit lane offset: no vector address math
// in the loop.  grid (ceil(W/62), ceil(H/2), B), block 256.  dynamic LDS: 2*Cin*12 floats + 4*4*64 (partial sums)
__global__ __launch_bounds__(256) void conv3x3_co2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int Cin, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sw = lds;                                   // [2][Cin][12]
    float* part = lds + SC_CO * Cin * SC_WPAD;         // [4 waves][2 rows][2 co][64]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = blockIdx.x * SC_TILE + lane - 1, py = blockIdx.y * 2, b = blockIdx.z;
    stage_weights(sw, w, Cin);
    __syncthreads();
    const unsigned plane = (unsigned)H * W;
    const bool col_ok = px >= 0 && px < W;
    // input rows py-1 .. py+2; a row outside the image is skipped (wave-uniform) and its register stays 0
    bool rok[4];
    unsigned roff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ry = py + r - 1;
        rok[r] = ry >= 0 && ry < H;
        roff[r] = (unsigned)(rok[r] ? ry : 0) * W + (unsigned)(col_ok ? px : 0);
    }
    float acc[2][SC_CO] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    constexpr int U = 4;                               // input channels whose 4 rows are requested together
    for (int c0 = wv; c0 < Cin; c0 += 4 * U) {
        float v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ci = c0 + 4 * u;                 // wave-uniform
            const float* __restrict__ base = x + ((size_t)b * Cin + (ci < Cin ? ci : c0)) * plane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = 0.0f;
                if (rok[r] && ci < Cin) t = base[roff[r]];
                v[u][r] = col_ok ? t : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ci = c0 + 4 * u;
            if (ci >= Cin) break;
            const float4* __restrict__ q0 = reinterpret_cast<const float4*>(sw + ci * SC_WPAD);
            const float4* __restrict__ q1 = reinterpret_cast<const float4*>(sw + (Cin + ci) * SC_WPAD);
            const float4 a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
            const float w0[9] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x};
            const float w1[9] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c = v[u][r];
                const float l = from_left(c), rr = from_right(c);
#pragma unroll
                for (int o = 0; o < 2; ++o) {           // output row o uses input row r as its tap row dy = r - o
                    const int dy = r - o;
                    if (dy < 0 || dy > 2) continue;
                    acc[o][0] = __builtin_fmaf(l, w0[dy * 3 + 0], acc[o][0]);
                    acc[o][0] = __builtin_fmaf(c, w0[dy * 3 + 1], acc[o][0]);
                    acc[o][0] = __builtin_fmaf(rr, w0[dy * 3 + 2], acc[o][0]);
                    acc[o][1] = __builtin_fmaf(l, w1[dy * 3 + 0], acc[o][1]);
                    acc[o][1] = __builtin_fmaf(c, w1[dy * 3 + 1], acc[o][1]);
                    acc[o][1] = __builtin_fmaf(rr, w1[dy * 3 + 2], acc[o][1]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int co = 0; co < SC_CO; ++co) part[((wv * 2 + o) * SC_CO + co) * 64 + lane] = acc[o][co];
    __syncthreads();
    if (wv < 2 && col_ok && lane >= 1 && lane <= SC_TILE && py + wv < H) {      // wave 0 finishes row py, wave 1 row py + 1
        const int o = wv;
#pragma unroll
        for (int co = 0; co < SC_CO; ++co) {
            const float s = ((part[((0 * 2 + o) * SC_CO + co) * 64 + lane] + part[((1 * 2 + o) * SC_CO + co) * 64 + lane]) +
                             (part[((2 * 2 + o) * SC_CO + co) * 64 + lane] + part[((3 * 2 + o) * SC_CO + co) * 64 + lane])) +
                            (bias ? bias[co] : 0.0f);
            y[((size_t)b * SC_CO + co) * plane + (size_t)(py + o) * W + px] = s;
        }
    }
}

// grid (ceil(W/62), ceil(H/2), B), block 256.  dynamic LDS: 2*Cin*12 floats
__global__ __launch_bounds__(256) void conv3x3_co2_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                    float* __restrict__ gx, int Cin, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sw = lds;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = blockIdx.x * SC_TILE + lane - 1, py = blockIdx.y * 2, b = blockIdx.z;
    stage_weights(sw, w, Cin);
    __syncthreads();
    const unsigned plane = (unsigned)H * W;
    const bool col_ok = px >= 0 && px < W;
    // gradient rows py-1 .. py+2 of both channels with their left / right neighbours: t[co][r][0..2] = columns px-1, px, px+1
    float t[SC_CO][4][3];
#pragma unroll
    for (int co = 0; co < SC_CO; ++co) {
        const float* __restrict__ gp = gy + ((size_t)b * SC_CO + co) * plane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ry = py + r - 1;
            const bool ok = ry >= 0 && ry < H;          // wave-uniform
            float c = 0.0f;
            if (ok) c = gp[(unsigned)ry * W + (unsigned)(col_ok ? px : 0)];
            c = col_ok ? c : 0.0f;
            t[co][r][0] = from_left(c);
            t[co][r][1] = c;
            t[co][r][2] = from_right(c);
        }
    }
    if (!(col_ok && lane >= 1 && lane <= SC_TILE)) return;
    const bool second = py + 1 < H;
    float* __restrict__ dst = gx + (size_t)b * Cin * plane + (size_t)py * W + px;
    for (int ci = wv; ci < Cin; ci += 4) {
        const float4* __restrict__ q0 = reinterpret_cast<const float4*>(sw + ci * SC_WPAD);
        const float4* __restrict__ q1 = reinterpret_cast<const float4*>(sw + (Cin + ci) * SC_WPAD);
        const float4 a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
        const float wq[SC_CO][9] = {{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x},
                                    {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x}};
        float o0 = 0.0f, o1 = 0.0f;
        // gx[y][x] = sum w[co][dy][dx] * gy[co][y - dy + 1][x - dx + 1]: output row o reads gradient row index o + 2 - dy
#pragma unroll
        for (int co = 0; co < SC_CO; ++co)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    o0 = __builtin_fmaf(t[co][0 + 2 - dy][2 - dx], wq[co][dy * 3 + dx], o0);
                    o1 = __builtin_fmaf(t[co][1 + 2 - dy][2 - dx], wq[co][dy * 3 + dx], o1);
                }
        dst[(size_t)ci * plane] = o0;
        if (second) dst[(size_t)ci * plane + W] = o1;
    }
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// grid (ceil(W/62), ceil(Cin/4), B), block 256: wave = one input channel.  partials [B*strips][2*Cin*9 + 2]
__global__ __launch_bounds__(256) void conv3x3_co2_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                      float* __restrict__ partials, int Cin, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci = blockIdx.y * 4 + wv;
    const int px = blockIdx.x * SC_TILE + lane - 1, b = blockIdx.z;
    if (ci >= Cin) return;
    const size_t plane = (size_t)H * W;
    const bool col_ok = px >= 0 && px < W;
    const bool owner = col_ok && lane >= 1 && lane <= SC_TILE;      // this lane's pixel belongs to this strip
    const int pxc = col_ok ? px : 0;
    const float* __restrict__ xp = x + ((size_t)b * Cin + ci) * plane + pxc;
    const float* __restrict__ g0p = gy + ((size_t)b * SC_CO + 0) * plane + pxc;
    const float* __restrict__ g1p = gy + ((size_t)b * SC_CO + 1) * plane + pxc;
    float acc[SC_CO][9];
#pragma unroll
    for (int co = 0; co < SC_CO; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = 0.0f;
    float bsum0 = 0.0f, bsum1 = 0.0f;
    // sliding window over the rows: r0 / r1 / r2 = x[ci][y-1 / y / y+1][px - 1 .. px + 1]
    float r0[3] = {0.0f, 0.0f, 0.0f}, r1[3], r2[3];
    {
        const float c1 = col_ok ? xp[0] : 0.0f;
        const float c2 = (col_ok && H > 1) ? xp[W] : 0.0f;
        r1[0] = from_left(c1); r1[1] = c1; r1[2] = from_right(c1);
        r2[0] = from_left(c2); r2[1] = c2; r2[2] = from_right(c2);
    }
    // UY rows per trip, their loads (the two gradient planes and the incoming x row) requested together: one row per
    // trip left 3 loads in flight per wave and the kernel waited on memory latency 68 times in a row
    constexpr int UY = 8;
    for (int y0 = 0; y0 < H; y0 += UY) {
        float g0v[UY], g1v[UY], cnv[UY];
#pragma unroll
        for (int u = 0; u < UY; ++u) {
            const int y = y0 + u;
            const bool row = y < H;
            const size_t o = (size_t)(row ? y : 0) * W;
            g0v[u] = (owner && row) ? g0p[o] : 0.0f;
            g1v[u] = (owner && row) ? g1p[o] : 0.0f;
            cnv[u] = (col_ok && y + 2 < H) ? xp[(size_t)(y + 2) * W] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < UY; ++u) {
            if (y0 + u >= H) break;                  // wave-uniform
            const float g0 = g0v[u], g1 = g1v[u], cn = cnv[u];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                acc[0][0 + dx] = __builtin_fmaf(g0, r0[dx], acc[0][0 + dx]);
                acc[0][3 + dx] = __builtin_fmaf(g0, r1[dx], acc[0][3 + dx]);
                acc[0][6 + dx] = __builtin_fmaf(g0, r2[dx], acc[0][6 + dx]);
                acc[1][0 + dx] = __builtin_fmaf(g1, r0[dx], acc[1][0 + dx]);
                acc[1][3 + dx] = __builtin_fmaf(g1, r1[dx], acc[1][3 + dx]);
                acc[1][6 + dx] = __builtin_fmaf(g1, r2[dx], acc[1][6 + dx]);
            }
            bsum0 += g0;
            bsum1 += g1;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                r0[dx] = r1[dx];
                r1[dx] = r2[dx];
            }
            r2[0] = from_left(cn);
            r2[1] = cn;
            r2[2] = from_right(cn);
        }
    }
    const int n_out = SC_CO * Cin * 9 + SC_CO;
    float* __restrict__ mine = partials + ((size_t)b * gridDim.x + blockIdx.x) * n_out;
#pragma unroll
    for (int co = 0; co < SC_CO; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float s = wave_sum_f32(acc[co][t]);
            if (lane == 0) mine[((size_t)co * Cin + ci) * 9 + t] = s;
        }
    if (ci == 0) {
        const float s0 = wave_sum_f32(bsum0), s1 = wave_sum_f32(bsum1);
        if (lane == 0) {
            mine[SC_CO * Cin * 9 + 0] = s0;
            mine[SC_CO * Cin * 9 + 1] = s1;
        }
    }
}

// out[e] (+)= sum over the n_part partial vectors, fixed order.  grid ceil(n_out/256), block 256
__global__ __launch_bounds__(256) void smallconv_reduce_kernel(const float* __restrict__ partials, int n_part, int n_out,
                                                                int n_w, float* __restrict__ gw, float* __restrict__ gb,
                                                                int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_out) return;
    float s = 0.0f;
    for (int p = 0; p < n_part; ++p) s += partials[(size_t)p * n_out + e];
    float* dst = e < n_w ? gw + e : (gb ? gb + (e - n_w) : nullptr);
    if (dst) *dst = accumulate ? *dst + s : s;
}

bool smallconv_args_ok(const char* what, int B, int Cin, int H, int W) {
    if (B < 0 || Cin < 1 || H < 1 || W < 1 || H > 65535 || B > 65535) {
        camli_set_error("%s: bad shape B=%d Cin=%d H=%d W=%d", what, B, Cin, H, W);
        return false;
    }
    return true;
}

}  // namespace

extern "C" int camli_conv3x3_co2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                                     int W, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!x || !w || !y) { camli_set_error("camli_conv3x3_co2_fwd: null pointer"); return CAMLI_EINVAL; }
    if (!smallconv_args_ok("camli_conv3x3_co2_fwd", B, Cin, H, W)) return CAMLI_EINVAL;
    const size_t lds = (size_t)(SC_CO * Cin * SC_WPAD + 4 * 2 * SC_CO * 64) * sizeof(float);
    if (lds > 64 * 1024) { camli_set_error("camli_conv3x3_co2_fwd: Cin=%d exceeds the LDS weight buffer", Cin); return CAMLI_ENOTSUP; }
    hipLaunchKernelGGL(conv3x3_co2_fwd_kernel, dim3(camli_divup(W, SC_TILE), camli_divup(H, 2), B), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), x, w, bias, y, Cin, H, W);
    return camli_check_launch("camli_conv3x3_co2_fwd");
}

extern "C" int camli_conv3x3_co2_bwd_data(const float* gy, const float* w, float* gx, int B, int Cin, int H, int W,
                                          void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gy || !w || !gx) { camli_set_error("camli_conv3x3_co2_bwd_data: null pointer"); return CAMLI_EINVAL; }
    if (!smallconv_args_ok("camli_conv3x3_co2_bwd_data", B, Cin, H, W)) return CAMLI_EINVAL;
    const size_t lds = (size_t)SC_CO * Cin * SC_WPAD * sizeof(float);
    if (lds > 64 * 1024) { camli_set_error("camli_conv3x3_co2_bwd_data: Cin=%d exceeds the LDS weight buffer", Cin); return CAMLI_ENOTSUP; }
    hipLaunchKernelGGL(conv3x3_co2_bwd_data_kernel, dim3(camli_divup(W, SC_TILE), camli_divup(H, 2), B), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), gy, w, gx, Cin, H, W);
    return camli_check_launch("camli_conv3x3_co2_bwd_data");
}

extern "C" long long camli_conv3x3_co2_bwd_weight_workspace_bytes(int B, int Cin, int W) {
    return (long long)B * camli_divup(W, SC_TILE) * (SC_CO * Cin * 9 + SC_CO) * (long long)sizeof(float);
}

extern "C" int camli_conv3x3_co2_bwd_weight(const float* gy, const float* x, float* workspace, float* gw, float* gb,
                                            int accumulate, int B, int Cin, int H, int W, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gy || !x || !workspace || !gw) { camli_set_error("camli_conv3x3_co2_bwd_weight: null pointer"); return CAMLI_EINVAL; }
    if (!smallconv_args_ok("camli_conv3x3_co2_bwd_weight", B, Cin, H, W)) return CAMLI_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int strips = camli_divup(W, SC_TILE);
    hipLaunchKernelGGL(conv3x3_co2_bwd_weight_kernel, dim3(strips, camli_divup(Cin, 4), B), dim3(256), 0, s, gy, x, workspace,
                       Cin, H, W);
    const int n_w = SC_CO * Cin * 9, n_out = n_w + SC_CO;
    hipLaunchKernelGGL(smallconv_reduce_kernel, dim3(camli_divup(n_out, 256)), dim3(256), 0, s, workspace, B * strips, n_out, n_w,
                       gw, gb, accumulate);
    return camli_check_launch("camli_conv3x3_co2_bwd_weight");
}

// Separable 5-tap convolutions (1x5 / 5x1, zero padding 2) of the RAFT update block as implicit GEMMs on the fp32 matrix
// cores, gfx950.
//
// Replaces, for GRU2D (models/raft_core.py:110-140), the library convolutions of the reference
//     convz / convr / convq : Conv2d(hidden + input, hidden, (1,5) | (5,1), padding (0,2) | (2,0))
// and the passes around them.  out[b,co,y,x] = sum_{ci,t} W[co,ci,t] * in[b,ci,y(+t-2),x(+t-2)]  is the GEMM
//     M = Cout,  N = pixels,  K = Cin * 5
// but nothing like an im2col matrix exists: a workgroup stages, per chunk of 8 input channels, the weight slab
// [8 ci][5 taps][128 co] and ONE haloed copy of the input tile (row segment of 128 + 4 pixels, or 4 + 4 rows of 32
// pixels), and the five taps are five shifted reads of that copy -- 80 matrix instructions per wave between two
// barriers (a plain GEMM step of the same staging traffic carries 16).  The input is read through two pointers (the
// reference concatenates [h | x] first: one copy of both tensors per gate per iteration).
//
//   * 128 (co) x 128 (pixel) tile per 256-thread workgroup, 2 x 2 waves, 2 x 2 v_mfma_f32_32x32x2_f32 tiles per wave
//     (exact fp32: an fmaf chain over k = (ci, tap) in a fixed order);
//   * weights pre-packed once per pass as Wp[ci][tap][co] (co contiguous): the slab of a chunk is 40 rows of 512 bytes,
//     staged with 16-byte loads / ds_write_b128, fragment of a lane = one ds_read_b32, conflict-free;
//   * horizontal (1x5): pixel tile = 128 consecutive x of one image row (x >= W masked), input copy [ci][4 + 128 + 4];
//     vertical (5x1): pixel tile = 4 rows x 32 columns, input copy [ci][2 + 4 + 2 rows][32]; out-of-image positions are
//     staged as zeros, so the matrix loop has no bounds logic;
//   * double-buffered LDS (2 x 26 KB), next chunk's global loads in flight in registers during the current chunk's MFMAs.
//
// Epilogues (EPI): 0 = out = acc (+ bias);
//   1 = GRU gates: acc + add[b,co,..] -> sigmoid; rows [0, Cout/2) are z, rows [Cout/2, Cout) are r: writes z and r*h
//       (gru_gates of raft_core.py:124-126 / :132-134 without the pre-activation round trip);
//   2 = GRU blend: q = tanh(acc + add); h' = (1 - z) h + z q, optionally nan_to_num (raft_core.py:127-130,135-138); writes h' and q.
#include "camli_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C5_KC = 8;              // input channels per stage
constexpr int C5_TM = 128;            // output channels per tile
constexpr int C5_TN = 128;            // pixels per tile
// LDS images of a stage: weights [k = (ci, tap)][co] (a fragment read = 32 consecutive floats of one row), input copy per
// channel: horizontal [4 + 128 + 4] positions (16-byte aligned segments), vertical [8 rows][32 + 1 pad].
// (Round 4 also tried K-fastest images -- [co][tap][8 ci] / [position][8 ci], one ds_read_b128 per operand block and tap,
// 20 LDS reads per 80 MFMAs instead of 80: the padded images are 28-34 KB per stage, two workgroups per CU instead of
// three, 520 us instead of 445 us at the GRU z|r shape.  profiles/r04_conv5_experiments.txt)
constexpr int C5_LDA = C5_TM;
constexpr int C5_A_DW = C5_KC * 5 * C5_LDA;
constexpr int C5_BH_LD = 136, C5_BV_LD = 8 * 33;
template <bool VERT>
constexpr int c5_stage_dw() { return C5_A_DW + C5_KC * (VERT ? C5_BV_LD : C5_BH_LD); }   // 24.8 KB (three workgroups per CU) / 28.9 KB

struct Conv5Args {
    const float* in0;      // [B, C0, H, W]
    const float* in1;      // [B, C1, H, W] (may be null when C1 == 0)
    const float* wp;       // packed weights [C0 + C1][5][Cout]
    const float* bias;     // [Cout] or null
    const float* add;      // EPI 1/2: per-pass context term [B, Cout, H, W]
    const float* h;        // EPI 1/2: hidden state [B, Ch, H, W] (Ch = Cout / 2 for the gates, Cout for the blend)
    const float* z;        // EPI 2: update gate [B, Cout, H, W]
    float* out;            // EPI 0: [B, Cout, H, W]; EPI 1: z [B, Cout/2, H, W]; EPI 2: h' [B, Cout, H, W]
    float* out2;           // EPI 1: r*h [B, Cout/2, H, W]; EPI 2: q [B, Cout, H, W] (may be null)
    float* out3;           // EPI 1: r [B, Cout/2, H, W] (may be null)
    int B, C0, C1, Cout, H, W;
    int nan_to_num;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <bool VERT, int EPI>
__global__ __launch_bounds__(256) void conv5_fwd_kernel(Conv5Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fk = lane >> 5, fm = lane & 31;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1, Cout = a.Cout;
    const int64_t plane = (int64_t)H * W;

    // pixel tile of this workgroup
    int b, y0, x0;
    if (VERT) {
        const int tx = (W + 31) / 32, ty = (H + 3) / 4;
        int t = blockIdx.x;
        b = t / (tx * ty);
        t -= b * tx * ty;
        y0 = (t / tx) * 4;
        x0 = (t % tx) * 32;
    } else {
        const int tx = (W + C5_TN - 1) / C5_TN;
        int t = blockIdx.x;
        b = t / (tx * H);
        t -= b * tx * H;
        y0 = t / tx;
        x0 = (t % tx) * C5_TN;
    }
    const int m0 = blockIdx.y * C5_TM;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- staging: registers <- global, LDS <- registers ----
    // weight slab of a chunk: 40 rows (ci, tap) x 128 co = 1280 float4, 5 per thread
    // input copy: horizontal 8 ci x 34 float4 = 272 (2 per thread, the second one partial); vertical 8 ci x 8 rows x 8 = 512 (2 per thread)
    float4 ra[5], rb[2];
    const bool vec_in = (W & 3) == 0;
    auto load_stage = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int f = tid + 256 * i;                 // float4 index: row (ci, tap) = f / 32, co quad = f % 32
            const int row = f >> 5, co = m0 + 4 * (f & 31);
            const int ci = ci0 + row / 5;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < Cin && co + 3 < Cout) v = *reinterpret_cast<const float4*>(a.wp + ((int64_t)(ci0 * 5 + row)) * Cout + co);
            else if (ci < Cin) {
                const float* p = a.wp + ((int64_t)(ci0 * 5 + row)) * Cout;
                if (co < Cout) v.x = p[co];
                if (co + 1 < Cout) v.y = p[co + 1];
                if (co + 2 < Cout) v.z = p[co + 2];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int cl, yy, xx;
            bool live;
            if (VERT) {
                cl = f >> 6;                              // 64 float4 per channel: 8 rows x 8
                yy = y0 - 2 + ((f >> 3) & 7);
                xx = x0 + 4 * (f & 7);
                live = true;
            } else {
                cl = f / 34;                              // 34 float4 per channel
                yy = y0;
                xx = x0 - 4 + 4 * (f - cl * 34);
                live = f < C5_KC * 34;
            }
            const int ci = ci0 + cl;
            if (live && ci < Cin && yy >= 0 && yy < H && xx + 3 >= 0 && xx < W) {
                const float* src = (ci < a.C0 ? a.in0 + ((int64_t)b * a.C0 + ci) * plane
                                              : a.in1 + ((int64_t)b * a.C1 + (ci - a.C0)) * plane) + (int64_t)yy * W;
                if (vec_in && xx >= 0 && xx + 3 < W) v = *reinterpret_cast<const float4*>(src + xx);
                else {
                    if (xx >= 0 && xx < W) v.x = src[xx];
                    if (xx + 1 >= 0 && xx + 1 < W) v.y = src[xx + 1];
                    if (xx + 2 >= 0 && xx + 2 < W) v.z = src[xx + 2];
                    if (xx + 3 >= 0 && xx + 3 < W) v.w = src[xx + 3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_stage = [&](float* s) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<float4*>(s + (f >> 5) * C5_LDA + 4 * (f & 31)) = ra[i];
        }
        float* sb = s + C5_A_DW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            if (VERT) {
                float* d = sb + (f >> 6) * C5_BV_LD + ((f >> 3) & 7) * 33 + 4 * (f & 7);
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;       // row stride 33: not 16-byte aligned
            } else if (f < C5_KC * 34) {
                const int cl = f / 34;
                *reinterpret_cast<float4*>(sb + cl * C5_BH_LD + 4 * (f - cl * 34)) = rb[i];
            }
        }
    };

    const int stages = (Cin + C5_KC - 1) / C5_KC;
    load_stage(0);
    store_stage(lds);
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < stages; ++s) {
        if (s + 1 < stages) load_stage((s + 1) * C5_KC);
        const float* sa = lds + buf * c5_stage_dw<VERT>() + wm + fm;
        const float* sb = lds + buf * c5_stage_dw<VERT>() + C5_A_DW;
        // B fragment of pixel block j, tap t, channel c: horizontal sb[c][4 + wn + 32 j + fm + t - 2];
        //                                                vertical   sb[c][(wn / 32 + j + t) * 33 + fm]
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            float fa0[4], fa1[4], fb0[4], fb1[4];
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                const int c = 2 * cp + fk;
                fa0[cp] = sa[(c * 5 + t) * C5_LDA];
                fa1[cp] = sa[(c * 5 + t) * C5_LDA + 32];
                if (VERT) {
                    fb0[cp] = sb[c * C5_BV_LD + (wn / 32 + t) * 33 + fm];
                    fb1[cp] = sb[c * C5_BV_LD + (wn / 32 + 1 + t) * 33 + fm];
                } else {
                    fb0[cp] = sb[c * C5_BH_LD + 2 + wn + fm + t];
                    fb1[cp] = sb[c * C5_BH_LD + 2 + wn + 32 + fm + t];
                }
            }
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cp], fb0[cp], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cp], fb1[cp], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cp], fb0[cp], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cp], fb1[cp], acc[1][1], 0, 0, 0);
            }
        }
        if (s + 1 < stages) store_stage(lds + (buf ^ 1) * c5_stage_dw<VERT>());
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: C/D layout col = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (output channel) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int yy, xx;
        if (VERT) {
            yy = y0 + wn / 32 + j;
            xx = x0 + fm;
        } else {
            yy = y0;
            xx = x0 + wn + 32 * j + fm;
        }
        if (yy >= H || xx >= W) continue;
        const int64_t pix = (int64_t)yy * W + xx;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (co >= Cout) continue;
                float v = acc[i][j][r];
                if (a.bias) v += a.bias[co];
                if (EPI == 0) {
                    a.out[((int64_t)b * Cout + co) * plane + pix] = v;
                } else if (EPI == 1) {
                    const int ch = Cout >> 1;
                    v = sigmoidf_(v + a.add[((int64_t)b * Cout + co) * plane + pix]);
                    if (co < ch) {
                        a.out[((int64_t)b * ch + co) * plane + pix] = v;                                   // z
                    } else {
                        const int64_t o = ((int64_t)b * ch + (co - ch)) * plane + pix;
                        a.out2[o] = v * a.h[o];                                                            // r * h
                        if (a.out3) a.out3[o] = v;                                                         // r
                    }
                } else {
                    const int64_t o = ((int64_t)b * Cout + co) * plane + pix;
                    const float q = tanhf(v + a.add[o]);
                    const float zz = a.z[o], hh = a.h[o];
                    float hn = (1.0f - zz) * hh + zz * q;
                    if (a.nan_to_num) hn = hn != hn ? 0.0f : fminf(fmaxf(hn, -3.4028234663852886e38f), 3.4028234663852886e38f);
                    a.out[o] = hn;
                    if (a.out2) a.out2[o] = q;
                }
            }
    }
}

template <bool VERT>
int launch_conv5(const Conv5Args& a, int epi, hipStream_t stream) {
    const int tiles = VERT ? a.B * ((a.H + 3) / 4) * ((a.W + 31) / 32) : a.B * a.H * ((a.W + C5_TN - 1) / C5_TN);
    dim3 grid(tiles, camli_divup(a.Cout, C5_TM));
    const size_t ldsb = (size_t)2 * c5_stage_dw<VERT>() * sizeof(float);
    if (epi == 0) hipLaunchKernelGGL((conv5_fwd_kernel<VERT, 0>), grid, dim3(256), ldsb, stream, a);
    else if (epi == 1) hipLaunchKernelGGL((conv5_fwd_kernel<VERT, 1>), grid, dim3(256), ldsb, stream, a);
    else hipLaunchKernelGGL((conv5_fwd_kernel<VERT, 2>), grid, dim3(256), ldsb, stream, a);
    return camli_check_launch("camli_conv5_fwd");
}

}  // namespace

// out = conv(cat[in0, in1], W) (+ bias) with the epilogue `epi` (0 plain, 1 GRU gates, 2 GRU blend; see the header).
// wp = the weights packed as [C0 + C1][5][Cout].  All tensors fp32 NCHW-contiguous on the current device.
extern "C" int camli_conv5_fwd(const float* in0, int C0, const float* in1, int C1, const float* wp, const float* bias,
                               const float* add, const float* h, const float* z, float* out, float* out2, float* out3, int B,
                               int Cout, int H, int W, int vertical, int epi, int nan_to_num, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!in0 || (C1 > 0 && !in1) || !wp || !out) {
        camli_set_error("camli_conv5_fwd: null pointer");
        return CAMLI_EINVAL;
    }
    if (B < 0 || C0 < 1 || C1 < 0 || Cout < 1 || H < 1 || W < 1 || epi < 0 || epi > 2 || (epi == 1 && ((Cout & 1) || !add || !h || !out2)) ||
        (epi == 2 && (!add || !h || !z))) {
        camli_set_error("camli_conv5_fwd: bad arguments B=%d C0=%d C1=%d Cout=%d H=%d W=%d epi=%d", B, C0, C1, Cout, H, W, epi);
        return CAMLI_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(wp)) & 15) {
        camli_set_error("camli_conv5_fwd: operands must be 16-byte aligned");
        return CAMLI_EINVAL;
    }
    Conv5Args a{in0, in1, wp, bias, add, h, z, out, out2, out3, B, C0, C1, Cout, H, W, nan_to_num};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return vertical ? launch_conv5<true>(a, epi, s) : launch_conv5<false>(a, epi, s);
}

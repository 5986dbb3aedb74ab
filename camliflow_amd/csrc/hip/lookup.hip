// All-pairs cost-volume pyramid lookup (RAFT "Correlation2D.forward") and its adjoint, gfx950.
//
// Replaces the composed path of models/raft_core.py:70-107 (per level: build a [B*P,9,9,2] grid,
// grid_sample, view; then cat + permute + contiguous; backward = grid_sampler_2d_backward that
// allocates and atomically scatters into a volume-sized zero tensor per level per iteration).
//
//   vol_l : [B*P, h_l, w_l]   coords : [B,2,h,w]   out : [B, L*(2r+1)^2, h, w]
//   out[b, l*Dd*Dd + i*Dd + j, p] = bilinear(vol_l[b*P+p], x = cx/2^l + (i-r), y = cy/2^l + (j-r))
//   (align_corners=True, zeros padding; note the transposed window: i walks x, j walks y.)
//
// All (2r+1)^2 taps of one (pixel, level) share ONE pair of fractional offsets, so they read a
// (2r+2)^2 window of the pixel's own volume slice.  HBM-bound gather: one wave owns 64 consecutive
// source pixels of one level; the 64 windows are pulled into LDS with the lanes running along the
// window (each load touches <= 2r+2 short row segments), then each lane (= pixel) interpolates its
// taps from registers and every output plane is stored as 64 consecutive floats (coalesced).
// The adjoint mirrors this: per-lane window gradients go to LDS and are added to the gradient
// volume with plain read-modify-write -- windows of different source pixels are disjoint memory
// and launches on one stream are ordered, so NO atomics and NO per-call volume-sized temporaries.
#include "camli_common.h"

#include <stdlib.h>

namespace {

constexpr int LK_MAX_LEVELS = 8;

struct LookupLevels {
    float* vol[LK_MAX_LEVELS];
    int h[LK_MAX_LEVELS];
    int w[LK_MAX_LEVELS];
    // adjoint only, optional: visit marks per level, [B][sb][tb[l]] bytes -- one per block of 32 source pixels x 32
    // target pixels of the gradient volume (camli_allpairs_build_bwd_marked skips the blocks never marked)
    unsigned char* mark[LK_MAX_LEVELS];
    int tb[LK_MAX_LEVELS];
    int sb;
};

// grid (ceil(P/64), B, L), block 64
template <int R, bool BACKWARD, int U>
__global__ __launch_bounds__(64) void allpairs_lookup_kernel(LookupLevels lv, const float* __restrict__ coords,
                                                              float* __restrict__ io /* out (fwd) | gout (bwd) */,
                                                              int P, int L) {
    constexpr int DD = 2 * R + 1;      // taps per axis
    constexpr int WN = 2 * R + 2;      // window extent
    constexpr int WE = WN * WN;        // window elements
    constexpr int LD = WE + 1;         // LDS row stride (odd -> conflict-free per-lane rows)
    __shared__ float win[64 * LD];

    const int lane = threadIdx.x;
    const int b = blockIdx.y, l = blockIdx.z;
    const int p0 = blockIdx.x * 64;
    const int p = p0 + lane;
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;
    const int hl = lv.h[l], wl = lv.w[l];
    float* __restrict__ vol = lv.vol[l] + ((size_t)b * P + p0) * (size_t)hl * wl;

    const float scale = 1.0f / (float)(1 << l);
    const float bx = coords[((size_t)b * 2 + 0) * P + pc] * scale;
    const float by = coords[((size_t)b * 2 + 1) * P + pc] * scale;
    const float fx = floorf(bx), fy = floorf(by);
    const float wx0 = bx - fx, wy0 = by - fy;           // weight of the +1 neighbour
    const float wx1 = (fx + 1.0f) - bx, wy1 = (fy + 1.0f) - by;
    // clamp far-away windows so the int conversion is safe; such windows are entirely outside
    const float lim = 1.0e6f;
    const int x0 = (int)fminf(fmaxf(fx, -lim), lim) - R;
    const int y0 = (int)fminf(fmaxf(fy, -lim), lim) - R;

    const int npix = min(64, P - p0);
    const size_t plane = (size_t)P;
    float* __restrict__ chan = io + ((size_t)b * L + l) * DD * DD * plane + p;   // + t*plane per tap

    if (!BACKWARD) {
        // ---- stage the 64 windows: lanes run along the window elements; U windows (2U loads) are
        // put in flight before any is written to LDS -- the kernel is latency-bound otherwise ----
        for (int pp0 = 0; pp0 < npix; pp0 += U) {
            float v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = min(pp0 + u, npix - 1);
                const int sx = __shfl(x0, pp, 64), sy = __shfl(y0, pp, 64);
                const float* __restrict__ src = vol + (size_t)pp * hl * wl;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = h * 64 + lane;
                    const int r = e / WN, c = e - r * WN;
                    const int gy = sy + r, gx = sx + c;
                    const bool in = (e < WE) & (gx >= 0) & (gx < wl) & (gy >= 0) & (gy < hl);
                    v[u][h] = in ? src[gy * wl + gx] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = pp0 + u;
                if (pp < npix) {
                    win[pp * LD + lane] = v[u][0];
                    if (lane + 64 < WE) win[pp * LD + 64 + lane] = v[u][1];
                }
            }
        }
        __syncthreads();
        if (!valid) return;
        // ---- per-lane interpolation: horizontal pass then vertical pass ----
        float hrow[2][DD];   // two consecutive horizontally-interpolated window rows
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            float wrow[WN];
#pragma unroll
            for (int c = 0; c < WN; ++c) wrow[c] = win[lane * LD + r * WN + c];
#pragma unroll
            for (int i = 0; i < DD; ++i) hrow[r & 1][i] = wrow[i] * wx1 + wrow[i + 1] * wx0;
            if (r >= 1) {
                const int j = r - 1;   // tap row j uses window rows j and j+1
#pragma unroll
                for (int i = 0; i < DD; ++i)
                    chan[(size_t)(i * DD + j) * plane] = hrow[(r - 1) & 1][i] * wy1 + hrow[r & 1][i] * wy0;
            }
        }
    } else {
        // ---- per-lane window gradient ----
        float g[WN][WN];
#pragma unroll
        for (int r = 0; r < WN; ++r)
#pragma unroll
            for (int c = 0; c < WN; ++c) g[r][c] = 0.0f;
        if (valid) {
#pragma unroll
            for (int j = 0; j < DD; ++j)
#pragma unroll
                for (int i = 0; i < DD; ++i) {
                    const float go = chan[(size_t)(i * DD + j) * plane];
                    const float gy1 = go * wy1, gy0 = go * wy0;
                    g[j][i] += gy1 * wx1;
                    g[j][i + 1] += gy1 * wx0;
                    g[j + 1][i] += gy0 * wx1;
                    g[j + 1][i + 1] += gy0 * wx0;
                }
        }
#pragma unroll
        for (int r = 0; r < WN; ++r)
#pragma unroll
            for (int c = 0; c < WN; ++c) win[lane * LD + r * WN + c] = g[r][c];
        __syncthreads();
        // ---- add each window into the gradient volume (disjoint per source pixel: plain RMW);
        // U windows' loads are issued before the first add/store ----
        for (int pp0 = 0; pp0 < npix; pp0 += U) {
            float v[U][2];
            int off[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = pp0 + u;
                const int ppc = min(pp, npix - 1);
                const int sx = __shfl(x0, ppc, 64), sy = __shfl(y0, ppc, 64);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = h * 64 + lane;
                    const int r = e / WN, c = e - r * WN;
                    const int gy = sy + r, gx = sx + c;
                    const bool in = (pp < npix) & (e < WE) & (gx >= 0) & (gx < wl) & (gy >= 0) & (gy < hl);
                    off[u][h] = in ? (ppc * hl + gy) * wl + gx : -1;
                    v[u][h] = in ? vol[off[u][h]] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ppc = min(pp0 + u, npix - 1);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (off[u][h] >= 0) vol[off[u][h]] = v[u][h] + win[ppc * LD + h * 64 + lane];
            }
        }
    }
}

// Four waves per 64 source pixels.  Measured on the bench shape (tools/ab_lookup.py): one level costs 18-21 us whether its
// slices are 32 KB or 480 B, and deeper load batches change nothing -- the time is the SERIAL chain of the single wave
// that owns a pixel group (128 staging loads with their address arithmetic, 100 LDS reads, ~350 FMAs, 81 stores) times
// the number of rounds the LDS footprint allows (6 one-wave workgroups per CU).  Here the same 25.8 KB window tile is
// shared by four waves: each stages 16 of the 64 windows and produces a quarter of the tap rows (forward) / window rows
// (adjoint), so the chain is ~4x shorter and a CU holds 24 waves.  Arithmetic per output element is unchanged
// (same products, same summation order).
// PB = source pixels per workgroup (64 or 32), block 4 * PB threads.  With PB = 32 the two waves of a workgroup each
// stage 16 windows and every wave interpolates all 32 pixels twice over -- lanes 0-31 and 32-63 take different row
// groups -- so a CU holds twice as many independent (stage -> barrier -> interpolate/store) chains in the same LDS.
// grid (ceil(P/PB), B, L)
template <int R, bool BACKWARD, int PB>
__global__ __launch_bounds__(4 * PB) void allpairs_lookup4_kernel(LookupLevels lv, const float* __restrict__ coords,
                                                                  float* __restrict__ io /* out (fwd) | gout (bwd) */,
                                                                  int P, int L) {
    constexpr int DD = 2 * R + 1;      // taps per axis
    constexpr int WN = 2 * R + 2;      // window extent
    constexpr int WE = WN * WN;        // window elements
    constexpr int LD = WE + 1;         // LDS row stride (odd -> conflict-free per-lane rows)
    constexpr int PW = 16;             // windows staged per wave
    static_assert(R == 4 && (PB == 64 || PB == 32), "row split below is written for radius 4");
    __shared__ float win[PB * LD];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pl = lane & (PB - 1);                                   // pixel of this lane within the group
    const int grp = PB == 64 ? wv : 2 * wv + (lane >> 5);              // row group 0..3 this lane works on
    const int b = blockIdx.y, l = blockIdx.z;
    const int p0 = blockIdx.x * PB;
    const int p = p0 + pl;
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;
    const int hl = lv.h[l], wl = lv.w[l];
    float* __restrict__ vol = lv.vol[l] + ((size_t)b * P + p0) * (size_t)hl * wl;

    const float scale = 1.0f / (float)(1 << l);
    const float bx = coords[((size_t)b * 2 + 0) * P + pc] * scale;
    const float by = coords[((size_t)b * 2 + 1) * P + pc] * scale;
    const float fx = floorf(bx), fy = floorf(by);
    const float wx0 = bx - fx, wy0 = by - fy;           // weight of the +1 neighbour
    const float wx1 = (fx + 1.0f) - bx, wy1 = (fy + 1.0f) - by;
    const float lim = 1.0e6f;
    const int x0 = (int)fminf(fmaxf(fx, -lim), lim) - R;
    const int y0 = (int)fminf(fmaxf(fy, -lim), lim) - R;

    const int npix = min(PB, P - p0);
    const size_t plane = (size_t)P;
    float* __restrict__ chan = io + ((size_t)b * L + l) * DD * DD * plane + p;   // + t*plane per tap
    // element of the window this lane carries during staging / scatter (two per lane: e and e + 64)
    const int e0r = lane / WN, e0c = lane - e0r * WN;
    const int e1 = lane + 64;
    const int e1r = e1 / WN, e1c = e1 - e1r * WN;
    const bool e1_in = e1 < WE;

    if (!BACKWARD) {
        // ---- stage: wave wv pulls windows [16 wv, 16 wv + 16), all 32 loads in flight before the first LDS write ----
        float v[PW][2];
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int pp = min(wv * PW + u, npix - 1);
            const int sx = __shfl(x0, pp, 64), sy = __shfl(y0, pp, 64);
            const float* __restrict__ src = vol + (size_t)pp * hl * wl;
            const int gy0 = sy + e0r, gx0 = sx + e0c, gy1 = sy + e1r, gx1 = sx + e1c;
            const bool in0 = (gx0 >= 0) & (gx0 < wl) & (gy0 >= 0) & (gy0 < hl);
            const bool in1 = e1_in & (gx1 >= 0) & (gx1 < wl) & (gy1 >= 0) & (gy1 < hl);
            v[u][0] = in0 ? src[gy0 * wl + gx0] : 0.0f;
            v[u][1] = in1 ? src[gy1 * wl + gx1] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int pp = wv * PW + u;
            if (pp < npix) {
                win[pp * LD + lane] = v[u][0];
                if (e1_in) win[pp * LD + e1] = v[u][1];
            }
        }
        __syncthreads();
        if (!valid) return;
        // ---- interpolate: tap rows {0,1,2} {3,4} {5,6} {7,8} per group; tap row j blends window rows j and j+1 ----
        const int j0 = grp == 0 ? 0 : 2 * grp + 1;
        const int j1 = grp == 0 ? 2 : 2 * grp + 2;         // inclusive
        float hrow[2][DD];
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            if (r >= j0 && r <= j1 + 1) {                   // wave-uniform for PB = 64, per half-wave for PB = 32
                float wrow[WN];
#pragma unroll
                for (int c = 0; c < WN; ++c) wrow[c] = win[pl * LD + r * WN + c];
#pragma unroll
                for (int i = 0; i < DD; ++i) hrow[r & 1][i] = wrow[i] * wx1 + wrow[i + 1] * wx0;
                if (r > j0) {
                    const int j = r - 1;
#pragma unroll
                    for (int i = 0; i < DD; ++i)
                        chan[(size_t)(i * DD + j) * plane] = hrow[(r - 1) & 1][i] * wy1 + hrow[r & 1][i] * wy0;
                }
            }
        }
    } else {
        // ---- visit marks: every window row is a run of <= 10 target pixels, i.e. at most two 32-pixel blocks ----
        if (lv.mark[l] && wv == 0 && valid && (PB == 64 || lane < 32)) {
            unsigned char* __restrict__ mk = lv.mark[l] + ((size_t)b * lv.sb + (p >> 5)) * lv.tb[l];
            const int xa = max(x0, 0), xb = min(x0 + WN - 1, wl - 1);
            if (xa <= xb) {
#pragma unroll
                for (int r = 0; r < WN; ++r) {
                    const int gy = y0 + r;
                    if (gy >= 0 && gy < hl) {
                        mk[(gy * wl + xa) >> 5] = 1;
                        mk[(gy * wl + xb) >> 5] = 1;
                    }
                }
            }
        }
        // ---- window rows {0,1,2} {3,4,5} {6,7} {8,9} per group; row r collects tap rows r-1 (weight wy0) and r (wy1),
        // added in the order of the one-wave kernel: (r-1,c-1) (r-1,c) (r,c-1) (r,c) ----
        const int r0 = grp < 2 ? 3 * grp : 2 * grp + 2;
        const int r1 = grp < 2 ? 3 * grp + 2 : 2 * grp + 3;   // inclusive
        float gprev[DD], gcur[DD];                          // gout tap rows r-1 and r of this lane's pixel
#pragma unroll
        for (int i = 0; i < DD; ++i) gprev[i] = 0.0f;
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            if (r >= r0 - 1 && r <= r1) {                   // r0 - 1 only loads the tap row
#pragma unroll
                for (int i = 0; i < DD; ++i) gcur[i] = (valid && r < DD) ? chan[(size_t)(i * DD + r) * plane] : 0.0f;
                if (r >= r0) {
#pragma unroll
                    for (int c = 0; c < WN; ++c) {
                        float acc = 0.0f;
                        if (r >= 1 && c >= 1) acc += (gprev[c - 1] * wy0) * wx0;
                        if (r >= 1 && c < DD) acc += (gprev[c] * wy0) * wx1;
                        if (r < DD && c >= 1) acc += (gcur[c - 1] * wy1) * wx0;
                        if (r < DD && c < DD) acc += (gcur[c] * wy1) * wx1;
                        win[pl * LD + r * WN + c] = acc;
                    }
                }
#pragma unroll
                for (int i = 0; i < DD; ++i) gprev[i] = gcur[i];
            }
        }
        __syncthreads();
        // ---- add windows [16 wv, 16 wv + 16) into the gradient volume (disjoint per source pixel: plain RMW) ----
        float v[PW][2];
        int off[PW][2];
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int pp = wv * PW + u;
            const int ppc = min(pp, npix - 1);
            const int sx = __shfl(x0, ppc, 64), sy = __shfl(y0, ppc, 64);
            const int gy0 = sy + e0r, gx0 = sx + e0c, gy1 = sy + e1r, gx1 = sx + e1c;
            const bool in0 = (pp < npix) & (gx0 >= 0) & (gx0 < wl) & (gy0 >= 0) & (gy0 < hl);
            const bool in1 = (pp < npix) & e1_in & (gx1 >= 0) & (gx1 < wl) & (gy1 >= 0) & (gy1 < hl);
            off[u][0] = in0 ? (ppc * hl + gy0) * wl + gx0 : -1;
            off[u][1] = in1 ? (ppc * hl + gy1) * wl + gx1 : -1;
            v[u][0] = in0 ? vol[off[u][0]] : 0.0f;
            v[u][1] = in1 ? vol[off[u][1]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int ppc = min(wv * PW + u, npix - 1);
            if (off[u][0] >= 0) vol[off[u][0]] = v[u][0] + win[ppc * LD + lane];
            if (off[u][1] >= 0) vol[off[u][1]] = v[u][1] + win[ppc * LD + e1];
        }
    }
}

template <bool BACKWARD>
int launch_lookup(float* const* vols, const int* hs, const int* ws, int L, const float* coords, float* io, int B,
                  int h, int w, int r, hipStream_t stream, const char* what, unsigned char* const* marks = nullptr) {
    if (!vols || !hs || !ws || !coords || !io) {
        camli_set_error("%s: null pointer", what);
        return CAMLI_EINVAL;
    }
    if (L < 1 || L > LK_MAX_LEVELS || B < 0 || h < 1 || w < 1 || B > 65535) {
        camli_set_error("%s: bad shape B=%d h=%d w=%d L=%d", what, B, h, w, L);
        return CAMLI_EINVAL;
    }
    if (r != 4) {
        camli_set_error("%s: radius %d not supported (the models use radius 4)", what, r);
        return CAMLI_ENOTSUP;
    }
    if (B == 0) return CAMLI_OK;
    LookupLevels lv;
    for (int l = 0; l < L; ++l) {
        if (!vols[l] || hs[l] < 1 || ws[l] < 1) {
            camli_set_error("%s: level %d is empty", what, l);
            return CAMLI_EINVAL;
        }
        lv.vol[l] = vols[l];
        lv.h[l] = hs[l];
        lv.w[l] = ws[l];
        if (marks && !marks[l]) {
            camli_set_error("%s: level %d has no mark array", what, l);
            return CAMLI_EINVAL;
        }
        lv.mark[l] = marks ? marks[l] : nullptr;
        lv.tb[l] = camli_divup(hs[l] * ws[l], 32);
    }
    for (int l = L; l < LK_MAX_LEVELS; ++l) { lv.mark[l] = nullptr; lv.tb[l] = 0; }
    lv.sb = camli_divup(h * w, 32);
    const int P = h * w;
    // CAMLI_LOOKUP_WAVES=1 keeps the one-wave-per-pixel-group form; CAMLI_LOOKUP_PB picks the pixel-group size (A/B runs)
    static const int waves = [] { const char* e = getenv("CAMLI_LOOKUP_WAVES"); return e ? atoi(e) : 4; }();
    static const int pb = [] { const char* e = getenv("CAMLI_LOOKUP_PB"); return e ? atoi(e) : 64; }();
    if (waves == 1 && !marks)      // the one-wave form does not write marks
        hipLaunchKernelGGL((allpairs_lookup_kernel<4, BACKWARD, BACKWARD ? 8 : 16>), dim3(camli_divup(P, 64), B, L), dim3(64), 0,
                           stream, lv, coords, io, P, L);
    else if (pb == 32)
        hipLaunchKernelGGL((allpairs_lookup4_kernel<4, BACKWARD, 32>), dim3(camli_divup(P, 32), B, L), dim3(128), 0, stream,
                           lv, coords, io, P, L);
    else
        hipLaunchKernelGGL((allpairs_lookup4_kernel<4, BACKWARD, 64>), dim3(camli_divup(P, 64), B, L), dim3(256), 0, stream,
                           lv, coords, io, P, L);
    return camli_check_launch(what);
}

}  // namespace

extern "C" int camli_allpairs_lookup_fwd(const float* const* vols, const int* hs, const int* ws, int L,
                                         const float* coords, float* out, int B, int h, int w, int r, void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    return launch_lookup<false>(const_cast<float* const*>(vols), hs, ws, L, coords, out, B, h, w, r,
                                reinterpret_cast<hipStream_t>(stream), "camli_allpairs_lookup_fwd");
}

extern "C" int camli_allpairs_lookup_bwd(float* const* gvols, const int* hs, const int* ws, int L,
                                         const float* coords, const float* gout, int B, int h, int w, int r,
                                         void* stream) {
    if (B == 0) return CAMLI_OK;   // empty problem: nothing to launch (pointers may be null)
    return launch_lookup<true>(gvols, hs, ws, L, coords, const_cast<float*>(gout), B, h, w, r,
                               reinterpret_cast<hipStream_t>(stream), "camli_allpairs_lookup_bwd");
}

// The adjoint plus visit marks: marks[l] ([B][ceil(h*w/32)][ceil(hs[l]*ws[l]/32)] bytes, zeroed by the caller once per
// backward pass) receives a non-zero byte for every 32 x 32 block (source pixels x target pixels) of gvols[l] a window
// was added into.  camli_allpairs_build_bwd_marked reads them.
extern "C" int camli_allpairs_lookup_bwd_marked(float* const* gvols, const int* hs, const int* ws, int L,
                                                const float* coords, const float* gout, int B, int h, int w, int r,
                                                unsigned char* const* marks, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!marks) { camli_set_error("camli_allpairs_lookup_bwd_marked: null marks"); return CAMLI_EINVAL; }
    return launch_lookup<true>(gvols, hs, ws, L, coords, const_cast<float*>(gout), B, h, w, r,
                               reinterpret_cast<hipStream_t>(stream), "camli_allpairs_lookup_bwd_marked", marks);
}

// ---- "clean after use" for a gradient pyramid that lives across steps --------------------------------------------------
// The lookups' adjoints write a band around the flow field (~20 % of the 32 x 32 blocks of level 0); zero-filling the
// whole 2.8 GB pyramid before every backward pass was 5.7 GB of stores per step.  With the marks in hand the pass can
// instead put back what it dirtied: every marked block is zeroed and its mark cleared, the rest was never written.
// grid (ceil(P/32) source blocks, B); block 256 = 32 rows x 8 lanes of 16 bytes (one 128-byte row segment of a block).
namespace {
__global__ __launch_bounds__(256) void clear_marked_kernel(float* __restrict__ vol, unsigned char* __restrict__ marks,
                                                           int P, int Pl, int tb) {
    const int sb = blockIdx.x, b = blockIdx.y;
    const int row = threadIdx.x >> 3, part = threadIdx.x & 7;
    unsigned char* __restrict__ mrow = marks + ((size_t)b * gridDim.x + sb) * tb;
    const int src = sb * 32 + row;
    float* __restrict__ vrow = vol + ((size_t)b * P + src) * Pl;
    const bool vec = (Pl & 3) == 0 && (reinterpret_cast<uintptr_t>(vol) & 15) == 0;
    for (int t = 0; t < tb; ++t) {
        if (!mrow[t]) continue;                 // block-uniform
        if (src < P) {
            const int col = t * 32 + part * 4;
            if (vec && col + 3 < Pl) {
                *reinterpret_cast<float4*>(vrow + col) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col + j < Pl) vrow[col + j] = 0.0f;
            }
        }
    }
    __syncthreads();                            // every thread has read the marks it needed
    for (int t = threadIdx.x; t < tb; t += 256) mrow[t] = 0;
}
}  // namespace

extern "C" int camli_allpairs_clear_marked(float* const* gvols, const int* p_levels, int L, unsigned char* const* marks, int B,
                                           int P, void* stream) {
    if (B == 0) return CAMLI_OK;
    if (!gvols || !p_levels || !marks || L < 1 || L > 8 || B < 0 || P < 1 || B > 65535) {
        camli_set_error("camli_allpairs_clear_marked: bad arguments L=%d B=%d P=%d", L, B, P);
        return CAMLI_EINVAL;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int l = 0; l < L; ++l) {
        if (!gvols[l] || !marks[l] || p_levels[l] < 1) { camli_set_error("camli_allpairs_clear_marked: level %d", l); return CAMLI_EINVAL; }
        hipLaunchKernelGGL(clear_marked_kernel, dim3(camli_divup(P, 32), B), dim3(256), 0, s, gvols[l], marks[l], P, p_levels[l],
                           camli_divup(p_levels[l], 32));
    }
    return camli_check_launch("camli_allpairs_clear_marked");
}

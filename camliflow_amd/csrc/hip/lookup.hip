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

namespace {

constexpr int LK_MAX_LEVELS = 8;

struct LookupLevels {
    float* vol[LK_MAX_LEVELS];
    int h[LK_MAX_LEVELS];
    int w[LK_MAX_LEVELS];
};

// grid (ceil(P/64), B, L), block 64
template <int R, bool BACKWARD>
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
        constexpr int U = 16;
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
        constexpr int U = 8;
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

template <bool BACKWARD>
int launch_lookup(float* const* vols, const int* hs, const int* ws, int L, const float* coords, float* io, int B,
                  int h, int w, int r, hipStream_t stream, const char* what) {
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
    }
    const int P = h * w;
    dim3 grid(camli_divup(P, 64), B, L);
    hipLaunchKernelGGL((allpairs_lookup_kernel<4, BACKWARD>), grid, dim3(64), 0, stream, lv, coords, io, P, L);
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

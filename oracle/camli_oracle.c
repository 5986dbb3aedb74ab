/*
 * camli_oracle.c -- CPU restatement of the CamLiFlow/CamLiRAFT hot-path operators.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the executable specification the HIP kernels in
 * camliflow_amd/csrc/hip are diffed against.  Nothing under camliflow_amd/ may import, link or
 * call it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity pin: every function here is checked against outputs of the reference's own Python
 * path (models/csrc/wrapper.py fallbacks, models/raft_core.py, models/utils.py) imported in the
 * build container; the resulting vectors are committed under tests/golden/ together with the
 * generating script (tests/golden/make_golden.py).  The reference ships no golden vectors of
 * its own and its CUDA path cannot be built here (no nvcc).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (see oracle/Makefile).  -ffp-contract=off
 * matters: the index-producing ops (KNN, FPS) are specified with UNFUSED fp32 arithmetic.
 *
 * All citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_K 64

/* ------------------------------------------------------------------------------------------
 * k_nearest_neighbor
 * follows models/csrc/k_nearest_neighbor/k_nearest_neighbor_kernel.cu:9-50 (2-D) and :52-95
 * (3-D); host shape handling k_nearest_neighbor.cpp:6-24.
 *   input  [B, M, D] channel-last, query [B, Nq, D], out int64 [B, Nq, k], D in {2, 3}.
 * Semantics that the HIP kernel must reproduce bit-for-bit:
 *   - d = ((ux-x)*(ux-x) + (uy-y)*(uy-y)) [+ (uz-z)*(uz-z)], unfused fp32, left to right
 *   - candidates scanned in index order; skipped iff d > dist[k-1]
 *   - insertion starts at slot min(idx, k-1) and moves left past entries with dist > d
 *     (stable: lands after every entry with dist <= d); the old slot k-1 is dropped, so a
 *     candidate that ties the current k-th distance REPLACES slot k-1
 *   - unfilled slots keep (1e9, index 0)
 * ------------------------------------------------------------------------------------------ */
int oracle_knn(const float *input, const float *query, int64_t *out,
               int B, int M, int Nq, int D, int k)
{
    if (k < 1 || k > ORACLE_MAX_K || (D != 2 && D != 3)) return -1;
    for (int b = 0; b < B; ++b) {
        const float *in_b = input + (size_t)b * M * D;
        for (int q = 0; q < Nq; ++q) {
            const float *qp = query + ((size_t)b * Nq + q) * D;
            float ux = qp[0], uy = qp[1], uz = (D == 3) ? qp[2] : 0.0f;
            float nn_d[ORACLE_MAX_K];
            int nn_i[ORACLE_MAX_K];
            for (int i = 0; i < ORACLE_MAX_K; ++i) { nn_d[i] = 1e9f; nn_i[i] = 0; }
            for (int idx = 0; idx < M; ++idx) {
                float x = in_b[idx * D + 0], y = in_b[idx * D + 1];
                float d = (ux - x) * (ux - x) + (uy - y) * (uy - y);
                if (D == 3) { float z = in_b[idx * D + 2]; d = d + (uz - z) * (uz - z); }
                if (d > nn_d[k - 1]) continue;
                int j = idx < k - 1 ? idx : k - 1;
                while (j > 0 && nn_d[j - 1] > d) {
                    nn_d[j] = nn_d[j - 1];
                    nn_i[j] = nn_i[j - 1];
                    --j;
                }
                nn_d[j] = d;
                nn_i[j] = idx;
            }
            int64_t *o = out + ((size_t)b * Nq + q) * k;
            for (int i = 0; i < k; ++i) o[i] = nn_i[i];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * furthest_point_sampling
 * follows models/csrc/wrapper.py:83-96 (the path that runs on every machine) which is
 * index-identical to furthest_point_sampling_kernel.cu:49-78 on tie-free data.
 *   xyz [B, N, 3], out int64 [B, n_samples]; start index 0; dist init 1e10;
 *   d = ((x2-x1)^2 + (y2-y1)^2) + (z2-z1)^2 unfused fp32; dist = min(dist, d);
 *   next = argmax(dist), LOWEST index among equal maxima (torch.max on CPU; the CUDA tree
 *   reduction's own tie order, kernel.cu:5-10,23-32, depends on the thread id bit pattern and
 *   is documented, not imitated).
 * ------------------------------------------------------------------------------------------ */
int oracle_fps(const float *xyz, int64_t *out, int B, int N, int n_samples)
{
    if (n_samples < 1 || N < 1) return -1;
    float *dist = (float *)malloc(sizeof(float) * (size_t)N);
    if (!dist) return -2;
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int i = 0; i < N; ++i) dist[i] = 1e10f;
        int cur = 0;
        for (int s = 0; s < n_samples; ++s) {
            out[(size_t)b * n_samples + s] = cur;
            float x1 = p[cur * 3 + 0], y1 = p[cur * 3 + 1], z1 = p[cur * 3 + 2];
            float best = -1.0f;
            int best_i = 0;
            for (int i = 0; i < N; ++i) {
                float dx = p[i * 3 + 0] - x1, dy = p[i * 3 + 1] - y1, dz = p[i * 3 + 2] - z1;
                float d = dx * dx + dy * dy + dz * dz;
                float nd = dist[i] < d ? dist[i] : d;
                dist[i] = nd;
                if (nd > best) { best = nd; best_i = i; }
            }
            cur = best_i;
        }
    }
    free(dist);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * correlation2d forward / backward
 * follows models/csrc/wrapper.py:41-50 == correlation_forward_kernel.cu:11-49 and
 * correlation_backward_kernel.cu:4-74.
 *   in1, in2: NHWC [B,H,W,C]   out: NCHW [B, Dd*Dd, H, W], Dd = 2*md+1
 *   out[n, (dy+md)*Dd+(dx+md), y, x] = (1/C) * sum_c in1[n,y,x,c] * in2[n,y+dy,x+dx,c]
 *   zero outside the image.  Tolerance-checked (fp32 summation order is not part of the spec;
 *   the reference's own criterion is mean-abs < 1e-6, correlation_test.cpp:82-89).
 *   backward returns NHWC grads (the layout wrapper.py:34-35 hands back to autograd).
 * ------------------------------------------------------------------------------------------ */
int oracle_corr2d_fwd(const float *in1, const float *in2, float *out,
                      int B, int C, int H, int W, int md)
{
    int Dd = 2 * md + 1;
    for (int n = 0; n < B; ++n)
        for (int dy = -md; dy <= md; ++dy)
            for (int dx = -md; dx <= md; ++dx) {
                int tc = (dy + md) * Dd + (dx + md);
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        int y2 = y + dy, x2 = x + dx;
                        float s = 0.0f;
                        if (x2 >= 0 && y2 >= 0 && x2 < W && y2 < H) {
                            const float *a = in1 + (((size_t)n * H + y) * W + x) * C;
                            const float *b = in2 + (((size_t)n * H + y2) * W + x2) * C;
                            for (int c = 0; c < C; ++c) s += a[c] * b[c];
                            s = s / (float)C;
                        }
                        out[(((size_t)n * Dd * Dd + tc) * H + y) * W + x] = s;
                    }
            }
    return 0;
}

int oracle_corr2d_bwd(const float *gout, const float *in1, const float *in2,
                      float *g1, float *g2, int B, int C, int H, int W, int md)
{
    int Dd = 2 * md + 1;
    size_t total = (size_t)B * H * W * C;
    memset(g1, 0, total * sizeof(float));
    memset(g2, 0, total * sizeof(float));
    for (int n = 0; n < B; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int dy = -md; dy <= md; ++dy)
                    for (int dx = -md; dx <= md; ++dx) {
                        int y2 = y + dy, x2 = x + dx;
                        if (x2 < 0 || y2 < 0 || x2 >= W || y2 >= H) continue;
                        int tc = (dy + md) * Dd + (dx + md);
                        float g = gout[(((size_t)n * Dd * Dd + tc) * H + y) * W + x] / (float)C;
                        size_t o1 = (((size_t)n * H + y) * W + x) * C;
                        size_t o2 = (((size_t)n * H + y2) * W + x2) * C;
                        for (int c = 0; c < C; ++c) {
                            g1[o1 + c] += g * in2[o2 + c];
                            g2[o2 + c] += g * in1[o1 + c];
                        }
                    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * All-pairs cost-volume pyramid lookup (RAFT "Correlation2D.forward")
 * follows models/raft_core.py:70-107 (grid_sample bilinear, align_corners=True, zeros padding).
 *   vol_l : level l of the pyramid, [B*P, h_l, w_l] row-major (P = h*w source pixels)
 *   coords: [B, 2, h, w] (x, y) in level-0 pixel units
 *   out   : [B, L*(2r+1)^2, h, w]; channel l*(2r+1)^2 + i*(2r+1) + j samples level l at
 *           (x/2^l + d[i], y/2^l + d[j]), d = -r..r  (the transposed RAFT window, SURVEY 8a
 *           note +)
 * The normalise/un-normalise round trip of raft_core.py:97-107 + grid_sample is restated
 * literally (ix = ((2*x/(w-1) - 1) + 1)/2*(w-1)) so the fp32 sample positions agree.
 * ------------------------------------------------------------------------------------------ */
static inline float unnorm(float p, int size)
{
    float g = 2.0f * p / (float)(size - 1) - 1.0f;
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}

int oracle_allpairs_lookup_fwd(const float *const *vols, const int *hs, const int *ws, int L,
                               const float *coords, float *out, int B, int h, int w, int r)
{
    int Dd = 2 * r + 1, P = h * w;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            float cx = coords[((size_t)b * 2 + 0) * P + p];
            float cy = coords[((size_t)b * 2 + 1) * P + p];
            for (int l = 0; l < L; ++l) {
                int hl = hs[l], wl = ws[l];
                const float *v = vols[l] + ((size_t)b * P + p) * hl * wl;
                float scale = (float)(1 << l);
                float bx = cx / scale, by = cy / scale;
                for (int i = 0; i < Dd; ++i)
                    for (int j = 0; j < Dd; ++j) {
                        float sx = unnorm(bx + (float)(i - r), wl);
                        float sy = unnorm(by + (float)(j - r), hl);
                        float fx = floorf(sx), fy = floorf(sy);
                        int x0 = (int)fx, y0 = (int)fy;
                        /* weights and accumulation order as ATen grid_sampler_2d (nw, ne, sw, se):
                         * nw = (x1 - ix)*(y1 - iy) etc. with x1 = x0 + 1 */
                        float wx1 = (fx + 1.0f) - sx, wx0 = sx - fx;
                        float wy1 = (fy + 1.0f) - sy, wy0 = sy - fy;
                        float acc = 0.0f;
                        if (x0 >= 0 && x0 < wl && y0 >= 0 && y0 < hl)
                            acc += v[y0 * wl + x0] * (wx1 * wy1);
                        if (x0 + 1 >= 0 && x0 + 1 < wl && y0 >= 0 && y0 < hl)
                            acc += v[y0 * wl + x0 + 1] * (wx0 * wy1);
                        if (x0 >= 0 && x0 < wl && y0 + 1 >= 0 && y0 + 1 < hl)
                            acc += v[(y0 + 1) * wl + x0] * (wx1 * wy0);
                        if (x0 + 1 >= 0 && x0 + 1 < wl && y0 + 1 >= 0 && y0 + 1 < hl)
                            acc += v[(y0 + 1) * wl + x0 + 1] * (wx0 * wy0);
                        int ch = l * Dd * Dd + i * Dd + j;
                        out[((size_t)b * L * Dd * Dd + ch) * P + p] = acc;
                    }
            }
        }
    return 0;
}

/* gradient of the lookup w.r.t. the pyramid levels (coords carry no gradient: they are built
 * from detached flow, models/raft_core.py:248, camliraft_core.py:105-106).  gvols[l] must be
 * zero-initialised by the caller; contributions are accumulated. */
int oracle_allpairs_lookup_bwd(float *const *gvols, const int *hs, const int *ws, int L,
                               const float *coords, const float *gout, int B, int h, int w, int r)
{
    int Dd = 2 * r + 1, P = h * w;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            float cx = coords[((size_t)b * 2 + 0) * P + p];
            float cy = coords[((size_t)b * 2 + 1) * P + p];
            for (int l = 0; l < L; ++l) {
                int hl = hs[l], wl = ws[l];
                float *v = gvols[l] + ((size_t)b * P + p) * hl * wl;
                float scale = (float)(1 << l);
                float bx = cx / scale, by = cy / scale;
                for (int i = 0; i < Dd; ++i)
                    for (int j = 0; j < Dd; ++j) {
                        float sx = unnorm(bx + (float)(i - r), wl);
                        float sy = unnorm(by + (float)(j - r), hl);
                        float fx = floorf(sx), fy = floorf(sy);
                        int x0 = (int)fx, y0 = (int)fy;
                        float wx1 = (fx + 1.0f) - sx, wx0 = sx - fx;
                        float wy1 = (fy + 1.0f) - sy, wy0 = sy - fy;
                        int ch = l * Dd * Dd + i * Dd + j;
                        float g = gout[((size_t)b * L * Dd * Dd + ch) * P + p];
                        if (x0 >= 0 && x0 < wl && y0 >= 0 && y0 < hl)
                            v[y0 * wl + x0] += g * (wx1 * wy1);
                        if (x0 + 1 >= 0 && x0 + 1 < wl && y0 >= 0 && y0 < hl)
                            v[y0 * wl + x0 + 1] += g * (wx0 * wy1);
                        if (x0 >= 0 && x0 < wl && y0 + 1 >= 0 && y0 + 1 < hl)
                            v[(y0 + 1) * wl + x0] += g * (wx1 * wy0);
                        if (x0 + 1 >= 0 && x0 + 1 < wl && y0 + 1 >= 0 && y0 + 1 < hl)
                            v[(y0 + 1) * wl + x0 + 1] += g * (wx0 * wy0);
                    }
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * batch_indexing (gather along the point axis) and its adjoint (scatter-add)
 * follows models/utils.py:61-104.
 *   channel-first: data [B,C,N], idx int64 [B,I] -> out [B,C,I]
 * ------------------------------------------------------------------------------------------ */
int oracle_gather_cf(const float *data, const int64_t *idx, float *out, int B, int C, int N, int I)
{
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float *row = data + ((size_t)b * C + c) * N;
            float *o = out + ((size_t)b * C + c) * I;
            const int64_t *ix = idx + (size_t)b * I;
            for (int i = 0; i < I; ++i) {
                if (ix[i] < 0 || ix[i] >= N) return -1;
                o[i] = row[ix[i]];
            }
        }
    return 0;
}

int oracle_scatter_add_cf(const float *gout, const int64_t *idx, float *gdata, int B, int C, int N, int I)
{
    memset(gdata, 0, sizeof(float) * (size_t)B * C * N);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float *row = gdata + ((size_t)b * C + c) * N;
            const float *g = gout + ((size_t)b * C + c) * I;
            const int64_t *ix = idx + (size_t)b * I;
            for (int i = 0; i < I; ++i) {
                if (ix[i] < 0 || ix[i] >= N) return -1;
                row[ix[i]] += g[i];
            }
        }
    return 0;
}

/*   channel-last (models/utils.py:85-104): data [B,N,C] (C = 1 for the rank-2 form [B,N]), idx int64 [B,I]
 *   -> out [B,I,C] = data[b, idx[b,i], :]; the adjoint adds gout rows into gdata rows in ascending i. */
int oracle_gather_cl(const float *data, const int64_t *idx, float *out, int B, int C, int N, int I)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < I; ++i) {
            const int64_t m = idx[(size_t)b * I + i];
            if (m < 0 || m >= N) return -1;
            memcpy(out + ((size_t)b * I + i) * C, data + ((size_t)b * N + m) * C, sizeof(float) * (size_t)C);
        }
    return 0;
}

int oracle_scatter_add_cl(const float *gout, const int64_t *idx, float *gdata, int B, int C, int N, int I)
{
    memset(gdata, 0, sizeof(float) * (size_t)B * N * C);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < I; ++i) {
            const int64_t m = idx[(size_t)b * I + i];
            if (m < 0 || m >= N) return -1;
            float *row = gdata + ((size_t)b * N + m) * C;
            const float *g = gout + ((size_t)b * I + i) * C;
            for (int c = 0; c < C; ++c) row[c] += g[c];
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * knn_interpolation  (k nearest + inverse-distance weights)
 * follows models/utils.py:130-146.
 *   in_xyz [B,3,M], feat [B,C,M], q_xyz [B,3,Nq] (all channel-first), knn int64 [B,Nq,k]
 *   dist_j = max(||in_xyz[:,knn_j] - q||_2, 1e-8); w_j = (1/dist_j) / sum_j (1/dist_j)
 *   out[b,c,q] = sum_j feat[b,c,knn_j] * w_j
 * ------------------------------------------------------------------------------------------ */
int oracle_knn_interp_fwd(const float *in_xyz, const float *feat, const float *q_xyz,
                          const int64_t *knn, float *out, int B, int C, int M, int Nq, int k)
{
    if (k > ORACLE_MAX_K) return -1;
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Nq; ++q) {
            float wgt[ORACLE_MAX_K], wsum = 0.0f;
            const int64_t *ix = knn + ((size_t)b * Nq + q) * k;
            for (int j = 0; j < k; ++j) {
                float s = 0.0f;
                for (int a = 0; a < 3; ++a) {
                    float d = in_xyz[((size_t)b * 3 + a) * M + ix[j]] - q_xyz[((size_t)b * 3 + a) * Nq + q];
                    s += d * d;
                }
                float dist = sqrtf(s);
                if (dist < 1e-8f) dist = 1e-8f;
                wgt[j] = 1.0f / dist;
                wsum += wgt[j];
            }
            for (int c = 0; c < C; ++c) {
                float acc = 0.0f;
                for (int j = 0; j < k; ++j)
                    acc += feat[((size_t)b * C + c) * M + ix[j]] * (wgt[j] / wsum);
                out[((size_t)b * C + c) * Nq + q] = acc;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * PointConvDW core: gather * weight -> max over the k neighbours, and its adjoint
 * follows models/point_conv.py:122-128 (batch_indexing, multiply, torch.max(dim=-1)).
 *   feat [B,C,M], weight [B,C,N,k], idx int64 rows of length idx_stride (first k used)
 *   out [B,C,N]; arg = FIRST index attaining the maximum (ties only matter where weight == 0,
 *   whose gradient the ReLU of weight_net masks anyway)
 * ------------------------------------------------------------------------------------------ */
int oracle_pointconv_dw_fwd(const float *feat, const float *weight, const int64_t *idx, int idx_stride,
                            float *out, unsigned char *arg, int B, int C, int M, int N, int k)
{
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int n = 0; n < N; ++n) {
                const int64_t *ir = idx + ((size_t)b * N + n) * idx_stride;
                const float *w = weight + (((size_t)b * C + c) * N + n) * k;
                const float *f = feat + ((size_t)b * C + c) * M;
                float best = -INFINITY;
                int bj = 0;
                for (int j = 0; j < k; ++j) {
                    if (ir[j] < 0 || ir[j] >= M) return -1;
                    float p = f[ir[j]] * w[j];
                    if (p > best) { best = p; bj = j; }
                }
                out[((size_t)b * C + c) * N + n] = best;
                arg[((size_t)b * C + c) * N + n] = (unsigned char)bj;
            }
    return 0;
}

int oracle_pointconv_dw_bwd(const float *gout, const float *feat, const float *weight, const int64_t *idx,
                            int idx_stride, const unsigned char *arg, float *gfeat, float *gweight,
                            int B, int C, int M, int N, int k)
{
    /* accumulates into gfeat / gweight (caller zero-fills) */
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int n = 0; n < N; ++n) {
                size_t e = ((size_t)b * C + c) * N + n;
                int j = arg[e];
                int64_t m = idx[((size_t)b * N + n) * idx_stride + j];
                gfeat[((size_t)b * C + c) * M + m] += gout[e] * weight[e * k + j];
                gweight[e * k + j] += gout[e] * feat[((size_t)b * C + c) * M + m];
            }
    return 0;
}

/* adjoint of oracle_knn_interp_fwd w.r.t. the features (accumulates into gfeat, caller zero-fills) */
int oracle_knn_interp_bwd(const float *in_xyz, const float *gout, const float *q_xyz,
                          const int64_t *knn, float *gfeat, int B, int C, int M, int Nq, int k)
{
    if (k > ORACLE_MAX_K) return -1;
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Nq; ++q) {
            float wgt[ORACLE_MAX_K], wsum = 0.0f;
            const int64_t *ix = knn + ((size_t)b * Nq + q) * k;
            for (int j = 0; j < k; ++j) {
                float s = 0.0f;
                for (int a = 0; a < 3; ++a) {
                    float d = in_xyz[((size_t)b * 3 + a) * M + ix[j]] - q_xyz[((size_t)b * 3 + a) * Nq + q];
                    s += d * d;
                }
                float dist = sqrtf(s);
                if (dist < 1e-8f) dist = 1e-8f;
                wgt[j] = 1.0f / dist;
                wsum += wgt[j];
            }
            for (int c = 0; c < C; ++c)
                for (int j = 0; j < k; ++j)
                    gfeat[((size_t)b * C + c) * M + ix[j]] += gout[((size_t)b * C + c) * Nq + q] * (wgt[j] / wsum);
        }
    return 0;
}

/* adjoint of oracle_knn_interp_fwd w.r.t. the coordinates: models/utils.py:138-146 differentiated the way
 * torch.autograd does it (linalg.norm: 0 at 0; clamp(1e-8): passes where the norm >= 1e-8).
 * g_in [B,3,M] accumulates (caller zero-fills), g_q [B,3,Nq] is written. */
int oracle_knn_interp_bwd_xyz(const float *in_xyz, const float *feat, const float *gout, const float *q_xyz,
                              const int64_t *knn, float *g_in, float *g_q, int B, int C, int M, int Nq, int k)
{
    if (k > ORACLE_MAX_K) return -1;
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Nq; ++q) {
            double w[ORACLE_MAX_K], raw[ORACLE_MAX_K], d[ORACLE_MAX_K][3], a[ORACLE_MAX_K], wsum = 0.0, s_tot = 0.0;
            const int64_t *ix = knn + ((size_t)b * Nq + q) * k;
            for (int j = 0; j < k; ++j) {
                float s = 0.0f;
                for (int ax = 0; ax < 3; ++ax) {
                    float df = in_xyz[((size_t)b * 3 + ax) * M + ix[j]] - q_xyz[((size_t)b * 3 + ax) * Nq + q];
                    d[j][ax] = df;
                    s += df * df;
                }
                raw[j] = sqrtf(s);
                w[j] = 1.0 / (raw[j] < 1e-8f ? 1e-8f : (float)raw[j]);
                wsum += w[j];
                a[j] = 0.0;
            }
            for (int c = 0; c < C; ++c) {
                double g = gout[((size_t)b * C + c) * Nq + q], o = 0.0;
                for (int j = 0; j < k; ++j) {
                    double f = feat[((size_t)b * C + c) * M + ix[j]];
                    o += f * (w[j] / wsum);
                    a[j] += g * f;
                }
                s_tot += g * o;
            }
            double gq[3] = {0.0, 0.0, 0.0};
            for (int j = 0; j < k; ++j) {
                if (!(raw[j] >= 1e-8f)) continue;
                double gd = -((a[j] - s_tot) / wsum) * (w[j] * w[j]);
                for (int ax = 0; ax < 3; ++ax) {
                    double v = gd * (d[j][ax] / raw[j]);
                    g_in[((size_t)b * 3 + ax) * M + ix[j]] += (float)v;
                    gq[ax] -= v;
                }
            }
            for (int ax = 0; ax < 3; ++ax) g_q[((size_t)b * 3 + ax) * Nq + q] = (float)gq[ax];
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * input of the point cost-volume lookup: follows models/camliraft_l_core.py:62-76
 *   out[b,0:3,n,j] = xyz2[b,:,knn[b,n,j]] - xyz1[b,:,n];  out[b,3,n,j] = cost[b,n,knn[b,n,j]]
 * ------------------------------------------------------------------------------------------ */
int oracle_corr3d_gather_fwd(const float *xyz1, const float *xyz2, const float *cost, const int64_t *knn,
                             float *out, int B, int N, int M, int k)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n)
            for (int j = 0; j < k; ++j) {
                int64_t m = knn[((size_t)b * N + n) * k + j];
                if (m < 0 || m >= M) return -1;
                for (int a = 0; a < 3; ++a)
                    out[(((size_t)b * 4 + a) * N + n) * k + j] =
                        xyz2[((size_t)b * 3 + a) * M + m] - xyz1[((size_t)b * 3 + a) * N + n];
                out[(((size_t)b * 4 + 3) * N + n) * k + j] = cost[((size_t)b * N + n) * M + m];
            }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * PointConv neighbourhood mixing: follows models/point_conv.py:60-66
 *   out[b,n,w,ch] = sum_j wgt[b,w,n,j] * feat_cl[b, idx[b,n,j], ch]
 *   feat_cl [B,M,CH], wgt [B,Wn,N,k], idx rows of stride idx_stride, out [B,N,Wn,CH]
 * ------------------------------------------------------------------------------------------ */
int oracle_pointconv_mix_fwd(const float *feat_cl, const float *wgt, const int64_t *idx, int idx_stride,
                             float *out, int B, int M, int N, int CH, int Wn, int k)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n)
            for (int w = 0; w < Wn; ++w)
                for (int ch = 0; ch < CH; ++ch) {
                    float acc = 0.0f;
                    for (int j = 0; j < k; ++j) {
                        int64_t m = idx[((size_t)b * N + n) * idx_stride + j];
                        if (m < 0 || m >= M) return -1;
                        acc += wgt[(((size_t)b * Wn + w) * N + n) * k + j] * feat_cl[((size_t)b * M + m) * CH + ch];
                    }
                    out[(((size_t)b * N + n) * Wn + w) * CH + ch] = acc;
                }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * PointPWC cost volume pieces, follow models/camlipwc_l_core.py:66-101 with the first cost-MLP layer split by
 * input block (W1 . cat[f1, f2_knn, d] = W1a.f1 + W1b.f2_knn + W1c.d; the caller supplies a = W1a.f1,
 * bm = W1b.f2 and e = W1c.d + b1):
 *   pair : h1[b,c,n,j] = leaky(a[b,c,n] + bm[b,c,idx[b,n,j]] + e[b,c,n,j])
 *   ksum : out[b,c,n]  = sum_j w[b,c,n,j] * h[b,c,n,j]                       (:79-81, torch.sum(weights2 * p2p_cost, 3))
 *   gather_wsum : out[b,c,n] = sum_j w[b,c,n,j] * feat[b,c,idx[b,n,j]]       (:97-101)
 * ------------------------------------------------------------------------------------------ */
int oracle_pwc3d_pair_fwd(const float *a, const float *bm, const float *e, const int64_t *idx, float *h1,
                          int B, int C, int M, int N, int k, float slope)
{
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int n = 0; n < N; ++n)
                for (int j = 0; j < k; ++j) {
                    int64_t m = idx[((size_t)b * N + n) * k + j];
                    if (m < 0 || m >= M) return -1;
                    size_t o = (((size_t)b * C + c) * N + n) * k + j;
                    float v = a[((size_t)b * C + c) * N + n] + bm[((size_t)b * C + c) * M + m] + e[o];
                    h1[o] = v > 0.0f ? v : slope * v;
                }
    return 0;
}

int oracle_ksum_fwd(const float *w, const float *h, float *out, int B, int C, int N, int k)
{
    for (size_t r = 0; r < (size_t)B * C * N; ++r) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc += (double)w[r * k + j] * h[r * k + j];
        out[r] = (float)acc;
    }
    return 0;
}

int oracle_gather_wsum_fwd(const float *w, const float *feat, const int64_t *idx, float *out,
                           int B, int C, int M, int N, int k)
{
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int n = 0; n < N; ++n) {
                double acc = 0.0;
                for (int j = 0; j < k; ++j) {
                    int64_t m = idx[((size_t)b * N + n) * k + j];
                    if (m < 0 || m >= M) return -1;
                    acc += (double)w[(((size_t)b * C + c) * N + n) * k + j] * feat[((size_t)b * C + c) * M + m];
                }
                out[((size_t)b * C + c) * N + n] = (float)acc;
            }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * convex up-sampling: follows models/utils.py:191-204 (softmax over the 9 taps of the 3x3 unfold)
 *   flow [B,2,h,w], mask [B,9*S*S,h,w] (already scaled), out [B,2,h*S,w*S]
 * ------------------------------------------------------------------------------------------ */
int oracle_convex_upsample_fwd(const float *flow, const float *mask, float *out, int B, int h, int w, int S)
{
    size_t plane = (size_t)h * w;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                for (int i = 0; i < S; ++i)
                    for (int j = 0; j < S; ++j) {
                        float p[9], mx = -INFINITY, den = 0.0f;
                        for (int k = 0; k < 9; ++k) {
                            p[k] = mask[((size_t)b * 9 * S * S + (size_t)k * S * S + i * S + j) * plane + (size_t)y * w + x];
                            if (p[k] > mx) mx = p[k];
                        }
                        for (int k = 0; k < 9; ++k) { p[k] = expf(p[k] - mx); den += p[k]; }
                        for (int c = 0; c < 2; ++c) {
                            float acc = 0.0f;
                            for (int k = 0; k < 9; ++k) {
                                int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
                                float f = (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                              ? flow[((size_t)b * 2 + c) * plane + (size_t)yy * w + xx] * (float)S : 0.0f;
                                acc += (p[k] / den) * f;
                            }
                            out[(((size_t)b * 2 + c) * h * S + (size_t)y * S + i) * ((size_t)w * S) + (size_t)x * S + j] = acc;
                        }
                    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * set-conv neighbour-weight network: follows models/point_conv.py:110-121 (knn_offset ->
 * weight_net = MLP2d(3,[8,32,C], act='relu'), models/mlp.py:100-162: conv1x1 + bias, ReLU).
 * Arithmetic order pinned for the HIP kernel: each layer = bias, then an input-ordered fmaf chain.
 *   xyz [B,3,M], centres [B,3,N], idx rows of stride idx_stride; out [B,C,N,k]; h2_out [B,32,N,k] or NULL
 * ------------------------------------------------------------------------------------------ */
static void weightnet_hidden(const float *xyz, const float *centres, const int64_t *idx, int idx_stride,
                             const float *w1, const float *b1, const float *w2, const float *b2,
                             int b, int M, int N, int n, int j, float *h2)
{
    int64_t m = idx[((size_t)b * N + n) * idx_stride + j];
    float off[3], h1[8];
    for (int d = 0; d < 3; ++d) off[d] = xyz[((size_t)b * 3 + d) * M + m] - centres[((size_t)b * 3 + d) * N + n];
    for (int i = 0; i < 8; ++i) {
        float a = b1[i];
        for (int d = 0; d < 3; ++d) a = fmaf(w1[i * 3 + d], off[d], a);
        h1[i] = a > 0.0f ? a : 0.0f;
    }
    for (int q = 0; q < 32; ++q) {
        float a = b2[q];
        for (int i = 0; i < 8; ++i) a = fmaf(w2[q * 8 + i], h1[i], a);
        h2[q] = a > 0.0f ? a : 0.0f;
    }
}

int oracle_weightnet_fwd(const float *xyz, const float *centres, const int64_t *idx, int idx_stride,
                         const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                         const float *b3, float *out, float *h2_out, int B, int C, int M, int N, int k)
{
    size_t NK = (size_t)N * k;
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n)
            for (int j = 0; j < k; ++j) {
                int64_t m = idx[((size_t)b * N + n) * idx_stride + j];
                if (m < 0 || m >= M) return -1;
                float h2[32];
                weightnet_hidden(xyz, centres, idx, idx_stride, w1, b1, w2, b2, b, M, N, n, j, h2);
                if (h2_out)
                    for (int q = 0; q < 32; ++q) h2_out[((size_t)b * 32 + q) * NK + (size_t)n * k + j] = h2[q];
                for (int c = 0; c < C; ++c) {
                    float a = b3[c];
                    for (int q = 0; q < 32; ++q) a = fmaf(w3[c * 32 + q], h2[q], a);
                    out[((size_t)b * C + c) * NK + (size_t)n * k + j] = a > 0.0f ? a : 0.0f;
                }
            }
    return 0;
}

/* backward of the whole network wrt its six parameters (double accumulation; masks from the fp32
 * forward chains above):  g3 = gout*(pre3>0), gw3 += g3 h2^T, gb3 += g3, gh2 = W3^T g3, g2 = gh2*(h2>0),
 * gw2 += g2 h1^T, gb2 += g2, gh1 = W2^T g2, g1 = gh1*(h1>0), gw1 += g1 d^T, gb1 += g1.
 * grads = [gw1 (24), gb1 (8), gw2 (256), gb2 (32), gw3 (C*32), gb3 (C)] concatenated. */
int oracle_weightnet_bwd(const float *xyz, const float *centres, const int64_t *idx, int idx_stride,
                         const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                         const float *b3, const float *gout, double *grads, int B, int C, int M, int N, int k)
{
    size_t NK = (size_t)N * k;
    double *gw1 = grads, *gb1 = gw1 + 24, *gw2 = gb1 + 8, *gb2 = gw2 + 256, *gw3 = gb2 + 32, *gb3 = gw3 + (size_t)C * 32;
    for (size_t i = 0; i < (size_t)24 + 8 + 256 + 32 + (size_t)C * 33; ++i) grads[i] = 0.0;
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n)
            for (int j = 0; j < k; ++j) {
                int64_t m = idx[((size_t)b * N + n) * idx_stride + j];
                if (m < 0 || m >= M) return -1;
                float off[3], h1[8], h2[32];
                for (int d = 0; d < 3; ++d) off[d] = xyz[((size_t)b * 3 + d) * M + m] - centres[((size_t)b * 3 + d) * N + n];
                for (int i = 0; i < 8; ++i) {
                    float a = b1[i];
                    for (int d = 0; d < 3; ++d) a = fmaf(w1[i * 3 + d], off[d], a);
                    h1[i] = a > 0.0f ? a : 0.0f;
                }
                for (int q = 0; q < 32; ++q) {
                    float a = b2[q];
                    for (int i = 0; i < 8; ++i) a = fmaf(w2[q * 8 + i], h1[i], a);
                    h2[q] = a > 0.0f ? a : 0.0f;
                }
                double g2[32], g1[8];
                for (int q = 0; q < 32; ++q) g2[q] = 0.0;
                for (int c = 0; c < C; ++c) {
                    float a = b3[c];
                    for (int q = 0; q < 32; ++q) a = fmaf(w3[c * 32 + q], h2[q], a);
                    if (!(a > 0.0f)) continue;
                    double g = gout[((size_t)b * C + c) * NK + (size_t)n * k + j];
                    gb3[c] += g;
                    for (int q = 0; q < 32; ++q) {
                        gw3[c * 32 + q] += g * h2[q];
                        g2[q] += g * w3[c * 32 + q];
                    }
                }
                for (int i = 0; i < 8; ++i) g1[i] = 0.0;
                for (int q = 0; q < 32; ++q) {
                    if (!(h2[q] > 0.0f)) continue;
                    gb2[q] += g2[q];
                    for (int i = 0; i < 8; ++i) {
                        gw2[q * 8 + i] += g2[q] * h1[i];
                        g1[i] += g2[q] * w2[q * 8 + i];
                    }
                }
                for (int i = 0; i < 8; ++i) {
                    if (!(h1[i] > 0.0f)) continue;
                    gb1[i] += g1[i];
                    for (int d = 0; d < 3; ++d) gw1[i * 3 + d] += g1[i] * off[d];
                }
            }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * bilinear sampling at pixel positions: follows models/utils.py:262-269 (normalise to [-1,1]) and
 * F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True): un-normalise, the four
 * corners nw/ne/sw/se with weights (x1-x)(y1-y) ..., out-of-image corners skipped.
 *   feat [B,C,H,W], uv [B,2,N], out [B,C,N]
 * ------------------------------------------------------------------------------------------ */
int oracle_bilinear_sample_fwd(const float *feat, const float *uv, float *out, int B, int C, int H, int W, int N)
{
    size_t plane = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            float gx = 2.0f * uv[((size_t)b * 2 + 0) * N + n] / (float)(W - 1) - 1.0f;
            float gy = 2.0f * uv[((size_t)b * 2 + 1) * N + n] / (float)(H - 1) - 1.0f;
            float x = ((gx + 1.0f) / 2.0f) * (float)(W - 1), y = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
            float x0 = floorf(x), y0 = floorf(y);
            float wx1 = x - x0, wy1 = y - y0, wx0 = (x0 + 1.0f) - x, wy0 = (y0 + 1.0f) - y;
            for (int c = 0; c < C; ++c) {
                const float *p = feat + ((size_t)b * C + c) * plane;
                float acc = 0.0f;
                for (int t = 0; t < 4; ++t) {
                    float fx = x0 + (float)(t & 1), fy = y0 + (float)(t >> 1);
                    if (!(fx >= 0.0f && fx < (float)W && fy >= 0.0f && fy < (float)H)) continue;
                    float w = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
                    acc += p[(size_t)fy * W + (size_t)fx] * w;
                }
                out[((size_t)b * C + c) * N + n] = acc;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * flow read-out of the IDS wrapper: follows models/ids.py:36-67 (paral2persp) as used at
 * models/camliraft.py:108-110:  out = paral2persp(pc1 + flow) - origin;  [B,3,N] tensors, f/cx/cy [B]
 * ------------------------------------------------------------------------------------------ */
int oracle_ids_flow_fwd(const float *pc1, const float *flow, const float *origin, const float *f, const float *cx,
                        const float *cy, float *out, float rw, float rh, float rm, float aw, float ah, int B, int N)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            size_t i0 = ((size_t)b * 3 + 0) * N + n, i1 = ((size_t)b * 3 + 1) * N + n, i2 = ((size_t)b * 3 + 2) * N + n;
            float u = ((pc1[i0] + flow[i0]) + aw) / rw, v = ((pc1[i1] + flow[i1]) + ah) / rh, d = (pc1[i2] + flow[i2]) / rm;
            float z = expf((d - 1.0f) / f[b]);
            out[i0] = (u - cx[b]) * z / f[b] - origin[i0];
            out[i1] = (v - cy[b]) * z / f[b] - origin[i1];
            out[i2] = z - origin[i2];
        }
    return 0;
}

int oracle_version(void) { return 1; }

/* ------------------------------------------------------------------------------------------
 * input side: persp2paral follows models/ids.py:4-33 expression by expression (fp32, unfused);
 * pad_normalize follows models/utils.py:7-15 (replicate padding: width split left/right, height at the
 * bottom) + models/camliraft.py:41-46 ((x - mean) / std with the ImageNet constants)
 *   pcs [B,6,N] -> out1, out2 [B,3,N];  intr [B,3] = f, cx, cy
 *   images [B,6,H,W] -> out1, out2 [B,3,Hp,Wp]
 * ------------------------------------------------------------------------------------------ */
int oracle_persp2paral(const float *pcs, const float *intr, float *out1, float *out2, int B, int N,
                       float rw, float rh, float rmin, float aw, float ah)
{
    for (int b = 0; b < B; ++b)
        for (int cloud = 0; cloud < 2; ++cloud)
            for (int n = 0; n < N; ++n) {
                const float f = intr[b * 3], cx = intr[b * 3 + 1], cy = intr[b * 3 + 2];
                const float *src = pcs + ((size_t)b * 6 + 3 * cloud) * N;
                float x = src[n], y = src[(size_t)N + n], z = src[2 * (size_t)N + n];
                float u = cx + (f / z) * x;
                float v = cy + (f / z) * y;
                float d = f * logf(z) + 1.0f;
                float *dst = (cloud == 0 ? out1 : out2) + (size_t)b * 3 * N;
                dst[n] = u * rw - aw;
                dst[(size_t)N + n] = v * rh - ah;
                dst[2 * (size_t)N + n] = d * rmin;
            }
    return 0;
}

/* project_pc2image follows models/utils.py:234-259, then the callers' feature-grid rescale
 * (camliraft_core.py:51-56, camlipwc_core.py:112-114):  pc [B,3,N] -> uv [B,2,N]
 *   perspective: u = (cx_b + (f_b / z) * x) * sx   (intr [B,3] = f, cx, cy);   parallel: u = (x + cx) * sx */
int oracle_project_pc2image(const float *pc, const float *intr, float *uv, int B, int N, int perspective,
                            float cx, float cy, float sx, float sy)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const float *src = pc + (size_t)b * 3 * N;
            float x = src[n], y = src[(size_t)N + n], z = src[2 * (size_t)N + n];
            float u, v;
            if (perspective) {
                float f = intr[b * 3], cxb = intr[b * 3 + 1], cyb = intr[b * 3 + 2];
                u = cxb + (f / z) * x;
                v = cyb + (f / z) * y;
            } else {
                u = x + cx;
                v = y + cy;
            }
            uv[(size_t)b * 2 * N + n] = u * sx;
            uv[(size_t)b * 2 * N + N + n] = v * sy;
        }
    return 0;
}

int oracle_pad_normalize(const float *images, float *out1, float *out2, int B, int H, int W, int Hp, int Wp, int left,
                         const float *mean3, const float *std3)
{
    for (int b = 0; b < B; ++b)
        for (int c6 = 0; c6 < 6; ++c6)
            for (int y = 0; y < Hp; ++y)
                for (int x = 0; x < Wp; ++x) {
                    int sy = y < H ? y : H - 1;
                    int sx = x - left;
                    if (sx < 0) sx = 0;
                    if (sx > W - 1) sx = W - 1;
                    int c = c6 % 3;
                    float v = images[(((size_t)b * 6 + c6) * H + sy) * W + sx];
                    float *dst = (c6 < 3 ? out1 : out2) + (((size_t)b * 3 + c) * Hp + y) * Wp + x;
                    *dst = (v - mean3[c]) / std3[c];
                }
    return 0;
}

/*
 * camli_hip.h -- C ABI of libcamli_hip.so, the MI355X (gfx950) implementation of the
 * CamLiFlow / CamLiRAFT fused 2D+3D flow hot path.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers + sizes, no torch types
 *   - every pointer is a DEVICE pointer on the current HIP device; tensors are contiguous fp32
 *     unless stated, index tensors are int64
 *   - the caller allocates every output (and owns it); nothing is allocated inside
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); the reference launches
 *     on the legacy default stream -- callers here pass torch's current stream
 *   - return value: 0 = ok, negative = error (-22 invalid argument, -5 launch failure,
 *     -95 unsupported size); the message is available from camli_last_error_string()
 *     (thread-local).  Nothing throws.
 *
 * Citations are relative to the reference tree (MCG-NJU/CamLiFlow).
 */
#ifndef CAMLI_HIP_H
#define CAMLI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library version: major*10000 + minor*100 + patch */
int camli_version(void);
/* message of the last failing call on this thread ("" if none) */
const char *camli_last_error_string(void);

/*
 * k-nearest-neighbour (brute force, ascending distance), D in {2,3}, 1 <= k <= 64.
 * Replaces _k_nearest_neighbor_cuda: models/csrc/k_nearest_neighbor/k_nearest_neighbor.cpp:6-24,
 * kernels k_nearest_neighbor_kernel.cu:9-95; bound by models/csrc/wrapper.py:106-127.
 *   input [B,M,D], query [B,Nq,D] (channel-last), out_idx int64 [B,Nq,k].
 * Bit-exact to the reference kernel's in-order insertion semantics (ties included); requires
 * finite coordinates with squared distances < 1e9.
 */
int camli_knn(const float *input, const float *query, int64_t *out_idx,
              int B, int M, int Nq, int D, int k, void *stream);

/*
 * furthest-point sampling from seed index 0.
 * Replaces _furthest_point_sampling_cuda: models/csrc/furthest_point_sampling/
 * furthest_point_sampling.cpp:5-16, kernel furthest_point_sampling_kernel.cu:34-85; bound by
 * models/csrc/wrapper.py:75-103.
 *   xyz [B,N,3], out_idx int64 [B,n_samples], 1 <= n_samples <= N <= 24576.
 * No scratch buffer: running distances live in registers (the reference allocates a [B,N] temp,
 * furthest_point_sampling.cpp:12).  Ties -> lowest index.
 */
/*
 * The same search over NESTED candidate prefixes in one scan: out_levels[l] [B,Nq,k] = the k nearest among the
 * first sizes[l] rows of every cloud (sizes strictly descending, sizes[0] = M; L <= 4; HOST arrays).  The FPS pyramid
 * is a chain of prefixes (models/utils.py:121-125) and the reference searches each level separately
 * (camliraft_l_core.py:62-66, four calls per GRU iteration); the sequential insertion semantics make the k-list after
 * the first M_l candidates the level-l answer.  Results are identical to L separate camli_knn calls.
 */
int camli_knn_prefixes(const float *input, const float *query, int64_t *const *out_levels, const int *sizes,
                       int L, int B, int M, int Nq, int D, int k, void *stream);
/* The same search given the out_levels of an EARLIER call (same sizes, same k) on the same clouds, moved since: the GRU loop
 * (camliraft_l_core.py:62-66, raft3d flow loop) repeats it every iteration on the back-warped target cloud.  The K earlier
 * neighbours of a query bound its k-th distance now, and the scan queues nothing beyond that bound.  Results are identical to
 * camli_knn_prefixes whatever the prior holds (indices outside a level, or a prior row that repeats an index, switch the bound off for that query); prior_levels == NULL is
 * camli_knn_prefixes. */
int camli_knn_prefixes_prior(const float *input, const float *query, int64_t *const *out_levels,
                             const int64_t *const *prior_levels, const int *sizes, int L, int B, int M, int Nq, int D, int k,
                             void *stream);

/* camli_fps tie rule: the LOWEST index among equal maxima (the reference's Python path, wrapper.py:83-96); the
 * reference's CUDA reduction tree keeps a different tied candidate (kernel.cu:5-10) -- see csrc/hip/fps.hip. */
int camli_fps(const float *xyz, int64_t *out_idx, int B, int N, int n_samples, void *stream);

/*
 * local correlation (PWC cost volume), max displacement md, Dd = 2*md+1.
 * Replaces _correlation_forward_cuda / _correlation_backward_cuda:
 * models/csrc/correlation/correlation.cpp:11-35, kernels correlation_forward_kernel.cu:11-55,
 * correlation_backward_kernel.cu:4-89; bound by models/csrc/wrapper.py:18-57.
 *   in1, in2 NHWC [B,H,W,C]; out NCHW [B,Dd*Dd,H,W] (fully written, no pre-zeroing needed).
 *   backward: gout NCHW; g1, g2 NHWC [B,H,W,C] (fully written) -- the layout wrapper.py:34-35
 *   produces after its permute+contiguous, written directly.
 */
int camli_corr2d_fwd(const float *in1_nhwc, const float *in2_nhwc, float *out_nchw,
                     int B, int C, int H, int W, int md, void *stream);
int camli_corr2d_bwd(const float *gout_nchw, const float *in1_nhwc, const float *in2_nhwc,
                     float *g1_nhwc, float *g2_nhwc,
                     int B, int C, int H, int W, int md, void *stream);

/*
 * All-pairs cost-volume pyramid, build and adjoint (internal composite op; the reference composes it from
 * torch.matmul, a division and three avg_pool2d passes over the volume: models/raft_core.py:52-68).
 *   f1 [B,C,P] source features, f2_levels[l] [B,C,P_l] the target features pooled l times (avg_pool2d(2,2) acts
 *   on the target pixel only and is linear, so level l = f1^T . pool_l(f2) * scale; HOST arrays of DEVICE pointers),
 *   vol_levels[l] [B,P,P_l] fully written, scale = 1/sqrt(C).  fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *   bwd: g_f1 [B,C,P] = scale * sum_l gV_l . f2_l^T and g_f2_levels[l] [B,C,P_l] = scale * f1 . gV_l, both fully
 *   written; the caller un-pools g_f2_levels on the small maps.
 */
int camli_allpairs_build_fwd(const float *f1, const float *const *f2_levels, float *const *vol_levels,
                             const int *p_levels, int L, int B, int C, int P, float scale, void *stream);
int camli_allpairs_build_bwd(const float *f1, const float *const *f2_levels, const float *const *gvol_levels,
                             const int *p_levels, int L, float *g_f1, float *const *g_f2_levels,
                             int B, int C, int P, float scale, void *stream);

/*
 * All-pairs cost-volume pyramid lookup and its adjoint (internal composite op of the cores; the
 * reference composes it from grid_sample/cat/permute: models/raft_core.py:70-107, and its
 * backward from grid_sampler_2d_backward).
 *   vols[l]  device pointer to level l, [B*P, hs[l], ws[l]], P = h*w   (vols/hs/ws are HOST arrays)
 *   coords   [B,2,h,w] (x,y) in level-0 pixels;  out / gout [B, L*(2r+1)^2, h, w];  r must be 4
 *   channel l*81 + i*9 + j samples level l at (x/2^l + (i-r), y/2^l + (j-r)), bilinear,
 *   align_corners=True, zeros padding.
 *   bwd ACCUMULATES into gvols (caller zero-fills once, then may call once per GRU iteration);
 *   coords carry no gradient (they are built from detached flow, raft_core.py:248).
 */
/* Adjoint of the pyramid build that skips the never-visited part of the gradient volume.  marks[l] is
 * [B][ceil(P/32)][ceil(p_levels[l]/32)] bytes: non-zero where camli_allpairs_lookup_bwd_marked added a window into that
 * block of 32 source pixels x 32 target pixels of gvol_levels[l] (any superset of the non-zero blocks is valid).  A K step
 * of the two adjoint GEMMs whose gradient tile holds no marked block is neither loaded nor multiplied; results equal
 * camli_allpairs_build_bwd bit for bit (only exact zeros are skipped).  Replaces the same reference lines. */
int camli_allpairs_build_bwd_marked(const float *f1, const float *const *f2_levels, const float *const *gvol_levels,
                                    const int *p_levels, int L, float *g_f1, float *const *g_f2_levels, int B, int C,
                                    int P, float scale, const unsigned char *const *marks, void *stream);

/* The same adjoint with the g_f2 GEMMs of the coarse levels split over K.  Those levels are a few output tiles with the
 * longest and densest K loops (8160 source pixels; level 3 of a 68x120 map is ONE 120-column tile per sample), so unsplit
 * the launch lasts as long as one workgroup walking 255 K steps.  Level l is cut into min(2^l, 8) ranges of K steps (each at
 * least 16 steps of 32 source pixels), every range writes its partial [B,C,P_l] into `workspace`, and the parts of a level are added left to
 * right -- deterministic; equal to camli_allpairs_build_bwd(_marked) up to fp32 summation order on levels >= 1, bit for bit
 * on level 0 and g_f1.  marks may be NULL (every gradient tile is examined).  workspace: 16-byte aligned device memory of
 * camli_allpairs_build_bwd_workspace_bytes(...) bytes; NULL / too small runs the unsplit form. */
int64_t camli_allpairs_build_bwd_workspace_bytes(const int *p_levels, int L, int B, int C, int P);
int camli_allpairs_build_bwd_splitk(const float *f1, const float *const *f2_levels, const float *const *gvol_levels,
                                    const int *p_levels, int L, float *g_f1, float *const *g_f2_levels, int B, int C,
                                    int P, float scale, const unsigned char *const *marks, void *workspace,
                                    int64_t workspace_bytes, void *stream);

int camli_allpairs_lookup_fwd(const float *const *vols, const int *hs, const int *ws, int L,
                              const float *coords, float *out, int B, int h, int w, int r, void *stream);
int camli_allpairs_lookup_bwd(float *const *gvols, const int *hs, const int *ws, int L,
                              const float *coords, const float *gout, int B, int h, int w, int r, void *stream);

/* camli_allpairs_lookup_bwd plus visit marks (see camli_allpairs_build_bwd_marked); the caller zeroes marks[l] once per
 * backward pass, together with the gradient pyramid. */
int camli_allpairs_lookup_bwd_marked(float *const *gvols, const int *hs, const int *ws, int L, const float *coords,
                                     const float *gout, int B, int h, int w, int r, unsigned char *const *marks,
                                     void *stream);

/* Puts a gradient pyramid that is kept across steps back to all-zero after camli_allpairs_build_bwd_marked has read it:
 * zeroes every marked 32 x 32 block of gvols[l] ([B,P,p_levels[l]]) and clears its mark -- the unmarked blocks were never
 * written.  (Zero-filling the whole pyramid before every backward pass was 5.7 GB of stores per step at batch 8.) */
int camli_allpairs_clear_marked(float *const *gvols, const int *p_levels, int L, unsigned char *const *marks, int B, int P,
                                void *stream);

/*
 * Depth-wise set-conv core and adjoint (internal composite op; the reference composes it from
 * gather * weight_net(...) -> max, models/point_conv.py:122-128).
 *   feat [B,C,M]; weight [B,C,N,k]; idx int64, row n of batch b at idx + (b*N+n)*idx_stride, first k
 *   entries used (lets a wider precomputed KNN tensor be sliced without a copy, point_conv.py:116-120)
 *   out [B,C,N] = max_j feat[b,c,idx[b,n,j]] * weight[b,c,n,j]; arg uint8 [B,C,N] = first arg-max j;
 *   optional compact record for the adjoint: wsel [B,C,N] = weight at arg, msel int32 [B,C,N] = idx at
 *   arg (both or neither).  k <= 255.
 *   bwd (from the compact record): gfeat [B,C,M] = scatter of gout*wsel (FULLY written, no zero-fill
 *   needed; may be NULL), gwsel [B,C,N] = gout*feat[msel] (fully written; may be NULL).
 *   expand: dense gweight [B,C,N,k] (fully written) = sum over the n_calls <= 64 calls of one pass of
 *   one-hot(arg_i) * gwsel_i; gwsel_list / arg_list are HOST arrays of device pointers.
 *   k-major forms (round 3; camli_pointconv_dw_fwd_kmajor, expand with k_major = 1): weight_kn / gweight are laid out
 *   [B,C,k,N] and the neighbour table is idx_kn int32 [B,k,N] -- the neighbour slot is the slow axis, so a lane's k
 *   weights (and its k neighbour indices) are k coalesced rows that go straight from HBM into registers, no LDS
 *   transposition.  Needs k in {4,8,16,32}, M % 4 == 0, M <= 8192, feat 16-byte aligned; results identical.
 */
int camli_pointconv_dw_fwd(const float *feat, const float *weight, const int64_t *idx, int idx_stride,
                           float *out, unsigned char *arg, float *wsel, int *msel,
                           int B, int C, int M, int N, int k, void *stream);
int camli_pointconv_dw_fwd_kmajor(const float *feat, const float *weight_kn, const int *idx_kn, float *out,
                                  unsigned char *arg, float *wsel, int *msel, int B, int C, int M, int N, int k,
                                  void *stream);
int camli_pointconv_dw_bwd(const float *gout, const float *feat, const float *wsel, const int *msel,
                           float *gfeat, float *gwsel, int B, int C, int M, int N, void *stream);
/* the same adjoint with a FIXED summation order (no float atomics: integer LDS tickets decide whose turn it is to add into a
 * target): bit-reproducible, ~1.7x the time; selected under torch.use_deterministic_algorithms(True).  M <= 5461. */
/* gout read in place from a channel slice of a wider gradient (what the adjoint of a cat hands over): batch stride in floats,
 * >= C*N; served by the LDS row kernel (M <= 8192), CAMLI_ENOTSUP otherwise */
int camli_pointconv_dw_bwd_strided(const float *gout, int64_t gout_batch_stride, const float *feat, const float *wsel,
                                   const int *msel, float *gfeat, float *gwsel, int B, int C, int M, int N, void *stream);
int camli_pointconv_dw_bwd_ordered(const float *gout, const float *feat, const float *wsel, const int *msel,
                           float *gfeat, float *gwsel, int B, int C, int M, int N, void *stream);
int camli_pointconv_dw_expand(const float *const *gwsel_list, const unsigned char *const *arg_list, int n_calls,
                              float *gweight, int B, int C, int N, int k, int k_major, void *stream);

/*
 * batch_indexing, channel-first (models/utils.py:61-83): out[b,c,i] = data[b,c,idx[b,i]].
 *   data [B,C,M], idx int64 [B,I] (contiguous), out [B,C,I].  bwd: gdata [B,C,M] += (float atomics,
 *   caller zero-fills).
 */
int camli_gather_cf_fwd(const float *data, const int64_t *idx, float *out, int B, int C, int M, int I, void *stream);
int camli_gather_cf_bwd(const float *gout, const int64_t *idx, float *gdata, int B, int C, int M, int I, void *stream);
/*
 * The same adjoint without atomics, through the INVERSE map of idx (what torch's index_put_(accumulate=True)
 * derives with a device-wide sort on every call; here the host builds it once per index tensor):
 *   inv_order int32 [B*I]: the flat positions b*I + i sorted by (b, idx[b,i]) (stable);
 *   inv_offsets int32 [B*M + 1]: segment s = b*M + m covers inv_order[inv_offsets[s] .. inv_offsets[s+1]).
 * gdata is fully written (no zero-fill needed), fixed summation order.
 */
int camli_gather_cf_bwd_sorted(const float *gout, const int32_t *inv_order, const int32_t *inv_offsets, float *gdata,
                               int B, int C, int M, int I, void *stream);

/*
 * batch_indexing, channel-last (models/utils.py:85-104): out[b,i,:] = data[b,idx[b,i],:].
 *   data [B,M,C] (C = 1: the rank-2 form [B,M] of models/camliraft_l_core.py:70-74), idx int64 [B,I], out [B,I,C].
 *   bwd through the inverse map of idx (see camli_gather_cf_bwd_sorted): gdata [B,M,C] fully written, no atomics,
 *   fixed summation order (ascending i).
 */
int camli_gather_cl_fwd(const float *data, const int64_t *idx, float *out, int B, int C, int M, int I, void *stream);
int camli_gather_cl_bwd_sorted(const float *gout, const int32_t *inv_order, const int32_t *inv_offsets, float *gdata,
                               int B, int C, int M, int I, void *stream);

/*
 * knn_interpolation tail (models/utils.py:138-146) given the k <= 8 nearest inputs of every query:
 *   w_j = (1/max(|in_xyz[:,knn_j] - q|, 1e-8)) / sum_j(...);  out[b,c,q] = sum_j feat[b,c,knn_j] * w_j
 *   in_xyz [B,3,M], feat [B,C,M], q_xyz [B,3,Nq] channel-first; knn int64 rows of stride knn_stride.
 *   bwd: gradient w.r.t. feat (gfeat += with atomics, caller zero-fills).
 *   bwd_xyz: gradient w.r.t. the coordinates -- CamLiPWC back-warps with a live flow
 *   (models/camlipwc_core.py:172-179 -> models/utils.py:149-159), so both in_xyz and q_xyz can carry a
 *   gradient there: g_in_xyz [B,3,M] += (atomics, caller zero-fills; may be NULL), g_q_xyz [B,3,Nq] is
 *   written (may be NULL).  Follows torch's subgradients: the clamp passes where |.| >= 1e-8, the norm's
 *   gradient is 0 at 0.
 */
int camli_knn_interp_fwd(const float *in_xyz, const float *feat, const float *q_xyz, const int64_t *knn,
                         int knn_stride, float *out, int B, int C, int M, int Nq, int k, void *stream);
int camli_knn_interp_bwd(const float *in_xyz, const float *gout, const float *q_xyz, const int64_t *knn,
                         int knn_stride, float *gfeat, int B, int C, int M, int Nq, int k, void *stream);
/* Adjoint of the interpolation wrt the features without atomics: the weights of every (query, slot) pair once
 * (camli_knn_interp_weights, the arithmetic of camli_knn_interp_fwd: models/utils.py:138-146), then a segment sum per input point:
 * offsets [B*M+1] = CSR bounds of the (sample, input point) segments of the neighbour table's inverse map, q_sorted / w_sorted = the
 * query (inside its sample) and the weight of every pair in segment order; fixed summation order, gfeat is written, not added to. */
int camli_knn_interp_weights(const float *in_xyz, const float *q_xyz, const int64_t *knn, int knn_stride, float *w_out,
                             int B, int M, int Nq, int k, void *stream);
int camli_knn_interp_bwd_sorted(const float *gout, const float *w_sorted, const int *q_sorted, const int *offsets, float *gfeat,
                                int B, int C, int M, int Nq, void *stream);
int camli_knn_interp_bwd_xyz(const float *in_xyz, const float *feat, const float *gout, const float *q_xyz,
                             const int64_t *knn, int knn_stride, float *g_in_xyz, float *g_q_xyz,
                             int B, int C, int M, int Nq, int k, void *stream);

/*
 * input tensor of the point cost-volume lookup (models/camliraft_l_core.py:62-76):
 *   out[b,0:3,n,j] = xyz2[b,:,knn[b,n,j]] - xyz1[b,:,n];  out[b,3,n,j] = cost[b,n,knn[b,n,j]]
 *   xyz1 [B,3,N], xyz2 [B,3,M], cost [B,N,M], knn int64 [B,N,k] contiguous, out [B,4,N,k].
 *   bwd: gcost [B,N,M] += gout[b,3,n,j] (atomics; caller zero-fills).
 */
int camli_corr3d_gather_fwd(const float *xyz1, const float *xyz2, const float *cost, const int64_t *knn,
                            float *out, int B, int N, int M, int k, void *stream);
/*
 * All levels of that lookup in ONE launch when the target levels are nested prefixes of one cloud (the FPS pyramid):
 * xyz2 [B,3,M0] is the level-0 cloud, level l uses its first sizes[l] points and cost_levels[l] [B,N,sizes[l]];
 * out / gout [B,4,N,L*k], column l*k + j = neighbour j of level l (the concatenation the shared cost MLP consumes).
 * bwd ADDS gout[b,3,n,l*k+j] into gcost_levels[l][b,n,knn] with plain read-modify-writes (a point's neighbours are
 * distinct when sizes[l] >= k; required): the caller keeps one zero-initialised gradient volume per level for the whole
 * pass and calls this once per GRU iteration.  HOST arrays of DEVICE pointers; L <= 4.
 */
int camli_corr3d_gather_levels_fwd(const float *xyz1, const float *xyz2, const float *const *cost_levels,
                                   const int64_t *const *knn_levels, const int *sizes, int L, float *out,
                                   int B, int N, int M0, int k, void *stream);
int camli_corr3d_gather_levels_bwd(const float *gout, const int64_t *const *knn_levels, float *const *gcost_levels,
                                   const int *sizes, int L, int B, int N, int M0, int k, void *stream);
int camli_corr3d_gather_bwd(const float *gout, const int64_t *knn, float *gcost, int B, int N, int M, int k,
                            void *stream);

/*
 * The cost MLP of that lookup and the sum over the neighbours in one pass (models/camliraft_l_core.py:96-100:
 * `self.cost_mlp(lookup).sum(dim=-1)` per level; cost_mlp = MLP2d(4 -> hidden -> hidden, bias, ReLU)).  The hidden
 * activations ([B,hidden,N,L*k] twice) never reach memory.
 *   lookup [B,4,N,L*k] (camli_corr3d_gather_levels_fwd's output); w1 [hidden,4], b1 [hidden], w2 [hidden,hidden], b2 [hidden]
 *   out [B, L*hidden, N], channel l*hidden + o = sum_j relu(b2[o] + w2[o,:] . relu(b1 + w1 lookup[b,:,n,l*k+j]))
 *   bwd: glookup [B,4,N,L*k]: ONLY channel 3 (the cost-volume entry) is written -- the offsets are not differentiable
 *        on this path and camli_corr3d_gather_levels_bwd reads channel 3 only;
 *        gw1 / gb1 / gw2 / gb2 += the parameter gradients (per-workgroup partial tiles in `workspace`,
 *        camli_corr3d_mlp_bwd_workspace_bytes, added in a fixed order: no atomics)
 * Covered: L = 4, k = 16, hidden = 32, N % 8 == 0 (camli_corr3d_mlp_supported); anything else returns CAMLI_EINVAL.
 */
int camli_corr3d_mlp_supported(int levels, int k, int hidden, int N);
int camli_corr3d_mlp_fwd(const float *lookup, const float *w1, const float *b1, const float *w2, const float *b2, float *out,
                         int B, int N, int levels, int k, int hidden, void *stream);
int64_t camli_corr3d_mlp_bwd_workspace_bytes(int B, int N);
int camli_corr3d_mlp_bwd(const float *lookup, const float *gout, const float *w1, const float *b1, const float *w2,
                         const float *b2, float *glookup, float *gw1, float *gb1, float *gw2, float *gb2, float *workspace,
                         int B, int N, int levels, int k, int hidden, void *stream);

/*
 * PointPWC learnable cost volume, PWC-style Correlation3D (internal composite op; the reference materialises
 * [B, 2C+3, N, k] = cat(f1 expanded, gather(f2), dxyz) and runs MLP2d over it: models/camlipwc_l_core.py:53-106).
 * The first MLP layer is split by input block (see csrc/hip/pwc3d.hip); tensors are [B,C,N,k] with k fastest, k a
 * power of two <= 64, idx int64 [B,N,k] contiguous.
 *   pair_fwd : h1 = leaky_relu(a[b,c,n] + bm[b,c,idx[b,n,j]] + e[b,c,n,j], slope)      a [B,C,N], bm [B,C,M]
 *   pair_bwd : gpre = gh1 * leaky'(h1) (fully written; = the gradient of e, and the source of bm's gradient through
 *              camli_gather_cf_bwd_sorted), ga[b,c,n] = sum_j gpre (fully written)
 *   ksum     : out[b,c,n] = sum_j w * h;    bwd: gw = g*h, gh = g*w (either may be NULL)
 *   gather_wsum : out[b,c,n] = sum_j w[b,c,n,j] * feat[b,c,idx[b,n,j]]   feat [B,C,M];
 *              bwd: gw = g*feat[idx], t = g*w (-> camli_gather_cf_bwd_sorted gives the gradient of feat)
 */
int camli_pwc3d_pair_fwd(const float *a, const float *bm, const float *e, const int64_t *idx, float *h1,
                         int B, int C, int M, int N, int k, float slope, void *stream);
int camli_pwc3d_pair_bwd(const float *gh1, const float *h1, float *gpre, float *ga,
                         int B, int C, int N, int k, float slope, void *stream);
int camli_ksum_fwd(const float *w, const float *h, float *out, int B, int C, int N, int k, void *stream);
int camli_ksum_bwd(const float *g, const float *w, const float *h, float *gw, float *gh,
                   int B, int C, int N, int k, void *stream);
int camli_gather_wsum_fwd(const float *w, const float *feat, const int64_t *idx, float *out,
                          int B, int C, int M, int N, int k, void *stream);
int camli_gather_wsum_bwd(const float *g, const float *w, const float *feat, const int64_t *idx, float *gw, float *t,
                          int B, int C, int M, int N, int k, void *stream);

/*
 * PointConv neighbourhood mixing and adjoint (internal composite op; the reference composes it from a
 * channel-last gather + matmul, models/point_conv.py:60-66).
 *   feat_cl [B,M,CH] channel-last (CH = in_channels + 3); wgt [B,Wn,N,k] (Wn <= 16, the layout
 *   weight_net produces); idx int64 rows of stride idx_stride (first k used); out [B,N,Wn,CH]:
 *   out[b,n,w,ch] = sum_j wgt[b,w,n,j] * feat_cl[b, idx[b,n,j], ch]
 *   bwd: gfeat_cl [B,M,CH] += (float atomics, caller zero-fills; may be NULL),
 *        gwgt [B,Wn,N,k] fully written (may be NULL).
 */
int camli_pointconv_mix_fwd(const float *feat_cl, const float *wgt, const int64_t *idx, int idx_stride,
                            float *out, int B, int M, int N, int CH, int Wn, int k, void *stream);
int camli_pointconv_mix_bwd(const float *gout, const float *feat_cl, const float *wgt, const int64_t *idx,
                            int idx_stride, float *gfeat_cl, float *gwgt,
                            int B, int M, int N, int CH, int Wn, int k, void *stream);
/*
 * Atomic-free adjoint for k = 16, Wn = 16 (every PointConv of the reference): the per-neighbour row gradients
 * T[b,n,j,:] = sum_w wgt[b,w,n,j] * gout[b,n,w,:] go to `scratch` (camli_pointconv_mix_bwd_scratch_bytes), then
 * gfeat_cl[b,m,:] = sum of the T rows whose neighbour index is m, walked through the inverse map of idx
 * (inv_order int32 [B*N*k]: flat positions (b*N + n)*k + j sorted by (b, idx); inv_offsets int32 [B*M + 1]).
 * gfeat_cl is fully written (no zero-fill), gwgt as above; either may be NULL.  Bit-reproducible.
 */
int64_t camli_pointconv_mix_bwd_scratch_bytes(int B, int N, int CH, int k);
int camli_pointconv_mix_bwd_sorted(const float *gout, const float *feat_cl, const float *wgt, const int64_t *idx,
                                   int idx_stride, const int32_t *inv_order, const int32_t *inv_offsets, float *scratch,
                                   float *gfeat_cl, float *gwgt, int B, int M, int N, int CH, int Wn, int k, void *stream);

/*
 * Convex flow up-sampling and adjoint (internal composite op; the reference composes it from
 * softmax / unfold / sum / permute, models/utils.py:191-204, with the mask pre-scaled at raft_core.py:195).
 *   flow [B,2,h,w]; mask [B, 9*S*S, h, w] (raw: mask_scale is applied inside); out [B,2,h*S,w*S]; S in {4,8}
 *   out[b,c,y*S+i,x*S+j] = sum_k softmax_k(mask_scale*mask[b,k*S*S+i*S+j,y,x]) * S * flow[b,c,y+dy_k,x+dx_k]
 *   mask_bias: NULL, or [9*S*S] added to the mask before the scale (the bias of the mask head's last convolution,
 *   raft_core.py:187-189, folded in: one pass less over the mask); its gradient = per-channel sum of gmask.
 *   bwd: gmask fully written; gflow += (float atomics, caller zero-fills).
 */
int camli_convex_upsample_fwd(const float *flow, const float *mask, const float *mask_bias, float *out,
                              int B, int h, int w, int scale, float mask_scale, void *stream);
int camli_convex_upsample_bwd(const float *gout, const float *flow, const float *mask, const float *mask_bias, float *gflow,
                              float *gmask, int B, int h, int w, int scale, float mask_scale, void *stream);
/* the same with the first out_rows of the h*S fine rows kept only: out / gout [B,2,out_rows,w*S] -- the un-padding of a
 * bottom-padded image (models/utils.py:7-20 InputPadder) folded into the up-sampling; cropped rows carry no gradient */
int camli_convex_upsample_rows_fwd(const float *flow, const float *mask, const float *mask_bias, float *out, int B, int h,
                                   int w, int scale, int out_rows, float mask_scale, void *stream);
int camli_convex_upsample_rows_bwd(const float *gout, const float *flow, const float *mask, const float *mask_bias,
                                   float *gflow, float *gmask, int B, int h, int w, int scale, int out_rows, float mask_scale,
                                   void *stream);

/*
 * Elementwise halves of the convolutional GRU (models/raft_core.py:123-139 composes them from ~9 torch
 * kernels per half-step).  All tensors fp32 contiguous, P = h*w, C*P a multiple of 4.
 *   gates: z = sigmoid(pre_zr[:, :C] + ctx_zr[:, :C]); r = sigmoid(pre_zr[:, C:] + ctx_zr[:, C:]); rh = r*h
 *          pre_zr, ctx_zr [B,2C,P]; h, z, r, rh [B,C,P]
 *   blend: q = tanh(pre_q + ctx_q); h_new = (1 - z)*h + z*q        all [B,C,P]
 *   adjoints: gates_bwd(gz, grh, z, r, h) -> gpre_zr [B,2C,P] (= gradient of ctx_zr too), gh [B,C,P]
 *             blend_bwd(g, z, h, q)       -> gpre_q (= gradient of ctx_q too), gz, gh   (all fully written)
 */
int camli_gru_gates_fwd(const float *pre_zr, const float *ctx_zr, const float *h, float *z, float *r, float *rh,
                        int B, int C, int P, void *stream);
int camli_gru_gates_bwd(const float *gz, const float *grh, const float *z, const float *r, const float *h,
                        float *gpre_zr, float *gh, int B, int C, int P, void *stream);
/* camli_gru_gates_bwd_strided with gh += instead of gh = (the hidden state's gradient of a half-step has three producers:
 * the blend adjoint writes it, this call and the data gradient of the z|r convolution add into it). */
int camli_gru_gates_bwd_into(const float *gz, int64_t gz_batch_stride, const float *grh, int64_t grh_batch_stride, const float *z,
                             const float *r, const float *h, float *gpre_zr, float *gh, float *gpre_acc, int B, int C, int P,
                             void *stream);
/* camli_gru_blend_bwd / the call above with gpre_acc += gpre (NULL: none): the pre-activation gradient of a half-step is also
 * the gradient of its hoisted context term, which is shared by every GRU iteration of a pass -- the running total is kept by
 * the adjoint kernels themselves instead of one tensor addition per iteration. */
int camli_gru_blend_bwd_acc(const float *g, const float *z, const float *h, const float *q, float *gpre_q, float *gz, float *gh,
                            float *gpre_acc, int B, int C, int P, int nan_to_num, void *stream);
/* the same with gz / grh read in place from channel slices of wider gradients (what the adjoint of cat([r*h, x]) hands
 * over): batch strides in floats, multiples of 4, >= C*P; pointers 16-byte aligned */
int camli_gru_gates_bwd_strided(const float *gz, int64_t gz_batch_stride, const float *grh, int64_t grh_batch_stride,
                                const float *z, const float *r, const float *h, float *gpre_zr, float *gh, int B, int C,
                                int P, void *stream);
/* nan_to_num = 1 (round 3): h_new = torch.nan_to_num(h_new), the last statement of the GRU (raft_core.py:138), folded
 * into the blend; its adjoint (zero gradient where the un-sanitised value was not finite) is folded into blend_bwd. */
int camli_gru_blend_fwd(const float *pre_q, const float *ctx_q, const float *z, const float *h, float *q, float *h_new,
                        int B, int C, int P, int nan_to_num, void *stream);
int camli_gru_blend_bwd(const float *g, const float *z, const float *h, const float *q, float *gpre_q, float *gz,
                        float *gh, int B, int C, int P, int nan_to_num, void *stream);

/*
 * Bias + activation epilogue of a convolution and its adjoint (the reference runs conv -> bias add ->
 * activation as separate passes: models/mlp.py:41-128, models/raft_core.py:155-197).
 *   x_inout [B,C,P] updated IN PLACE to act(x + bias[c]); act: 0 identity, 1 relu, 2 leaky_relu(0.1),
 *   3 sigmoid, 4 tanh.  bwd: gx = gy * act'(y) (fully written; may alias gy), gbias[c] += sum gx
 *   (float atomics; caller zero-fills gbias).
 *   sign_mask (optional, act 1 / 2 with P % 4 == 0, camli_bias_act_mask_bytes(B,C,P) bytes): the forward
 *   also records one bit per element (y > 0); a backward given the mask does not read y at all
 *   (8.1 instead of 12 bytes per element).  Pass NULL for the y-based form.
 *   act 0 (identity): gx equals gy, so pass gx = NULL (and y = NULL): the call then only reduces the bias
 *   sums, 4 bytes per element read and nothing written; the caller hands gy on as the input gradient.
 */
int64_t camli_bias_act_mask_bytes(int B, int C, int P);
int camli_bias_act_fwd(float *x_inout, const float *bias, void *sign_mask, int B, int C, int P, int act, void *stream);
int camli_bias_act_bwd(const float *gy, const float *y, const void *sign_mask, float *gx, float *gbias,
                       int B, int C, int P, int act, void *stream);
/* act(x + bias[c]) written to a channel slice of a wider tensor, out[b * out_batch_stride + c * P + p] (the concatenation the
 * next convolution reads: no cat pass); x [B,C,P] is only read; sign mask as camli_bias_act_fwd; the adjoint is
 * camli_bias_act_bwd_strided on the slice of the wider gradient */
int camli_bias_act_into_fwd(const float *x, const float *bias, void *sign_mask, float *out, int64_t out_batch_stride, int B,
                            int C, int P, int act, void *stream);
/* gy read in place from a channel slice of a wider gradient: batch stride in floats (>= C*P; with P % 4 == 0 a multiple of 4
 * and a 16-byte aligned pointer).  gx (dense [B,C,P]) and everything else as above. */
int camli_bias_act_bwd_strided(const float *gy, int64_t gy_batch_stride, const float *y, const void *sign_mask, float *gx,
                               float *gbias, int B, int C, int P, int act, void *stream);
/* y = act(x + bias[c] + res) in place on x (round 3): the closing relu(bn3(conv3(.)) + shortcut) of a residual block in
 * one pass; act 0 or 1, res [B,C,P] like x.  The adjoint is camli_bias_act_bwd (the gradient of res equals that of x). */
/* channels-last forms (round 3, the ResNet trunk): tensor [n_pix, C] with C fastest, C a power of two in [4, 1024], act 0 or
 * 1; res NULL or shaped like x; sign mask of camli_bias_act_nhwc_mask_bytes(n_pix, C) bytes (act 1).  bwd: act 1 writes
 * gx = gy * mask, act 0 only adds the per-channel sums of gy into gbias (gx ignored). */
int64_t camli_bias_act_nhwc_mask_bytes(long long n_pix, int C);
int camli_bias_act_nhwc_fwd(float *x_inout, const float *bias, const float *res, void *sign_mask, long long n_pix, int C,
                            int act, void *stream);
int64_t camli_bias_act_nhwc_bwd_workspace_bytes(long long n_pix, int C);
/* workspace: NULL (float atomics into gbias) or camli_bias_act_nhwc_bwd_workspace_bytes bytes (per-workgroup partial sums,
 * added in a fixed order by a second kernel: reproducible, and faster on large tensors) */
int camli_bias_act_nhwc_bwd(const float *gy, const void *sign_mask, float *gx, float *gbias, float *workspace, long long n_pix,
                            int C, int act, void *stream);
int camli_bias_act_res_fwd(float *x_inout, const float *bias, const float *res, void *sign_mask, int B, int C, int P,
                           int act, void *stream);

/*
 * Neighbour-weight network of a depth-wise set-conv: weight_net = MLP2d(3 -> 8 -> 32 -> C, ReLU after
 * every layer, bias, no norm) applied to the centred neighbour offsets (models/point_conv.py:102-121,
 * models/mlp.py:100-162; the reference runs gather, subtract, 3x (1x1 conv, bias, ReLU)).  One launch:
 * offsets and the two small layers on the VALU, the 32 -> C layer on the matrix cores
 * (v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate), output written once.
 *   xyz [B,3,M], centres [B,3,N] (channel-first), idx int64 rows of stride idx_stride (first k used),
 *   w1 [8,3] b1 [8] w2 [32,8] b2 [32] w3 [C,32] b3 [C], out [B,C,N,k] (fully written), C <= 128.
 * Every layer is bias-first followed by an input-ordered fmaf chain (bit-exact restatement:
 * oracle_weightnet_fwd).
 * bwd: gout [B,C,N,k] -> gradients of all six parameters, gw1 [8,3] gb1 [8] gw2 [32,8] gb2 [32] gw3 [C,32]
 *   gb3 [C], fully written (an empty problem writes zeros).  Hidden layers and ReLU masks are recomputed
 *   with the forward's arithmetic; every contraction runs on the matrix cores.  Workgroup partial sums go to
 *   `workspace` (camli_weightnet_bwd_workspace_bytes(C) bytes, contents undefined afterwards) and are
 *   added in a fixed order by a second kernel: no atomics, bit-reproducible.  The coordinates receive no
 *   gradient (they are constants on this path: models/camliraft_core.py:105-106).
 * k_major = 1: out / gout are [B,C,k,N] (the columns of the [C x N*k] product are enumerated with the point index
 *   fastest) -- the layout camli_pointconv_dw_fwd streams without an LDS transposition.  Same values.
 */
int camli_weightnet_fwd(const float *xyz, const float *centres, const int64_t *idx, int idx_stride,
                        const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                        const float *b3, float *out, int B, int C, int M, int N, int k, int k_major, void *stream);
int64_t camli_weightnet_bwd_workspace_bytes(int C);
int camli_weightnet_bwd(const float *xyz, const float *centres, const int64_t *idx, int idx_stride,
                        const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                        const float *b3, const float *gout, float *gw1, float *gb1, float *gw2, float *gb2,
                        float *gw3, float *gb3, float *workspace, int64_t workspace_bytes,
                        int B, int C, int M, int N, int k, int k_major, void *stream);

/*
 * Bilinear sampling of a feature map at pixel positions (2-D -> 3-D half of the CLFM fusion).
 * Replaces grid_sample_wrapper: models/utils.py:262-269 (normalise, F.grid_sample(bilinear,
 * align_corners=True, padding zeros)); its output is detached at the only call site on this path
 * (models/clfm.py:187-190), hence no adjoint.
 *   feat [B,C,H,W], uv [B,2,N] (x,y in pixels), out [B,C,N] (fully written); H, W >= 2.
 */
int camli_bilinear_sample_fwd(const float *feat, const float *uv, float *out, int B, int C, int H, int W, int N,
                              void *stream);

/*
 * Selective-kernel fusion of the image and point branches, full-size part (models/clfm.py:170-213:
 * squeezed = avg_pool(a + b); out = a * w[...,0] + b * w[...,1], w = softmax(gate(squeezed))).
 *   a, b, out, g, ga, gb [B,C,P];  s, gs [B,C];  w, gw [B,C,2];  B*C <= 65535 for the mix kernels.
 *   pool_fwd : s = mean_p(a + b)
 *   mix_fwd  : out = a * w0 + b * w1
 *   mix_bwd_w: gw0 = sum_p g * a, gw1 = sum_p g * b                      (fully written)
 *   mix_bwd_x: ga = g * w0 + gs / P, gb = g * w1 + gs / P  (gs may be NULL: no pooled term; fully written)
 */
int camli_sk_pool_fwd(const float *a, const float *b, float *s, int B, int C, int P, void *stream);
int camli_sk_mix_fwd(const float *a, const float *b, const float *w, float *out, int B, int C, int P, void *stream);
int camli_sk_mix_bwd_w(const float *g, const float *a, const float *b, float *gw, int B, int C, int P, void *stream);
int camli_sk_mix_bwd_x(const float *g, const float *w, const float *gs, float *ga, float *gb, int B, int C, int P,
                       void *stream);

/*
 * The SKFusion gate on [B,C] vectors (models/clfm.py:183-184,199-203): m = relu(s Wmid^T),
 * z = sigmoid(m Wout^T), w[b,c,:] = softmax(z[b,2c], z[b,2c+1]).  C <= 1024, R <= 512.
 *   s [B,C], wmid [R,C], wout [2C,R];  m [B,R], z [B,2C], w [B,C,2] (all fully written; m and z are kept
 *   for the backward).
 *   bwd: gw [B,C,2] -> gs [B,C] (fully written), gwmid [R,C] += , gwout [2C,R] += (plain read-modify-write in batch
 *   order, one workgroup per hidden unit -- no atomics since round 3; the caller zero-fills or passes its running
 *   accumulators, which must belong to the launch stream).
 */
int camli_sk_gate_fwd(const float *s, const float *wmid, const float *wout, float *m, float *z, float *w,
                      int B, int C, int R, void *stream);
int camli_sk_gate_bwd(const float *gw, const float *s, const float *m, const float *z, const float *w,
                      const float *wmid, const float *wout, float *gs, float *gwmid, float *gwout,
                      int B, int C, int R, void *stream);

/*
 * Flow read-out of the inverse-depth-scaling wrapper: out = paral2persp(pc1 + flow) - origin
 * (models/ids.py:36-67 applied per flow iterate at models/camliraft.py:108-110 / camlipwc.py / *_l.py).
 *   pc1, flow, origin, out [B,3,N]; f, cx, cy [B] (perspective intrinsics);
 *   ratio_w / ratio_h = (paral sensor - 1) / (persp sensor - 1), ratio_min = min of the two,
 *   half_w / half_h = (paral sensor - 1) / 2.
 *   bwd: gflow = d out / d flow applied to gout (pc1, origin constant); fully written.
 */
int camli_ids_flow_fwd(const float *pc1, const float *flow, const float *origin, const float *f, const float *cx,
                       const float *cy, float *out, float ratio_w, float ratio_h, float ratio_min,
                       float half_w, float half_h, int B, int N, void *stream);
int camli_ids_flow_bwd(const float *pc1, const float *flow, const float *gout, const float *f, const float *cx,
                       const float *cy, float *gflow, float ratio_w, float ratio_h, float ratio_min,
                       float half_w, float half_h, int B, int N, void *stream);

/*
 * Masked end-point-error sum of one flow iterate and its adjoint (sequence losses, order 'l2-norm':
 * models/losses.py:64-119 -- err = ||pred - target[:, :C]||_2 over the channels, averaged over the mask).
 *   pred [B,C,P] (C = 2 or 3), target [B,target_channels,P] with target_channels = C (no mask) or C + 1
 *   (mask = last channel > 0).
 *   fwd: *sum_out += sum over the masked (b,p) of err  (float atomics; caller zero-fills; the caller
 *        divides by the mask count and applies the iterate's weight).
 *   bwd: gpred = coef[0] * (pred - target) / err on the mask, 0 elsewhere and where err == 0
 *        (fully written); coef is a DEVICE scalar.
 */
int camli_masked_l2_fwd(const float *pred, const float *target, int target_channels, float *sum_out,
                        int B, int C, int P, void *stream);
int camli_masked_l2_bwd(const float *pred, const float *target, int target_channels, const float *coef,
                        float *gpred, int B, int C, int P, void *stream);

/*
 * out[b,c,p] = scale[b,c,p] * data[b,c,idx[b,p]]: nearest-point feature of every pixel times its score
 * (FusionAwareInterp with k = 1, models/clfm.py:70-76).  data [B,C,M], scale/out [B,C,P], idx int64 [B,P]
 * (values in [0,M)).  Linear in `scale` with the same kernel as adjoint (gscale = gout * data[idx]);
 * `data` is detached at the call site (clfm.py:186).
 */
int camli_gather_scale_fwd(const float *data, const float *scale, const int64_t *idx, float *out,
                           int B, int C, int M, int P, void *stream);

/*
 * Input side.
 * persp2paral: models/ids.py:4-33 for BOTH clouds in one launch.  pcs [B,6,N] (xyz of cloud 1, xyz of cloud 2),
 *   intrinsics [B,3] = (f, cx, cy) on the DEVICE, out1 / out2 [B,3,N]; ratio_w/h = (paral - 1)/(persp - 1),
 *   ratio_min = min of the two, half_w/h = (paral - 1)/2 (HOST scalars, as the reference folds them).
 *   Same expression order as the reference, unfused: equal to the torch composition bit for bit.
 * pad_normalize: models/camliraft.py:38-46 + models/utils.py:7-15.  images [B,6,H,W] (frame 1 RGB, frame 2 RGB)
 *   -> out1 / out2 [B,3,Hp,Wp] = (replicate_pad(frame) - mean[c]) / std[c]; pad_left columns on the left, the rest of
 *   Wp - W on the right, Hp - H rows at the bottom; mean3 / std3 are HOST arrays of 3 floats.
 */
int camli_persp2paral(const float *pcs, const float *intrinsics, float *out1, float *out2, int B, int N,
                      float ratio_w, float ratio_h, float ratio_min, float half_w, float half_h, void *stream);
int camli_pad_normalize(const float *images, float *out1, float *out2, int B, int H, int W, int Hp, int Wp,
                        int pad_left, const float *mean3, const float *std3, void *stream);
/*
 * project_pc2image (models/utils.py:234-259) + the feature-grid rescale of its callers (camliraft_core.py:51-56,
 * camlipwc_core.py:112-114): pc [B,3,N] -> uv [B,2,N].
 *   perspective = 0: u = (x + cx) * scale_x, v = (y + cy) * scale_y with HOST scalars cx, cy (the parallel camera);
 *   perspective = 1: u = (cx_b + (f_b / z) * x) * scale_x, ... with intrinsics [B,3] = (f, cx, cy) on the DEVICE
 *   (cx, cy arguments ignored).  Same expression order as the reference, unfused.  No autograd (clouds are inputs).
 */
int camli_project_pc2image(const float *pc, const float *intrinsics, float *uv, int B, int N, int perspective,
                           float cx, float cy, float scale_x, float scale_y, void *stream);

/*
 * Two-channel 3x3 convolution heads (FlowHead2D.conv2 of models/raft_core.py:169-181 and PWC's
 * conv_last): fp32 NCHW, stride 1, zero padding 1, Cout = 2.  x [B,Cin,H,W], w [2,Cin,3,3], bias [2] or NULL,
 * y [B,2,H,W] fully written.  bwd_data: gx [B,Cin,H,W] fully written.  bwd_weight: gw [2,Cin,3,3] and gb [2] (NULL to
 * skip) are WRITTEN (accumulate = 0) or ADDED TO (accumulate = 1: iteration-shared parameters sum over the GRU
 * iterations); workspace = camli_conv3x3_co2_bwd_weight_workspace_bytes(B, Cin, W) bytes of per-wave partial sums,
 * reduced in a fixed order (no atomics).
 */
int camli_conv3x3_co2_fwd(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int H, int W,
                          void *stream);

/*
 * Channels-last ("tap") convolution on the fp32 matrix cores (round 5, csrc/hip/convcl.h, wrwcl.h): GRU2D's 1x5 / 5x1
 * convolutions (models/raft_core.py:110-140: nn.Conv2d(hidden + input, hidden, (1, 5) | (5, 1), padding (0, 2) | (2, 0))) with both
 * adjoints, replacing the library's NHWC implicit-GEMM kernels and their layout transposes.  All tensors fp32 NHWC:
 *     y[p][n] = sum_t sum_c x[p + (dy[t], dx[t])][c] * wp[n][t][c]          p = (b, y, x); zero outside the image; stride 1
 * The input is cat[x0 (C0 channels, ldx0 floats per pixel), x1 (C1 channels; NULL / 0: none)], the output channels [0, N0) go to
 * y0 (ldy0 floats per pixel) and [N0, Cout) to y1 (N0 = Cout: one output).  wp = the weights packed [Cout][T][C0 + C1]
 * (from [Cout, Cin, kh, kw]: permute(0, 2, 3, 1), taps in row-major order, dy = ky - pad_h, dx = kx - pad_w).  The DATA GRADIENT
 * is the same call on gy with the negated taps and wp' = [Cin][T][Cout] (permute(1, 2, 3, 0)).  C0, C1 multiples of 16, Cout of
 * 128, every tensor below 2 GB.  dy / dx are HOST arrays of T <= 32 entries.
 */
int camli_convcl_fwd(const float *x0, int ldx0, int C0, const float *x1, int ldx1, int C1, const float *wp, float *y0, int ldy0,
                     int N0, float *y1, int ldy1, int B, int H, int W, int Cout, int T, const signed char *dy, const signed char *dx,
                     int accumulate0, int accumulate1, void *stream);
/*
 * One GRU2D half-step convolution with its gate arithmetic in the epilogue (models/raft_core.py:124-130, 132-138; the context
 * term of the input hoisted out of the iteration: cores/raft2d.GRU2D.prepare).  Dense NHWC tensors, hidden width 128:
 * h, z, r, rh, q, h_new [P][128], x [P][CX], ctx_zr [P][256], ctx_q [P][128]; wp_zr [256][T][128 + CX], wp_q [128][T][128 + CX].
 *   gates:  z | r = sigmoid(conv(cat[h, x]; wp_zr) + ctx_zr);  outputs z, rh = r * h, r (kept for the adjoint)
 *   blend:  q = tanh(conv(cat[rh, x]; wp_q) + ctx_q);  h_new = (1 - z) h + z q  (nan_to_num != 0: then torch.nan_to_num, :138)
 * The adjoints are camli_gru_blend_bwd / camli_gru_gates_bwd_into on the same NHWC tensors (read as [P][128][1]) followed by
 * camli_convcl_fwd on the negated taps and camli_convcl_wrw.
 */
int camli_convcl_gru_gates(const float *h, const float *x, int CX, const float *wp_zr, const float *ctx_zr, float *z, float *rh,
                           float *r, int B, int H, int W, int T, const signed char *dy, const signed char *dx, void *stream);
int camli_convcl_gru_blend(const float *rh, const float *x, int CX, const float *wp_q, const float *ctx_q, const float *z,
                           const float *h, float *h_new, float *q, int nan_to_num, int B, int H, int W, int T, const signed char *dy,
                           const signed char *dx, void *stream);
/*
 * The same half-step convolutions and their data gradient as 1-D Winograd F(4, 5) (round 6, csrc/hip/wino1d.h): 8 multiplications
 * per 4 outputs and channel pair where the 5-tap form spends 20; three launches each -- input transform (cat[x0, x1] -> V
 * [8][tiles][C], tile = 4 consecutive pixels along the kernel's axis), 8 plane contractions on the k-contiguous core of the tap
 * convolution (one launch), output transform with the SAME epilogue arithmetic.  axis 0: a 1 x 5 kernel with padding (0, 2),
 * axis 1: 5 x 1 with (2, 0).  U = camli_wino1d_weights(wp, ...): [8][N][C] from the packed weights wp [N][5][C] of the entry
 * points above (flip = 1 on the transposed packing [Cin][5][Cout]: the data gradient's weights).  workspace =
 * camli_wino1d_workspace_bytes(B, H, W, Cin, Cout, axis) bytes.  Rounding differs from the tap form's (factors up to 21/4 and 8
 * in the transforms): 1.3e-6 relative L2 at 256 channels.  camli_wino1d_conv = camli_convcl_fwd's plain form (two inputs, output
 * split at N0, = or +=); _gru_gates / _gru_blend = camli_convcl_gru_gates / _gru_blend.  v_keep (may be null): [8][tiles][128 + CX]
 * floats that receive the transformed input V instead of the workspace -- the weight gradient of the same convolution contracts
 * it again (camli_wino1d_wrw's v_in) instead of transforming the input a second time.
 */
int64_t camli_wino1d_workspace_bytes(int B, int H, int W, int Cin, int Cout, int axis);
int camli_wino1d_weights(const float *wp, float *U, int N, int C, int flip, void *stream);
int camli_wino1d_conv(const float *x0, int ldx0, int C0, const float *x1, int ldx1, int C1, const float *U, float *y0, int ldy0, int N0,
                      float *y1, int ldy1, float *workspace, int64_t workspace_bytes, int B, int H, int W, int Cout, int axis,
                      int accumulate0, int accumulate1, void *stream);
int camli_wino1d_gru_gates(const float *h, const float *x, int CX, const float *U_zr, const float *ctx_zr, float *z, float *rh, float *r,
                           float *v_keep, float *workspace, int64_t workspace_bytes, int B, int H, int W, int axis, void *stream);
int camli_wino1d_gru_blend(const float *rh, const float *x, int CX, const float *U_q, const float *ctx_q, const float *z, const float *h,
                           float *h_new, float *q, int nan_to_num, float *v_keep, float *workspace, int64_t workspace_bytes, int B, int H,
                           int W, int axis, void *stream);
/* camli_convcl_wrw's result for a 1 x 5 / 5 x 1 kernel, contracted in the transform domain (gw [Cout][C0 + C1][5] = | +=): input
 * transform of cat[x0, x1], A-transform of gy, 8 plane contractions over the tiles on the weight-gradient core of wrwcl.h (the
 * planes laid end to end, K splits that do not straddle planes), G^T over the planes.  C0 + C1 a multiple of 256, Cout of 128;
 * workspace = camli_wino1d_wrw_workspace_bytes (0 = unsupported shape).  Deterministic.  v_in (may be null): the transformed input a
 * forward launch kept (v_keep above; x0 / x1 are then not read) -- allowed when camli_wino1d_wrw_reuse says 1: the planes are
 * contracted as they lie, [8][tiles][Cin] without padding rows, so a K split that fills the device must divide them. */
int64_t camli_wino1d_wrw_workspace_bytes(int B, int H, int W, int Cin, int Cout, int axis);
int camli_wino1d_wrw_reuse(int B, int H, int W, int Cin, int Cout, int axis);
int camli_wino1d_wrw(const float *x0, int ldx0, int C0, const float *x1, int ldx1, int C1, const float *gy, int ldg, float *gw,
                     const float *v_in, float *workspace, int64_t workspace_bytes, int B, int H, int W, int Cout, int axis, int accumulate, void *stream);
/*
 * Weight gradient of the same convolution: gw [Cout][C0 + C1][T] (= the [Cout, Cin, kh, kw] weight tensor) = or += (accumulate)
 *     sum_p gy[p][n] * x[p + (dy[t], dx[t])][c]
 * gy NHWC [P][ldg].  Split over the pixels into about one workgroup per CU, parts in `workspace`
 * (camli_convcl_wrw_workspace_bytes), added in a fixed order: deterministic, no atomics.  C0 + C1 a multiple of 256 and Cout of 128;
 * or ONE input (C1 = 0) of a multiple of 128 channels and Cout a multiple of 256 (the kernel then contracts with the operands' roles
 * exchanged -- the flow / mask heads' 128 -> 512 convolution).
 */
int64_t camli_convcl_wrw_workspace_bytes(int B, int H, int W, int Cin, int Cout, int T);
int camli_convcl_wrw(const float *x0, int ldx0, int C0, const float *x1, int ldx1, int C1, const float *gy, int ldg, float *workspace,
                     int64_t workspace_bytes, float *gw, int accumulate, int B, int H, int W, int Cout, int T, const signed char *dy,
                     const signed char *dx, void *stream);
/*
 * Winograd F(M x M, 3x3) convolution, M = `tile` = 2 | 4, on the fp32 matrix cores (round 6, csrc/hip/winograd.h + gemm_w128.h):
 * (M + 2)^2 multiplications per M^2 outputs and channel pair instead of 9 M^2 (2.25 x | 4 x fewer); tile 4 pays ~10 x the
 * rounding error of tile 2 (factors up to 8 and 1 / 24 in the transforms; ~3e-6 relative L2 at 256 channels).  The 3x3 / stride 1 /
 * padding 1 convolutions of the RAFT update block -- MotionEncoder2D.conv_c2 (256 -> 192) and .conv (256 -> 126),
 * models/raft_core.py:148,151; FlowHead2D.conv1 (128 -> 256), :173; the mask head's first convolution (128 -> 256), :188 --
 * forward and data gradient, replacing the library's fp32 Winograd / implicit-GEMM kernels.  NCHW fp32 tensors:
 *     y[b][n][oy][ox] (= | +=) act( sum_c sum_{i,j} x[b][c][oy+i-1][ox+j-1] * w[n][c][i][j] + bias[n] )      zero outside the image
 * as three launches: input transform (x -> V [P][C][tiles], P = (tile + 2)^2), P plane GEMMs U[t]^T V[t] (one launch), output
 * transform.  Every call of one convolution takes the same `tile`.
 *   camli_wino_weights: U [P][Kp][Mp] = G g G^T of w [Cout][Cin][3][3]; flip = 0: K = Cin, M = Cout (forward); flip = 1: K = Cout,
 *     M = Cin, taps reversed -- the weights of the DATA GRADIENT, which is the same convolution of the output gradient.
 *     Kp = K rounded up to a multiple of 16, Mp = M to a multiple of 4 (zeros beyond); camli_wino_weight_floats(K, M, tile) = P Kp Mp.
 *   camli_wino_conv3x3: image b of x at x + b * x_bs (C dense H x W planes: a channel slice of a wider NCHW tensor is fine), of y
 *     at y + b * y_bs (N planes); bias optional; act 0 none | 1 ReLU | 2 ReLU then nan_to_num (raft_core.py:163-164);
 *     accumulate: y += (before act).  y_bits (optional, act != 0): the ACTIVATION BITS of y are written -- [B][N][H][ceil(W / 8)]
 *     bytes (camli_wino_mask_bytes), bit j of byte s = output pixel 8 s + j passes the gradient (pre-activation > 0; act 2: and
 *     finite); x_bits (optional): such bits for the INPUT, x reads as zero where its bit is clear -- the ReLU adjoint of the
 *     data gradient rides on the input transform at 1/32 of the bytes of re-reading the forward's output.  C > 32, N >= 4;
 *     workspace = camli_wino_workspace_bytes(B, C, N, H, W, tile) bytes (V and the transform-domain output: P / tile^2 x the
 *     input + the output), 16-byte aligned.  Differs from the direct fp32 form by its rounding only (tests/test_winograd_gpu.py).
 *   camli_wino_wrw: WEIGHT GRADIENT in the transform domain: gU[t][c][n] = sum_tiles V[t][c][tile] * (A gy A^T)[t][n][tile], then
 *     gw [N][C][3][3] (= | +=) G^T gU G -- the same 2.25 x fewer multiplications as the forward.  x [B][C][H][W] (image stride
 *     x_bs), gy [B][N][H][W] (image stride gy_bs); gy_bits optional: the forward's activation bits, gy reads as zero where its
 *     bit is clear.  gbias [N] (optional) (= | +=, gbias_accumulate) the bias gradient: the per-channel sum of the masked gy,
 *     taken from the transform-domain plane that holds the tile sums.  workspace = camli_wino_wrw_workspace_bytes(B, C, N, H, W,
 *     tile) bytes (0 = unsupported shape).  The contraction over the tiles is split over the CUs, the parts are summed in a fixed
 *     order: deterministic, no atomics.
 */
int64_t camli_wino_weight_floats(int K, int M, int tile);
int camli_wino_weights(const float *w, float *U, int Cout, int Cin, int flip, int tile, void *stream);
int64_t camli_wino_workspace_bytes(int B, int C, int N, int H, int W, int tile);
int64_t camli_wino_mask_bytes(int B, int C, int H, int W);
int camli_wino_conv3x3(const float *x, int64_t x_bs, const unsigned char *x_bits, const float *U, const float *bias, float *y,
                       int64_t y_bs, unsigned char *y_bits, float *workspace, int64_t workspace_bytes, int B, int C, int N, int H,
                       int W, int act, int accumulate, int tile, void *stream);
int64_t camli_wino_wrw_workspace_bytes(int B, int C, int N, int H, int W, int tile);
int camli_wino_wrw(const float *x, int64_t x_bs, const float *gy, int64_t gy_bs, const unsigned char *gy_bits, float *gw,
                   float *gbias, float *workspace, int64_t workspace_bytes, int B, int C, int N, int H, int W, int accumulate,
                   int gbias_accumulate, int tile, void *stream);
int camli_conv3x3_co2_bwd_data(const float *gy, const float *w, float *gx, int B, int Cin, int H, int W, void *stream);
long long camli_conv3x3_co2_bwd_weight_workspace_bytes(int B, int Cin, int W);
int camli_conv3x3_co2_bwd_weight(const float *gy, const float *x, float *workspace, float *gw, float *gb, int accumulate,
                                 int B, int Cin, int H, int W, void *stream);

/*
 * 3x3 / stride 2 / padding 1 max pooling (the ResNet stem's nn.MaxPool2d(3, 2, 1)) over `planes` = B*C planes of H x W,
 * Ho = (H - 1) / 2 + 1, Wo likewise.  fwd: y [planes,Ho,Wo] and arg uint8 [planes,Ho,Wo] = window-local position 0..8 of the
 * first maximum in row-major order (a NaN wins, as in torch).  bwd: gx [planes,H,W] fully written = sum of gy over the (at
 * most four) windows whose arg points at the element: a gather, no atomics.
 */
int camli_maxpool3x3s2_fwd(const float *x, float *y, unsigned char *arg, int planes, int H, int W, int Ho, int Wo,
                           void *stream);
int camli_maxpool3x3s2_bwd(const float *gy, const unsigned char *arg, float *gx, int planes, int H, int W, int Ho, int Wo,
                           void *stream);

/*
 * Batched 2-D transpose between channel-first and channels-last maps (internal glue of the cores: the update block's wide
 * convolutions run on channels-last operands, cores/blocks.py; the reference keeps everything NCHW and lets the library
 * transpose around every call, raft_core.py:112-139).
 *   dst[b*dst_batch_stride + j*dst_row_stride + i] = src[b*src_batch_stride + i*src_row_stride + j],  i < rows, j < cols
 *   (strides in floats; a channel slice of a wider channels-last map is a row stride = the wide channel count).
 *   16-byte accesses when rows, cols, all strides are multiples of 4 and both pointers 16-byte aligned, scalar otherwise.
 */
int camli_transpose_planes(const float *src, int64_t src_batch_stride, int64_t src_row_stride, float *dst,
                           int64_t dst_batch_stride, int64_t dst_row_stride, int B, int rows, int cols, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CAMLI_HIP_H */

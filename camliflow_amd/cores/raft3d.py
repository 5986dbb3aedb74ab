"""Point branch of CamLiRAFT (host-side counterpart of the reference's models/camliraft_l_core.py).

Pieces, in data-flow order:
  Encoder3D        point-feature pyramid: shared MLPs + down-sampling PointConv
  Correlation3D    dense point cost volume, pooled over the target pyramid, looked up through KNN
  MotionEncoder3D  (correlation, flow) -> motion features           [4 depth-wise set-convs]
  GRU3D            gated recurrent update of the hidden state       [3 depth-wise set-convs]
  FlowHead3D       hidden state -> residual scene flow              [2 depth-wise set-convs + 1x1]
  CamLiRAFT_L_Core the point-only model built from them
Module / parameter names are a checkpoint-compatibility surface and match the reference.
"""
import os

import torch
import torch.nn as nn

from ..csrc import wrapper as _ops
from . import runtime
from .blocks import Conv1dNormRelu, MLP1d, MLP2d
from .geometry import (backwarp_3d, backwarp_3d_levels, batch_indexing, build_pc_pyramid, knn_channel_first,
                       knn_interpolation)
from .setconv import PointConv, PointConvDW, pass_cache

# number of points kept at each pyramid level; every CamLi* model hard-codes it
# (camliraft_l_core.py:174-176) and the iterations run on level 2 (2048 points)
PYRAMID_SIZES = [4096, 2048, 1024, 512, 256]
HIDDEN = 128          # hidden-state / context width of the recurrent update
SELF_KNN = 32         # width of the precomputed self-neighbour table; set-convs slice what they need


class Encoder3D(nn.Module):
    """level 0: MLP(3 -> c0 -> c0); level i+1: MLP(c_i -> c_i -> c_{i+1}) then PointConv onto the
    next (sparser) pyramid level (camliraft_l_core.py:8-37)."""

    def __init__(self, n_channels, norm=None, k=16):
        super().__init__()
        first = n_channels[0]
        self.level0_mlp = MLP1d(3, [first, first])
        stages = list(zip(n_channels[:-1], n_channels[1:]))
        self.mlps = nn.ModuleList(MLP1d(a, [a, b]) for a, b in stages)
        self.convs = nn.ModuleList(PointConv(b, b, norm=norm, k=k) for _, b in stages)

    def forward(self, xyzs):
        assert len(xyzs) == len(self.mlps) + 1
        pyramid = [self.level0_mlp(xyzs[0])]
        for level in range(len(self.mlps)):
            widened = self.mlps[level](pyramid[level])
            pyramid.append(self.convs[level](xyzs[level], widened, xyzs[level + 1]))
        return pyramid


class Correlation3D(nn.Module):
    """RAFT-style point cost volume (camliraft_l_core.py:40-101).

    build : V0 = f1^T f2 / C  [B,N,M0];  V_l = mean over the 3 nearest level-(l-1) targets
    lookup: per level take the k nearest (warped) targets of every source point, feed
            (dxyz, V_l entry) through ``cost_mlp`` and sum over the neighbours; the four levels are
            concatenated and merged by a 1x1 conv.
    The pyramid is module state between ``build_cost_volume_pyramid`` and the lookups, like the
    reference's (not re-entrant)."""

    def __init__(self, out_channels, k=16):
        super().__init__()
        self.k = k
        quarter = out_channels // 4
        self.cost_mlp = MLP2d(4, [quarter, quarter], act='relu')
        self.merge = Conv1dNormRelu(out_channels, out_channels)
        self.cost_volume_pyramid = None

    def build_cost_volume_pyramid(self, feat1, feat2, xyzs2, k=3, nested=False):
        """``nested``: the caller guarantees that xyzs2[l+1] is the first n_{l+1} points of xyzs2[l] (the FPS pyramid,
        models/utils.py:121-125); the lookups of the pass then search and gather all levels in one launch each."""
        parents = [_ops.k_nearest_neighbor(fine, coarse, k=k)                  # [B,M_l,k] into level l-1
                   for coarse, fine in zip(xyzs2[1:], xyzs2[:-1])]
        if runtime.fused() and feat1.is_cuda and len(parents) < 8 and os.environ.get('CAMLI_CORR3D_BUILD', 'hip') == 'hip':
            from ..csrc import fused
            levels = fused.point_volume_pyramid(feat1, feat2, parents)
        else:
            levels = [torch.bmm(feat1.float().transpose(1, 2), feat2.float()) / feat1.shape[1]]
            for idx in parents:
                levels.append(batch_indexing(levels[-1], idx).mean(dim=-1))
        dense = levels[0]
        self.cost_volume_pyramid = levels
        self._nested = None
        self._prior_crosses = None        # a new pass: other clouds
        if nested and runtime.fused() and dense.is_cuda and len(levels) <= 4 and min(lvl.shape[2] for lvl in levels) >= self.k:
            from ..csrc import fused
            self._nested = fused.Corr3DPyramid(levels)

    def release(self):
        """Let go of the pass's volumes once the last lookup is enqueued (see raft2d.Correlation2D.release)."""
        self.cost_volume_pyramid = None
        self._nested = None
        self._prior_crosses = None

    def calc_matching_cost(self, xyz1, xyz2, cost_volume):
        bs, n_src, n_dst = cost_volume.shape
        cross = _ops.k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)       # [B,N,k]
        plain = not (xyz1.requires_grad or xyz2.requires_grad)
        if runtime.fused() and plain:
            from ..csrc import fused
            lookup = fused.corr3d_lookup_input(cost_volume, xyz1, xyz2, cross)            # [B,4,N,k]
        else:
            offset = batch_indexing(xyz2, cross) - xyz1.view(bs, 3, n_src, 1)
            entry = batch_indexing(cost_volume.reshape(bs * n_src, n_dst), cross.reshape(bs * n_src, self.k),
                                   layout='channel_last').reshape(bs, 1, n_src, self.k)
            lookup = torch.cat([offset, entry], dim=1)
        return self.cost_mlp(lookup).sum(dim=-1)

    def forward(self, xyz1, xyzs2):
        if runtime.fused() and xyz1.is_cuda:
            if not (xyz1.requires_grad or any(x.requires_grad for x in xyzs2[:4])):
                return self._forward_batched(xyz1, xyzs2)
            runtime.fallback('Correlation3D(RAFT)', 'differentiable coordinates')
        per_level = [self.calc_matching_cost(xyz1, xyzs2[lvl], self.cost_volume_pyramid[lvl]) for lvl in range(4)]
        return self.merge(torch.cat(per_level, dim=1))

    def _forward_batched(self, xyz1, xyzs2):
        """Same math with the four levels' (dxyz, cost) columns laid side by side, so ``cost_mlp`` (a
        per-column MLP) runs once on [B,4,N,4k] instead of four times on [B,4,N,k]: a quarter of the
        launches in both directions, one larger GEMM."""
        from ..csrc import fused
        bs, n_src = xyz1.shape[0], xyz1.shape[2]
        pyr = getattr(self, '_nested', None)
        if pyr is not None and len(xyzs2) >= len(pyr.levels) and xyzs2[0].shape[2] == pyr.sizes[0]:
            # nested target levels: one search over the level-0 cloud yields every prefix's neighbours, one gather
            # writes the concatenated columns, and the adjoint accumulates into per-pass gradient volumes
            from .geometry import _channel_last
            level0 = xyzs2[0]
            # round 5: the previous iteration's neighbours bound this iteration's k-th distances (the target cloud is the same
            # one back-warped a little further): same indices, the scan queues a fraction of the candidates
            prior = getattr(self, '_prior_crosses', None) if os.environ.get('CAMLI_KNN_PRIOR', '1') != '0' else None
            queries = _channel_last(xyz1, True)
            if prior is not None and (len(prior) != len(pyr.sizes) or prior[0].shape[:2] != queries.shape[:2]):
                prior = None
            crosses = _ops.k_nearest_neighbor_prefixes(level0.detach().transpose(1, 2).contiguous(), queries, pyr.sizes, self.k,
                                                       prior=prior)
            self._prior_crosses = crosses
            lookup = fused.corr3d_lookup_levels(pyr, xyz1, level0, crosses)
        else:
            columns = []
            for lvl in range(4):
                cross = knn_channel_first(xyzs2[lvl], xyz1, self.k, invariant_query=True)
                columns.append(fused.corr3d_lookup_input(self.cost_volume_pyramid[lvl], xyz1, xyzs2[lvl], cross))
            lookup = torch.cat(columns, dim=3)
        convs = [layer.conv_fn for layer in self.cost_mlp.convs]
        if os.environ.get('CAMLI_CORR3D_MLP', 'fused') == 'fused':
            if (all(layer._epilogue == 'relu' for layer in self.cost_mlp.convs)
                    and fused.corr3d_cost_mlp_supported(lookup, convs, 4)):
                # both layers, the ReLUs and the sum over the neighbours in one kernel each way (fp32 also under autocast,
                # like every point op): the two [B,C/4,N,4k] activations stay in registers
                return self.merge(fused.corr3d_cost_mlp(lookup, convs[0], convs[1], 4))
            runtime.fallback('Correlation3D(RAFT).cost_mlp', 'outside the fused cost MLP (4 levels x 16 neighbours, width 32, N % 8 == 0)')
        cost = self.cost_mlp(lookup)                                                  # [B,C/4,N,4k]
        cost = cost.view(bs, -1, n_src, 4, self.k).sum(dim=-1)                        # [B,C/4,N,4]
        cost = cost.permute(0, 3, 1, 2).reshape(bs, -1, n_src)                        # level-major channels
        return self.merge(cost)


class FlowHead3D(nn.Module):
    def __init__(self, input_dim=128):
        super().__init__()
        self.conv1 = PointConvDW(input_dim, 128, k=32)
        self.conv2 = PointConvDW(128, 64, k=32)
        self.fc = nn.Conv1d(64, 3, kernel_size=1)

    def forward(self, xyz, features, knn_indices=None):
        x = features.float()
        for conv in (self.conv1, self.conv2):
            x = conv(xyz, x, knn_indices=knn_indices)
        return self.fc(x)


class GRU3D(nn.Module):
    """h' = (1-z) h + z q with z, r, q three k=4 depth-wise set-convs over [h | x]
    (camliraft_l_core.py:119-134)."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        width = hidden_dim + input_dim
        self.conv_z = PointConvDW(width, hidden_dim, act=None, k=4)
        self.conv_r = PointConvDW(width, hidden_dim, act=None, k=4)
        self.conv_q = PointConvDW(width, hidden_dim, act=None, k=4)
        self._zeros = {}

    def _zero_context(self, like):
        key = (tuple(like.shape), like.device)
        z = self._zeros.get(key)
        if z is None:
            z = self._zeros[key] = torch.zeros_like(like)
        return z

    def forward(self, xyz, h, x, knn_indices=None):
        h, x = h.float(), x.float()
        joint = torch.cat([h, x], dim=1)
        if runtime.fused() and h.is_cuda and (h.shape[1] * h.shape[2]) % 4 != 0:
            runtime.fallback('GRU3D', 'plane of %d elements is not a multiple of 4' % (h.shape[1] * h.shape[2]))
        if runtime.fused() and h.is_cuda and (h.shape[1] * h.shape[2]) % 4 == 0:
            # the elementwise halves through the GRU kernels of the image branch (camli_gru_gates / _blend) with a
            # zero context term: 3 launches instead of 8 forward, 2 instead of ~14 backward
            from ..csrc import fused
            pre_zr = torch.cat([self.conv_z(xyz, joint, knn_indices=knn_indices),
                                self.conv_r(xyz, joint, knn_indices=knn_indices)], dim=1)
            update, reset_h = fused.gru_gates(pre_zr, self._zero_context(pre_zr), h.contiguous())
            pre_q = self.conv_q(xyz, torch.cat([reset_h, x], dim=1), knn_indices=knn_indices)
            return fused.gru_blend(pre_q, self._zero_context(pre_q), update, h.contiguous())
        update = torch.sigmoid(self.conv_z(xyz, joint, knn_indices=knn_indices))
        reset = torch.sigmoid(self.conv_r(xyz, joint, knn_indices=knn_indices))
        candidate = torch.tanh(self.conv_q(xyz, torch.cat([reset * h, x], dim=1), knn_indices=knn_indices))
        return (1 - update) * h + update * candidate


class MotionEncoder3D(nn.Module):
    def __init__(self, corr_dim=128):
        super().__init__()
        self.conv_c1 = PointConvDW(corr_dim, corr_dim)
        self.conv_f1 = PointConvDW(3, 32, k=32)
        self.conv_f2 = PointConvDW(32, 16, k=16)
        self.conv = PointConvDW(corr_dim + 16, 128 - 3, k=16)

    def forward(self, xyz, flow, corr, knn_indices):
        corr, flow = corr.float(), flow.float()
        from_corr = self.conv_c1(xyz, corr, knn_indices=knn_indices)
        from_flow = self.conv_f1(xyz, flow, knn_indices=knn_indices)
        from_flow = self.conv_f2(xyz, from_flow, knn_indices=knn_indices)
        mixed = self.conv(xyz, torch.cat([from_corr, from_flow], dim=1), knn_indices=knn_indices)
        return torch.cat([mixed, flow], dim=1)


class CamLiRAFT_L_Core(nn.Module):
    """Point-only model (camliraft_l_core.py:158-225): FPS pyramid -> encoders on levels 0..2 ->
    RAFT iterations on the 2048-point level -> every iterate interpolated back to the input cloud."""

    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        widths = [64, 96, 128]
        self.fnet = Encoder3D(n_channels=widths, norm='batch_norm', k=16)
        self.cnet = Encoder3D(n_channels=widths, norm='batch_norm', k=16)
        self.cnet_aligner = nn.Conv1d(128, 2 * HIDDEN, kernel_size=1)
        self.correlation = Correlation3D(out_channels=128, k=16)
        self.motion_encoder = MotionEncoder3D(corr_dim=128)
        self.gru = GRU3D(input_dim=128 + 128, hidden_dim=HIDDEN)
        self.flow_head = FlowHead3D(input_dim=HIDDEN)

    def forward(self, pc1, pc2):
        with pass_cache():
            return self._run(pc1, pc2)

    def _run(self, pc1, pc2):
        pyramid1, pyramid2, _, _ = build_pc_pyramid(pc1, pc2, PYRAMID_SIZES)
        top = 2                                                     # features come from pyramid level 2
        feat1 = self.fnet(pyramid1[:top + 1])[top]
        feat2 = self.fnet(pyramid2[:top + 1])[top]
        context = self.cnet_aligner(self.cnet(pyramid1[:top + 1])[top])

        work1, work2 = pyramid1[top:], pyramid2[top:]               # [2048, 1024, 512, 256]
        xyz1 = work1[0]
        self.correlation.build_cost_volume_pyramid(feat1, feat2, work2, nested=True)   # build_pc_pyramid: FPS prefixes
        hidden, ctx = torch.split(context, [HIDDEN, HIDDEN], dim=1)
        hidden, ctx = torch.tanh(hidden), torch.relu(ctx)
        neighbours = _ops.k_nearest_neighbor(xyz1, xyz1, k=SELF_KNN)
        n_iters = self.cfgs.n_iters_train if self.training else self.cfgs.n_iters_eval

        flow = torch.zeros_like(xyz1)
        targets = work2
        iterates = []
        for step in range(n_iters):
            if step:
                flow = flow.detach()
                targets = backwarp_3d_levels(xyz1, work2, flow, nested=True)
            corr = self.correlation(xyz1, targets)
            motion = self.motion_encoder(xyz1, flow, corr, knn_indices=neighbours)
            hidden = self.gru(xyz1, h=hidden, x=torch.cat([ctx, motion], dim=1), knn_indices=neighbours)
            flow = flow + self.flow_head(xyz1, hidden, neighbours).float()
            iterates.append(flow)
        self.correlation.release()
        return [knn_interpolation(xyz1, f, pc1, k=3, invariant_input=True, invariant_query=True) for f in iterates]

"""3-D PWC branch (counterpart of models/camlipwc_l_core.py): PointConv feature pyramid, the
PointPWC learnable cost volume, PointConv flow estimator and coarse-to-fine decoding.
"""
import torch
import torch.nn as nn

from ..csrc import wrapper as _ops
from . import runtime
from .blocks import Conv1dNormRelu, MLP1d, MLP2d
from .geometry import backwarp_3d, batch_indexing, knn_channel_first, knn_interpolation
from .pwc2d import pyramid_aligners
from .setconv import PointConv

PYRAMID_CHANNELS_3D = [16, 32, 64, 96, 128, 192]


class FeaturePyramid3D(nn.Module):
    def __init__(self, n_channels, norm=None, k=16):
        super().__init__()
        self.level0_mlp = MLP1d(3, [n_channels[0], n_channels[0]])
        self.pyramid_mlps = nn.ModuleList()
        self.pyramid_convs = nn.ModuleList()
        for c_in, c_out in zip(n_channels[:-1], n_channels[1:]):
            self.pyramid_mlps.append(MLP1d(c_in, [c_in, c_out]))
            self.pyramid_convs.append(PointConv(c_out, c_out, norm=norm, k=k))

    def forward(self, xyzs):
        assert len(xyzs) == len(self.pyramid_mlps) + 1
        feats = [self.level0_mlp(xyzs[0])]
        for i, (mlp, conv) in enumerate(zip(self.pyramid_mlps, self.pyramid_convs)):
            feats.append(conv(xyzs[i], mlp(feats[-1]), xyzs[i + 1]))
        return feats


class Correlation3D(nn.Module):
    """PointPWC learnable cost volume (camlipwc_l_core.py:39-106), three stages per source point p:

    point-to-point   c(p, q)  = MLP2d([f1(p) | f2(q) | q - p])            q in KNN_k(p; cloud 2)
    point-to-patch   C(p)     = sum_q  weight_net2(q - p) * c(p, q)
    patch-to-patch   out(p)   = sum_p' weight_net1(p' - p) * C(p')         p' in KNN_k(p; cloud 1)
    """

    def __init__(self, in_channels, out_channels, align_channels=None, k=16):
        super().__init__()
        self.k = k
        self.cost_mlp = MLP2d(3 + 2 * in_channels, [out_channels, out_channels], act='leaky_relu')
        self.weight_net1 = MLP2d(3, [8, 8, out_channels], act='relu')
        self.weight_net2 = MLP2d(3, [8, 8, out_channels], act='relu')
        self.feat_aligner = nn.Identity() if align_channels is None else Conv1dNormRelu(out_channels, align_channels)

    def _self_neighbours(self, xyz1, given):
        if given is None:
            return _ops.k_nearest_neighbor(input_xyz=xyz1, query_xyz=xyz1, k=self.k)
        assert given.shape[:2] == torch.Size([xyz1.shape[0], xyz1.shape[2]]) and given.shape[2] >= self.k
        return given[:, :, :self.k]

    def forward(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None):
        if runtime.fused() and xyz1.is_cuda:
            if (self.k & (self.k - 1)) == 0 and self.k <= 64 and not xyz1.requires_grad and len(self.cost_mlp.convs) == 2:
                return self._forward_fused(xyz1, feat1, xyz2, feat2, knn_indices_1in1)
            runtime.fallback('Correlation3D(PWC)', 'k = %d is not a power of two <= 64, or differentiable xyz1' % self.k)
        bs, channels, n = feat1.shape
        origin = xyz1.view(bs, 3, n, 1)

        # cloud-2 neighbourhood of every source point
        cross = _ops.k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        d_cross = batch_indexing(xyz2, cross) - origin                                  # [B,3,N,k]
        pair = torch.cat([feat1[:, :, :, None].expand(bs, channels, n, self.k), batch_indexing(feat2, cross), d_cross],
                         dim=1)
        to_patch = (self.weight_net2(d_cross) * self.cost_mlp(pair)).sum(dim=3)         # [B,Cout,N]

        # cloud-1 neighbourhood: aggregate the point-to-patch costs of the neighbours
        own = self._self_neighbours(xyz1, knn_indices_1in1)
        d_own = batch_indexing(xyz1, own) - origin
        patch = (self.weight_net1(d_own) * batch_indexing(to_patch, own)).sum(dim=3)
        return self.feat_aligner(patch)


    def _forward_fused(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1):
        """Same math without the [B, 2C+3, N, k] tensor (csrc/hip/pwc3d.hip): the first cost-MLP layer is linear in
        the concatenated channels, so its f1 / f2 blocks become per-POINT 1x1 convolutions and only the offset block
        is evaluated per pair; gather + add + leaky run as one kernel, the two weighted neighbour sums as one kernel
        each.  xyz2 may carry a gradient (CamLiPWC warps it with a live flow, camlipwc_core.py:179): the offsets and
        weight_net2 stay differentiable; weight_net1 sees constant coordinates and runs on the matrix cores."""
        from ..csrc import fused
        from torch.nn.functional import conv1d, conv2d
        bs, channels, n = feat1.shape
        k = self.k
        cross = knn_channel_first(xyz2.detach(), xyz1, k, invariant_query=True).contiguous()
        d_cross = batch_indexing(xyz2, cross) - xyz1.view(bs, 3, n, 1)                   # [B,3,N,k], grad -> xyz2
        first = self.cost_mlp.convs[0].conv_fn
        w = first.weight.reshape(first.weight.shape[0], -1).float()
        w_f1, w_f2, w_d = w[:, :channels], w[:, channels:2 * channels], w[:, 2 * channels:]
        a = conv1d(feat1.float(), w_f1[:, :, None])                                         # [B,C,N]
        bm = conv1d(feat2.float(), w_f2[:, :, None])                                        # [B,C,M]
        e = conv2d(d_cross, w_d[:, :, None, None], first.bias)                              # [B,C,N,k]
        h1 = fused.pwc3d_pair(a, bm, e, cross, slope=0.1)
        h2 = self.cost_mlp.convs[1](h1)
        to_patch = fused.ksum(self.weight_net2(d_cross), h2)                                # [B,C,N]

        own = self._self_neighbours(xyz1, knn_indices_1in1).contiguous()
        if fused.weightnet_hidden8_supported(self.weight_net1):
            w1 = fused.weightnet_hidden8(xyz1, xyz1, own, k, self.weight_net1)
        else:
            runtime.fallback('weight_net1', 'not MLP2d(3,[8,8,C<=256],relu)')
            w1 = self.weight_net1(batch_indexing(xyz1, own) - xyz1.view(bs, 3, n, 1))
        return self.feat_aligner(fused.gather_wsum(w1, to_patch, own))


class FlowEstimator3D(nn.Module):
    def __init__(self, n_channels, norm=None, conv_last=True, k=16):
        super().__init__()
        self.point_conv1 = PointConv(in_channels=n_channels[0], out_channels=n_channels[1], norm=norm, k=k)
        self.point_conv2 = PointConv(in_channels=n_channels[1], out_channels=n_channels[2], norm=norm, k=k)
        self.mlp = MLP1d(n_channels[2], [n_channels[2], n_channels[3]])
        self.flow_feat_dim = n_channels[3]
        self.conv_last = nn.Conv1d(n_channels[3], 3, kernel_size=1) if conv_last else None

    def forward(self, xyz, feat, knn_indices):
        feat = self.point_conv1(xyz, feat, knn_indices=knn_indices)
        feat = self.point_conv2(xyz, feat, knn_indices=knn_indices)
        feat = self.mlp(feat)
        if self.conv_last is None:
            return feat
        return feat, self.conv_last(feat)


class CamLiPWC_L_Core(nn.Module):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.feature_pyramid = FeaturePyramid3D(n_channels=PYRAMID_CHANNELS_3D, norm=cfgs.norm.feature_pyramid)
        self.correlations = nn.ModuleList([nn.Identity()] + [Correlation3D(c, c, 64) for c in PYRAMID_CHANNELS_3D[1:]])
        self.pyramid_feat_aligners = pyramid_aligners(Conv1dNormRelu)
        self.flow_estimator = FlowEstimator3D(n_channels=[64 + 64 + 3, 128, 128, 64], norm=cfgs.norm.flow_estimator)

    def encode(self, xyzs):
        return self.feature_pyramid(xyzs)

    def decode(self, xyzs1, xyzs2, feats1_3d, feats2_3d):
        """coarse-to-fine: at each level warp cloud 2 with the up-sampled flow of the coarser level,
        build the cost volume, estimate the residual (camlipwc_l_core.py:172-208)."""
        coarsest = len(xyzs1) - 1
        coarse_to_fine = []
        for level in range(coarsest, 0, -1):
            xyz1, xyz2 = xyzs1[level], xyzs2[level]
            own = _ops.k_nearest_neighbor(xyz1, xyz1, k=16)
            if level == coarsest:
                prior = torch.zeros([xyz1.shape[0], 3, xyz1.shape[2]], device=xyz1.device)
                warped = xyz2
            else:
                prior = knn_interpolation(xyzs1[level + 1], coarse_to_fine[-1], xyz1)
                warped = backwarp_3d(xyz1, xyz2, prior)
            cost = self.correlations[level](xyz1, feats1_3d[level], warped, feats2_3d[level], own)
            x = torch.cat([self.pyramid_feat_aligners[level](feats1_3d[level]), cost, prior], dim=1)
            coarse_to_fine.append(prior + self.flow_estimator(xyz1, x, own)[1])
        fine_to_coarse = [f.float() for f in reversed(coarse_to_fine)]
        return [knn_interpolation(xyzs1[i + 1], flow, xyzs1[i]) for i, flow in enumerate(fine_to_coarse)]

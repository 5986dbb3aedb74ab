"""3-D PWC branch (counterpart of models/camlipwc_l_core.py): PointConv feature pyramid, the
PointPWC learnable cost volume, PointConv flow estimator and coarse-to-fine decoding.
"""
import torch
import torch.nn as nn

from ..csrc import wrapper as _ops
from .blocks import Conv1dNormRelu, MLP1d, MLP2d
from .geometry import backwarp_3d, batch_indexing, knn_interpolation
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
    """PointPWC cost volume (camlipwc_l_core.py:39-106).

    point-to-point:    MLP2d([f1 | f2_knn | dxyz]) over the k nearest points of cloud 2
    point-to-patch:    sum_k weight_net2(dxyz) * p2p
    patch-to-patch:    sum_k weight_net1(dxyz_self) * gather(p2patch, self-KNN)
    """

    def __init__(self, in_channels, out_channels, align_channels=None, k=16):
        super().__init__()
        self.k = k
        self.cost_mlp = MLP2d(3 + 2 * in_channels, [out_channels, out_channels], act='leaky_relu')
        self.weight_net1 = MLP2d(3, [8, 8, out_channels], act='relu')
        self.weight_net2 = MLP2d(3, [8, 8, out_channels], act='relu')
        self.feat_aligner = Conv1dNormRelu(out_channels, align_channels) if align_channels is not None else nn.Identity()

    def forward(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None):
        batch_size, in_channels, n_points = feat1.shape
        centre = xyz1.view(batch_size, 3, n_points, 1)

        knn_1in2 = _ops.k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        offset2 = batch_indexing(xyz2, knn_1in2) - centre                               # [B,3,N,k]
        feat2_knn = batch_indexing(feat2, knn_1in2)                                     # [B,C,N,k]
        feat1_rep = feat1[:, :, :, None].expand(batch_size, in_channels, n_points, self.k)
        p2p_cost = self.cost_mlp(torch.cat([feat1_rep, feat2_knn, offset2], dim=1))
        p2n_cost = torch.sum(self.weight_net2(offset2) * p2p_cost, dim=3)               # [B,Cout,N]

        if knn_indices_1in1 is not None:
            assert knn_indices_1in1.shape[:2] == torch.Size([batch_size, n_points])
            assert knn_indices_1in1.shape[2] >= self.k
            knn_indices_1in1 = knn_indices_1in1[:, :, :self.k]
        else:
            knn_indices_1in1 = _ops.k_nearest_neighbor(input_xyz=xyz1, query_xyz=xyz1, k=self.k)
        offset1 = batch_indexing(xyz1, knn_indices_1in1) - centre
        n2n_cost = torch.sum(self.weight_net1(offset1) * batch_indexing(p2n_cost, knn_indices_1in1), dim=3)
        return self.feat_aligner(n2n_cost)


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
        flows_3d = []
        top = len(xyzs1) - 1
        for level in range(top, 0, -1):
            xyz1, xyz2 = xyzs1[level], xyzs2[level]
            knn1 = _ops.k_nearest_neighbor(xyz1, xyz1, k=16)
            bs, _, n_points = xyz1.shape
            if level == top:
                last_flow = torch.zeros([bs, 3, n_points], device=xyz1.device)
                xyz2_warp = xyz2
            else:
                last_flow = knn_interpolation(xyzs1[level + 1], flows_3d[-1], xyz1)
                xyz2_warp = backwarp_3d(xyz1, xyz2, last_flow)
            x = torch.cat([self.pyramid_feat_aligners[level](feats1_3d[level]),
                           self.correlations[level](xyz1, feats1_3d[level], xyz2_warp, feats2_3d[level], knn1),
                           last_flow], dim=1)
            _, flow_delta = self.flow_estimator(xyz1, x, knn1)
            flows_3d.append(last_flow + flow_delta)
        flows_3d = [f.float() for f in flows_3d][::-1]
        return [knn_interpolation(xyzs1[i + 1], flow, xyzs1[i]) for i, flow in enumerate(flows_3d)]

"""3-D (point) RAFT branch (counterpart of models/camliraft_l_core.py): point encoder, point
cost-volume pyramid with KNN lookup, set-conv GRU, motion encoder and flow head.
"""
import torch
import torch.nn as nn

from ..csrc import wrapper as _ops
from . import runtime
from .blocks import Conv1dNormRelu, MLP1d, MLP2d
from .geometry import backwarp_3d, batch_indexing, build_pc_pyramid, knn_interpolation
from .setconv import PointConv, PointConvDW, pass_cache

PYRAMID_SIZES = [4096, 2048, 1024, 512, 256]   # hard-coded in every CamLi* model (camliraft_l_core.py:174-176)


class Encoder3D(nn.Module):
    def __init__(self, n_channels, norm=None, k=16):
        super().__init__()
        self.level0_mlp = MLP1d(3, [n_channels[0], n_channels[0]])
        self.mlps = nn.ModuleList()
        self.convs = nn.ModuleList()
        for c_in, c_out in zip(n_channels[:-1], n_channels[1:]):
            self.mlps.append(MLP1d(c_in, [c_in, c_out]))
            self.convs.append(PointConv(c_out, c_out, norm=norm, k=k))

    def forward(self, xyzs):
        assert len(xyzs) == len(self.mlps) + 1
        feats = [self.level0_mlp(xyzs[0])]
        for i, (mlp, conv) in enumerate(zip(self.mlps, self.convs)):
            feats.append(conv(xyzs[i], mlp(feats[-1]), xyzs[i + 1]))
        return feats


class Correlation3D(nn.Module):
    """RAFT-style point cost volume (camliraft_l_core.py:40-101): dense [N,N] feature correlation,
    pooled over the target pyramid by k=3 neighbour averaging; each lookup takes the k=16 nearest
    warped targets per level, runs (dxyz, cost) through a small MLP and sums over neighbours."""

    def __init__(self, out_channels, k=16):
        super().__init__()
        self.k = k
        self.cost_mlp = MLP2d(4, [out_channels // 4, out_channels // 4], act='relu')
        self.merge = Conv1dNormRelu(out_channels, out_channels)
        self.cost_volume_pyramid = None

    def build_cost_volume_pyramid(self, feat1, feat2, xyzs2, k=3):
        volume = torch.bmm(feat1.float().transpose(1, 2), feat2.float()) / feat1.shape[1]   # [B,N,M0]
        self.cost_volume_pyramid = [volume]
        for i in range(1, len(xyzs2)):
            knn_indices = _ops.k_nearest_neighbor(xyzs2[i - 1], xyzs2[i], k=k)
            pooled = torch.mean(batch_indexing(self.cost_volume_pyramid[i - 1], knn_indices), dim=-1)
            self.cost_volume_pyramid.append(pooled)

    def calc_matching_cost(self, xyz1, xyz2, cost_volume):
        bs, n_points1, n_points2 = cost_volume.shape
        knn_cross = _ops.k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)       # [B,N,k]
        if runtime.fused() and not xyz1.requires_grad and not xyz2.requires_grad:
            from ..csrc import fused
            lookup = fused.corr3d_lookup_input(cost_volume, xyz1, xyz2, knn_cross)            # [B,4,N,k]
            return torch.sum(self.cost_mlp(lookup), dim=-1)
        knn_offset = batch_indexing(xyz2, knn_cross) - xyz1.view(bs, 3, n_points1, 1)
        knn_corr = batch_indexing(cost_volume.reshape(bs * n_points1, n_points2),
                                  knn_cross.reshape(bs * n_points1, self.k),
                                  layout='channel_last').reshape(bs, 1, n_points1, self.k)
        return torch.sum(self.cost_mlp(torch.cat([knn_offset, knn_corr], dim=1)), dim=-1)

    def forward(self, xyz1, xyzs2):
        costs = [self.calc_matching_cost(xyz1, xyzs2[lvl], self.cost_volume_pyramid[lvl]) for lvl in range(4)]
        return self.merge(torch.cat(costs, dim=1))


class FlowHead3D(nn.Module):
    def __init__(self, input_dim=128):
        super().__init__()
        self.conv1 = PointConvDW(input_dim, 128, k=32)
        self.conv2 = PointConvDW(128, 64, k=32)
        self.fc = nn.Conv1d(64, 3, kernel_size=1)

    def forward(self, xyz, features, knn_indices=None):
        features = self.conv1(xyz, features.float(), knn_indices=knn_indices)
        features = self.conv2(xyz, features, knn_indices=knn_indices)
        return self.fc(features)


class GRU3D(nn.Module):
    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.conv_z = PointConvDW(hidden_dim + input_dim, hidden_dim, act=None, k=4)
        self.conv_r = PointConvDW(hidden_dim + input_dim, hidden_dim, act=None, k=4)
        self.conv_q = PointConvDW(hidden_dim + input_dim, hidden_dim, act=None, k=4)

    def forward(self, xyz, h, x, knn_indices=None):
        h, x = h.float(), x.float()
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.conv_z(xyz, hx, knn_indices=knn_indices))
        r = torch.sigmoid(self.conv_r(xyz, hx, knn_indices=knn_indices))
        q = torch.tanh(self.conv_q(xyz, torch.cat([r * h, x], dim=1), knn_indices=knn_indices))
        return (1 - z) * h + z * q


class MotionEncoder3D(nn.Module):
    def __init__(self, corr_dim=128):
        super().__init__()
        self.conv_c1 = PointConvDW(corr_dim, corr_dim)
        self.conv_f1 = PointConvDW(3, 32, k=32)
        self.conv_f2 = PointConvDW(32, 16, k=16)
        self.conv = PointConvDW(corr_dim + 16, 128 - 3, k=16)

    def forward(self, xyz, flow, corr, knn_indices):
        corr, flow = corr.float(), flow.float()
        corr_feat = self.conv_c1(xyz, corr, knn_indices=knn_indices)
        flow_feat = self.conv_f2(xyz, self.conv_f1(xyz, flow, knn_indices=knn_indices), knn_indices=knn_indices)
        out = self.conv(xyz, torch.cat([corr_feat, flow_feat], dim=1), knn_indices=knn_indices)
        return torch.cat([out, flow], dim=1)


class CamLiRAFT_L_Core(nn.Module):
    """Point-only model: pyramid -> encoders at levels 0..2 -> RAFT iterations on the 2048-point
    level -> per-iteration interpolation back to the input cloud (camliraft_l_core.py:158-225)."""

    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.fnet = Encoder3D(n_channels=[64, 96, 128], norm='batch_norm', k=16)
        self.cnet = Encoder3D(n_channels=[64, 96, 128], norm='batch_norm', k=16)
        self.cnet_aligner = nn.Conv1d(128, 256, kernel_size=1)
        self.correlation = Correlation3D(out_channels=128, k=16)
        self.motion_encoder = MotionEncoder3D(corr_dim=128)
        self.gru = GRU3D(input_dim=128 + 128, hidden_dim=128)
        self.flow_head = FlowHead3D(input_dim=128)

    def forward(self, pc1, pc2):
        with pass_cache():
            return self._forward(pc1, pc2)

    def _forward(self, pc1, pc2):
        xyzs1, xyzs2, _, _ = build_pc_pyramid(pc1, pc2, PYRAMID_SIZES)
        feat1 = self.fnet(xyzs1[:3])[2]
        feat2 = self.fnet(xyzs2[:3])[2]
        featc = self.cnet_aligner(self.cnet(xyzs1[:3])[2])

        xyzs1, xyzs2 = xyzs1[2:], xyzs2[2:]
        xyz1, xyz2 = xyzs1[0], xyzs2[0]
        self.correlation.build_cost_volume_pyramid(feat1, feat2, xyzs2)

        h, x = torch.split(featc, [128, 128], dim=1)
        h, x = torch.tanh(h), torch.relu(x)
        knn_indices = _ops.k_nearest_neighbor(xyz1, xyz1, k=32)
        n_iters = self.cfgs.n_iters_train if self.training else self.cfgs.n_iters_eval

        flow_preds = []
        flow_pred = torch.zeros_like(xyz1)
        xyzs2_warp = xyzs2
        for it in range(n_iters):
            if it > 0:
                flow_pred = flow_pred.detach()
                xyzs2_warp = [backwarp_3d(xyz1, level, flow_pred) for level in xyzs2]
            corr = self.correlation(xyz1, xyzs2_warp)
            motion_feat = self.motion_encoder(xyz1, flow_pred, corr, knn_indices=knn_indices)
            h = self.gru(xyz1, h=h, x=torch.cat([x, motion_feat], dim=1), knn_indices=knn_indices)
            flow_pred = flow_pred + self.flow_head(xyz1, h, knn_indices).float()
            flow_preds.append(flow_pred)
        return [knn_interpolation(xyz1, flow, pc1, k=3) for flow in flow_preds]

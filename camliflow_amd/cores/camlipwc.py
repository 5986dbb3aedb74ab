"""CamLiPWC: the fused 2-D + 3-D PWC model, plus the single-modality wrappers (counterpart of
models/camlipwc_core.py, camlipwc.py, camlipwc_l.py, pwc.py, raft.py).
"""
import torch
import torch.nn as nn
from torch.nn.functional import interpolate, leaky_relu

from ..csrc import wrapper as _ops
from .blocks import Conv1dNormRelu, Conv2dNormRelu, flow_conv
from .camliraft import _FreezableBN, _camera_pair
from .fusion import CLFM
from .geometry import (InputPadder, backwarp_2d, backwarp_3d, build_pc_pyramid, flows_paral2persp, knn_interpolation,
                       paral2persp,
                       persp2paral, persp2paral_both, project_pc2image, resize_flow2d, resize_to_64x)
from .objectives import (FlowModel, calc_pyramid_loss_2d, calc_pyramid_loss_3d, calc_sequence_loss_2d)
from .pwc2d import (PYRAMID_CHANNELS_2D, ContextNetwork2D, FeaturePyramid2D, FlowEstimatorDense2D,
                    FlowEstimatorLite2D, PWCCore, finalize_flows_2d, pyramid_aligners, up_mask_head,
                    upsample_flow_x2)
from .pwc3d import PYRAMID_CHANNELS_3D, CamLiPWC_L_Core, Correlation3D, FeaturePyramid3D, FlowEstimator3D
from .raft2d import RAFTCore
from .raft3d import PYRAMID_SIZES


def _per_level(factory):
    """[Identity] + one module per pyramid level 1..5 built from its 3-D channel width."""
    return nn.ModuleList([nn.Identity()] + [factory(c) for c in PYRAMID_CHANNELS_3D[1:]])


class CamLiPWC_Core(nn.Module):
    def __init__(self, cfgs2d, cfgs3d, cfgs):
        super().__init__()
        self.cfgs, self.cfgs2d, self.cfgs3d = cfgs, cfgs2d, cfgs3d
        corr_2d = (2 * cfgs2d.max_displacement + 1) ** 2
        k = cfgs3d.k

        # image branch
        self.branch_2d_fnet = FeaturePyramid2D(PYRAMID_CHANNELS_2D, norm=cfgs2d.norm.feature_pyramid)
        self.branch_2d_fnet_aligners = pyramid_aligners(Conv2dNormRelu)
        estimator = FlowEstimatorLite2D if cfgs2d.lite_estimator else FlowEstimatorDense2D
        self.branch_2d_flow_estimator = estimator([64 + corr_2d + 2 + 32, 128, 128, 96, 64, 32],
                                                  norm=cfgs2d.norm.flow_estimator, conv_last=not cfgs.fuse_estimator)
        self.branch_2d_context_network = ContextNetwork2D(
            [self.branch_2d_flow_estimator.flow_feat_dim + 2, 128, 128, 128, 96, 64, 32],
            dilations=[1, 2, 4, 8, 16, 1], norm=cfgs2d.norm.context_network)
        self.branch_2d_up_mask_head = up_mask_head()

        # point branch
        self.branch_3d_fnet = FeaturePyramid3D(n_channels=PYRAMID_CHANNELS_3D, norm=cfgs3d.norm.feature_pyramid, k=k)
        self.branch_3d_fnet_aligners = pyramid_aligners(Conv1dNormRelu)
        self.branch_3d_correlations = _per_level(lambda c: Correlation3D(c, c, k=k))
        self.branch_3d_correlation_aligners = pyramid_aligners(Conv1dNormRelu)
        self.branch_3d_flow_estimator = FlowEstimator3D([64 + 64 + 3 + 64, 128, 128, 64], cfgs3d.norm.flow_estimator,
                                                        conv_last=not cfgs.fuse_estimator, k=k)

        # fusion
        if cfgs.fuse_pyramid:
            self.pyramid_clfms = _per_level(lambda c: CLFM(c, c, norm=cfgs2d.norm.feature_pyramid))
        if cfgs.fuse_correlation:
            self.corr_clfms = _per_level(lambda c: CLFM(corr_2d, c))
        if cfgs.fuse_estimator:
            dim_2d = self.branch_2d_flow_estimator.flow_feat_dim
            dim_3d = self.branch_3d_flow_estimator.flow_feat_dim
            self.estimator_clfm = CLFM(dim_2d, dim_3d)
            self.branch_2d_conv_last = nn.Conv2d(dim_2d, 2, kernel_size=3, stride=1, padding=1)
            self.branch_3d_conv_last = nn.Conv1d(dim_3d, 3, kernel_size=1)

    def encode(self, image, xyzs):
        return self.branch_2d_fnet(image), self.branch_3d_fnet(xyzs)

    @staticmethod
    def _project(xyz, camera_info, image_h, image_w):
        return project_pc2image(xyz, camera_info, grid_hw=(image_h, image_w))

    def decode(self, xyzs1, xyzs2, feats1_2d, feats2_2d, feats1_3d, feats2_3d, camera_info):
        assert len(xyzs1) == len(xyzs2) == len(feats1_2d) == len(feats2_2d) == len(feats1_3d) == len(feats2_3d)
        cfgs = self.cfgs
        flows_2d, flows_3d, flow_feats_2d, flow_feats_3d = [], [], [], []
        top = len(xyzs1) - 1

        for level in range(top, 0, -1):
            xyz1, feat1_2d, feat1_3d = xyzs1[level], feats1_2d[level], feats1_3d[level]
            xyz2, feat2_2d, feat2_3d = xyzs2[level], feats2_2d[level], feats2_3d[level]
            bs, image_h, image_w, n_points = feat1_2d.shape[0], feat1_2d.shape[2], feat1_2d.shape[3], xyz1.shape[-1]

            uv1 = self._project(xyz1, camera_info, image_h, image_w)
            uv2 = self._project(xyz2, camera_info, image_h, image_w)
            knn_xyz1 = _ops.k_nearest_neighbor(xyz1, xyz1, k=self.cfgs3d.k)

            if cfgs.fuse_pyramid:
                feat1_2d, feat1_3d = self.pyramid_clfms[level](uv1, feat1_2d, feat1_3d)
                feat2_2d, feat2_3d = self.pyramid_clfms[level](uv2, feat2_2d, feat2_3d)

            if level == top:
                def zeros(*shape):
                    return torch.zeros(shape, dtype=uv1.dtype, device=uv1.device)
                last_flow_2d, last_feat_2d = zeros(bs, 2, image_h, image_w), zeros(bs, 32, image_h, image_w)
                last_flow_3d, last_feat_3d = zeros(bs, 3, n_points), zeros(bs, 64, n_points)
                xyz2_warp, feat2_2d_warp = xyz2, feat2_2d
            else:
                last_flow_2d = upsample_flow_x2(flows_2d[-1])
                last_feat_2d = interpolate(flow_feats_2d[-1], scale_factor=2, mode='bilinear', align_corners=True)
                last_flow_3d, last_feat_3d = torch.split(
                    knn_interpolation(xyzs1[level + 1], torch.cat([flows_3d[-1], flow_feats_3d[-1]], dim=1), xyz1),
                    [3, 64], dim=1)
                feat2_2d_warp = backwarp_2d(feat2_2d, last_flow_2d, padding_mode='border')
                xyz2_warp = backwarp_3d(xyz1, xyz2, last_flow_3d)

            corr_3d = self.branch_3d_correlations[level](xyz1, feat1_3d, xyz2_warp, feat2_3d, knn_xyz1)
            corr_2d = leaky_relu(_ops.correlation2d(feat1_2d, feat2_2d_warp, self.cfgs2d.max_displacement), 0.1)
            if cfgs.fuse_correlation:
                corr_2d, corr_3d = self.corr_clfms[level](uv1, corr_2d, corr_3d)

            x_2d = torch.cat([corr_2d, self.branch_2d_fnet_aligners[level](feat1_2d), last_flow_2d, last_feat_2d], dim=1)
            x_3d = torch.cat([self.branch_3d_correlation_aligners[level](corr_3d),
                              self.branch_3d_fnet_aligners[level](feat1_3d), last_flow_3d, last_feat_3d], dim=1)

            if cfgs.fuse_estimator:
                flow_feat_2d = self.branch_2d_flow_estimator(x_2d)
                flow_feat_3d = self.branch_3d_flow_estimator(xyz1, x_3d, knn_xyz1)
                flow_feat_2d, flow_feat_3d = self.estimator_clfm(uv1, flow_feat_2d, flow_feat_3d)
                flow_delta_2d = flow_conv(self.branch_2d_conv_last, flow_feat_2d)
                flow_delta_3d = self.branch_3d_conv_last(flow_feat_3d)
            else:
                flow_feat_2d, flow_delta_2d = self.branch_2d_flow_estimator(x_2d)
                flow_feat_3d, flow_delta_3d = self.branch_3d_flow_estimator(xyz1, x_3d, knn_xyz1)

            flow_2d = last_flow_2d + flow_delta_2d
            flow_3d = last_flow_3d + flow_delta_3d
            flow_feat_2d, flow_delta_2d = self.branch_2d_context_network(torch.cat([flow_feat_2d, flow_2d], dim=1))
            flow_2d = flow_delta_2d + flow_2d

            flows_2d.append(torch.clip(flow_2d, min=-1000, max=1000))
            flows_3d.append(torch.clip(flow_3d, min=-100, max=100))
            flow_feats_2d.append(flow_feat_2d)
            flow_feats_3d.append(flow_feat_3d)

        flows_2d = finalize_flows_2d(flows_2d, self.branch_2d_up_mask_head(flow_feat_2d))
        flows_3d = [f.float() for f in flows_3d][::-1]
        flows_3d = [knn_interpolation(xyzs1[i + 1], flow, xyzs1[i]) for i, flow in enumerate(flows_3d)]
        return flows_2d, flows_3d


class CamLiPWC(_FreezableBN, FlowModel):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.core = CamLiPWC_Core(cfgs.pwc2d, cfgs.pwc3d, cfgs.fusion)

    def forward(self, inputs):
        images = inputs['images'].float() / 255.0
        pc1, pc2 = inputs['pcs'][:, :3], inputs['pcs'][:, 3:]
        origin_h, origin_w = images.shape[2:]
        images = resize_to_64x(images, None)[0]
        image1, image2 = images[:, :3], images[:, 3:]

        # the perspective camera keeps the ORIGINAL sensor size, the parallel one follows the resized image
        persp, _ = _camera_pair(origin_h, origin_w, inputs['intrinsics'])
        _, paral = _camera_pair(image1.shape[-2], image1.shape[-1], inputs['intrinsics'])
        pc1, pc2 = persp2paral_both(inputs['pcs'], persp, paral)

        xyzs1, xyzs2, sample_indices1, _ = build_pc_pyramid(pc1, pc2, PYRAMID_SIZES)
        feats1_2d, feats1_3d = self.core.encode(image1, xyzs1)
        feats2_2d, feats2_3d = self.core.encode(image2, xyzs2)
        flows_2d, flows_3d = self.core.decode(xyzs1, xyzs2, feats1_2d, feats2_2d, feats1_3d, feats2_3d, paral)
        flows_3d = [flows_paral2persp(xyz1, [f], persp, paral)[0] for xyz1, f in zip(xyzs1, flows_3d)]

        final = {'flow_2d': resize_flow2d(flows_2d[0], origin_h, origin_w), 'flow_3d': flows_3d[0]}
        return self.supervise(inputs, final, {
            'flow_2d': lambda target: calc_pyramid_loss_2d(flows_2d, target, self.cfgs.loss2d),
            'flow_3d': lambda target: calc_pyramid_loss_3d(flows_3d, target, self.cfgs.loss3d, sample_indices1)})

    RANKED_BY = 'epe2d'


class CamLiPWC_L(FlowModel):
    """Point-only PWC (models/camlipwc_l.py)."""

    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.core = CamLiPWC_L_Core(cfgs)

    def forward(self, inputs):
        pc1, pc2 = inputs['pcs'][:, :3], inputs['pcs'][:, 3:]
        persp, paral = _camera_pair(540, 960, inputs['intrinsics'])
        use_ids = self.cfgs.ids.enabled
        if use_ids:
            pc1, pc2 = persp2paral_both(inputs['pcs'], persp, paral)
        xyzs1, xyzs2, sample_indices1, _ = build_pc_pyramid(pc1, pc2, n_samples_list=PYRAMID_SIZES)
        flows_3d = self.core.decode(xyzs1, xyzs2, self.core.encode(xyzs1), self.core.encode(xyzs2))
        if use_ids:
            flows_3d = [flows_paral2persp(xyz1, [f], persp, paral)[0] for xyz1, f in zip(xyzs1, flows_3d)]
        return self.supervise(inputs, {'flow_3d': flows_3d[0]}, {
            'flow_3d': lambda target: calc_pyramid_loss_3d(flows_3d, target, self.cfgs.loss, sample_indices1)},
            targets={'flow_3d': inputs['flow_3d']} if 'flow_3d' in inputs else None)

    RANKED_BY = 'epe3d'


class PWC(FlowModel):
    """Image-only PWC (models/pwc.py)."""

    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.core = PWCCore(cfgs)

    def forward(self, inputs):
        images = inputs['images'].float() / 255.0
        origin_h, origin_w = images.shape[2:]
        images = resize_to_64x(images, None)[0]
        flows = self.core.decode(self.core.encode(images[:, :3]), self.core.encode(images[:, 3:]))
        return self.supervise(inputs, {'flow_2d': resize_flow2d(flows[0], origin_h, origin_w)}, {
            'flow_2d': lambda target: calc_pyramid_loss_2d(flows, target, self.cfgs.loss)})

    RANKED_BY = 'epe2d'


class RAFT(FlowModel):
    """Image-only RAFT (models/raft.py): images scaled to [-1, 1], padded to a multiple of 8."""

    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.core = RAFTCore(cfgs)

    def forward(self, inputs):
        images = 2 * (inputs['images'].float() / 255.0) - 1.0
        padder = InputPadder(images.shape, x=8)
        image1, image2 = padder.pad(images[:, :3], images[:, 3:])
        flow_preds = [padder.unpad(f) for f in self.core(image1, image2)]
        return self.supervise(inputs, {'flow_2d': flow_preds[-1]}, {
            'flow_2d': lambda target: calc_sequence_loss_2d(flow_preds, target, self.cfgs.loss)})

    RANKED_BY = 'epe2d'

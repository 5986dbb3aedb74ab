"""CamLiRAFT: the fused 2-D + 3-D RAFT model (counterpart of models/camliraft_core.py,
models/camliraft.py and models/camliraft_l.py).

Sub-module names (``core.branch_2d``, ``core.branch_3d``, ``clfm_*``) are a compatibility surface:
reference checkpoints key on them and its optimizer splits parameter groups on the prefix
``core.branch_3d`` (factory.py:50-58).
"""
import os

import torch
import torch.nn as nn

from ..csrc import wrapper as _ops
from . import runtime
from .fusion import CLFM
from .geometry import (InputPadder, backwarp_3d, backwarp_3d_levels, build_pc_pyramid, flows_paral2persp, knn_interpolation,
                       mesh_grid, paral2persp,
                       persp2paral, persp2paral_both, project_pc2image)
from .objectives import FlowModel, calc_sequence_loss_2d, calc_sequence_loss_3d
from .raft2d import RAFTCore
from .raft3d import PYRAMID_SIZES, CamLiRAFT_L_Core
from .setconv import pass_cache

_IMAGENET_MEAN = (123.675, 116.280, 103.530)
_IMAGENET_STD = (58.395, 57.120, 57.375)


class CamLiRAFT_Core(nn.Module):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.corr_levels = 4
        self.corr_radius = 4
        self.branch_2d = RAFTCore(cfgs)
        self.branch_3d = CamLiRAFT_L_Core(cfgs)
        if cfgs.fuse_fnet:
            self.clfm_fnet = CLFM(128, 128, norm='batch_norm')
        if cfgs.fuse_cnet:
            self.clfm_cnet = CLFM(128, 128, norm='batch_norm')
        if cfgs.fuse_corr:
            self.clfm_corr = CLFM(81 * 4, 128)
        if cfgs.fuse_motion:
            self.clfm_motion = CLFM(128, 128)
        if cfgs.fuse_hidden:
            self.clfm_hidden = CLFM(128, 128)

    def _project_to_feature_grid(self, xyz, camera_info, feat_hw):
        return project_pc2image(xyz, camera_info, grid_hw=feat_hw)

    def forward(self, image1, image2, pc1, pc2, camera_info, flow_rows=None):
        with pass_cache():   # iteration-invariant set-conv weights live for exactly one pass
            return self._forward(image1, image2, pc1, pc2, camera_info, flow_rows)

    def _forward(self, image1, image2, pc1, pc2, camera_info, flow_rows=None):
        """Op order, detach points and module call sequence are those of camliraft_core.py:33-145.
        The image branch is issued on the current stream, the point branch inside ``lanes.side()``
        (a second HIP stream when ``runtime.overlap()`` is on, otherwise a no-op context); the two
        only meet at the CLFM fusion points, where ``to_main`` / ``to_side`` order the streams."""
        b2d, b3d, cfgs = self.branch_2d, self.branch_3d, self.cfgs
        lanes = runtime.Lanes(image1.device, key=(tuple(image1.shape), tuple(pc1.shape), self.training))
        feat_hw = (image1.shape[-2] // 8, image1.shape[-1] // 8)

        lanes.to_side(pc1, pc2)
        with lanes.side():
            xyzs1, xyzs2, _, _ = build_pc_pyramid(pc1, pc2, PYRAMID_SIZES)
            feat1_3d = b3d.fnet(xyzs1[:3])[2]
            feat2_3d = b3d.fnet(xyzs2[:3])[2]
            featc_3d = b3d.cnet(xyzs1[:3])[2]
            xyzs1, xyzs2 = xyzs1[2:], xyzs2[2:]          # working pyramid [2048, 1024, 512, 256]
            xyz1, xyz2 = xyzs1[0], xyzs2[0]
            uv1 = self._project_to_feature_grid(xyz1, camera_info, feat_hw)
            uv2 = self._project_to_feature_grid(xyz2, camera_info, feat_hw)
        # the context encoder is an independent chain (its own ResNet trunk on frame 1): auxiliary stream, so that its
        # HBM-bound epilogues run under the feature encoder's convolutions and vice versa (runtime.Branch, slot 3)
        cnet_branch = runtime.Branch(image1, slot=3)
        with cnet_branch:
            featc_2d = b2d.cnet(image1)
        if runtime.fused():
            # both frames through the feature encoder in one batch: its BatchNorms always run in eval
            # mode (norm_eval) and `align` has no norm, so per-sample results are unchanged
            feat1_2d, feat2_2d = torch.chunk(b2d.fnet(torch.cat([image1, image2], dim=0)), 2, dim=0)
        else:
            feat1_2d, feat2_2d = b2d.fnet(image1), b2d.fnet(image2)
        cnet_branch.join(featc_2d)
        assert tuple(feat1_2d.shape[-2:]) == feat_hw

        lanes.to_main(feat1_3d, feat2_3d, featc_3d, uv1, uv2)
        if cfgs.fuse_fnet:
            feat1_2d, feat1_3d = self.clfm_fnet(uv1, feat1_2d, feat1_3d)
            feat2_2d, feat2_3d = self.clfm_fnet(uv2, feat2_2d, feat2_3d)
        if cfgs.fuse_cnet:
            featc_2d, featc_3d = self.clfm_cnet(uv1, featc_2d, featc_3d)

        lanes.to_side(feat1_3d, feat2_3d, featc_3d)
        with lanes.side():
            h_3d, x_3d = torch.split(b3d.cnet_aligner(featc_3d), [128, 128], dim=1)
            h_3d, x_3d = torch.tanh(h_3d), torch.relu(x_3d)
            b3d.correlation.build_cost_volume_pyramid(feat1_3d, feat2_3d, xyzs2, nested=True)   # build_pc_pyramid: FPS prefixes
            knn_indices = _ops.k_nearest_neighbor(xyz1, xyz1, k=32)
            flow_3d_pred = torch.zeros_like(xyz1)
        h_2d, x_2d = torch.split(b2d.cnet_aligner(featc_2d), [128, 128], dim=1)
        h_2d, x_2d = torch.tanh(h_2d), torch.relu(x_2d)
        b2d.correlation.build_cost_volume_pyramid(feat1_2d, feat2_2d)
        gru2d_state = b2d.gru.prepare(x_2d) if runtime.fused() else None

        n_iters = cfgs.n_iters_train if self.training else cfgs.n_iters_eval
        bs = image1.shape[0]
        grid_coords = mesh_grid(bs, feat_hw[0], feat_hw[1], device=image1.device)
        flow_2d_pred = torch.zeros_like(grid_coords)
        xyzs2_warp = xyzs2

        flow_2d_preds, flow_3d_preds = [], []
        for it in range(n_iters):
            # ---- correlation lookups -----------------------------------------------------------
            with lanes.side():
                if it > 0:
                    flow_3d_pred = flow_3d_pred.detach()
                    xyzs2_warp = backwarp_3d_levels(xyz1, xyzs2, flow_3d_pred, nested=True)
                corr3d = b3d.correlation(xyz1, xyzs2_warp)
            if it > 0:
                flow_2d_pred = flow_2d_pred.detach()
            flow_branch = b2d.motion_encoder.begin(flow_2d_pred) if runtime.fused() else None     # aux stream (runtime.Branch)
            corr2d = b2d.correlation(grid_coords + flow_2d_pred)
            if cfgs.fuse_corr:
                lanes.to_main(corr3d)
                corr2d, corr3d = self.clfm_corr(uv1, corr2d, corr3d)
                lanes.to_side(corr3d)

            # ---- motion features ---------------------------------------------------------------
            with lanes.side():
                motion_feat3d = b3d.motion_encoder(xyz1, flow_3d_pred, corr3d, knn_indices=knn_indices)
            motion_feat2d = b2d.motion_encoder(flow_2d_pred, corr2d, flow_branch=flow_branch)
            if cfgs.fuse_motion:
                lanes.to_main(motion_feat3d)
                motion_feat2d, motion_feat3d = self.clfm_motion(uv1, motion_feat2d, motion_feat3d)
                lanes.to_side(motion_feat3d)

            # ---- recurrent update --------------------------------------------------------------
            with lanes.side():
                h_3d = b3d.gru(xyz1, h=h_3d, x=torch.cat([x_3d, motion_feat3d], dim=1), knn_indices=knn_indices)
            if gru2d_state is not None:
                h_2d = b2d.gru.step(h_2d, motion_feat2d, gru2d_state)
            else:
                h_2d = b2d.gru(h=h_2d, x=torch.cat([x_2d, motion_feat2d], dim=1))
            if cfgs.fuse_hidden:
                lanes.to_main(h_3d)
                h_2d, h_3d = self.clfm_hidden(uv1, h_2d, h_3d)
                lanes.to_side(h_3d)

            # ---- flow heads --------------------------------------------------------------------
            with lanes.side():
                flow_3d_pred = flow_3d_pred + b3d.flow_head(xyz1, h_3d, knn_indices)
                flow_3d_preds.append(knn_interpolation(xyz1, flow_3d_pred, pc1, k=3, invariant_input=True, invariant_query=True))
            mask_branch = b2d.convex_upsampler.begin(h_2d) if runtime.fused() else None           # aux stream
            flow_2d_pred = flow_2d_pred + b2d.flow_head(h_2d)
            flow_2d_preds.append(b2d.convex_upsampler.finish(mask_branch, h_2d, flow_2d_pred, flow_rows))

        lanes.to_main(flow_3d_preds)
        b2d.correlation.release()
        b3d.correlation.release()
        return flow_2d_preds, flow_3d_preds


def _camera_pair(sensor_h, sensor_w, intrinsics):
    """Perspective camera of the (padded) image and the 1/32-scale parallel camera the point
    branch works in (camliraft.py:48-62)."""
    persp = {'projection_mode': 'perspective', 'sensor_h': sensor_h, 'sensor_w': sensor_w,
             'f': intrinsics[:, 0], 'cx': intrinsics[:, 1], 'cy': intrinsics[:, 2]}
    paral_h, paral_w = round(sensor_h / 32), round(sensor_w / 32)
    paral = {'projection_mode': 'parallel', 'sensor_h': paral_h, 'sensor_w': paral_w,
             'cx': (paral_w - 1) / 2, 'cy': (paral_h - 1) / 2}
    return persp, paral


class _FreezableBN:
    """``train()`` that optionally keeps every BatchNorm in eval mode (camliraft.py:17-30)."""

    def train(self, mode=True):
        self.training = mode
        for module in self.children():
            module.train(mode)
        if self.cfgs.freeze_bn:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def eval(self):
        return self.train(False)


class CamLiRAFT(_FreezableBN, FlowModel):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.core = CamLiRAFT_Core(cfgs)
        # normalisation constants live on the device (non-persistent: not part of the state dict);
        # the reference re-uploads them from Python lists on every forward (camliraft.py:41-42)
        self.register_buffer('_norm_mean', torch.tensor(_IMAGENET_MEAN).reshape(1, 3, 1, 1), persistent=False)
        self.register_buffer('_norm_std', torch.tensor(_IMAGENET_STD).reshape(1, 3, 1, 1), persistent=False)

    def forward(self, inputs):
        images = inputs['images'].float()
        pc1, pc2 = inputs['pcs'][:, :3], inputs['pcs'][:, 3:]

        padder = InputPadder(images.shape, x=8)
        on_hip = runtime.fused() and images.is_cuda
        if on_hip and torch.is_grad_enabled() and images.requires_grad:
            # the kernel has no autograd node: input-gradient uses (saliency, adversarial images) keep the composition
            runtime.fallback('pad_normalize', 'differentiable input images')
            on_hip = False
        if on_hip:      # both frames padded + normalised in one pass (camli_pad_normalize)
            from ..csrc import fused
            image1, image2 = fused.pad_normalize(images, padder._pad, _IMAGENET_MEAN, _IMAGENET_STD)
        else:
            image1, image2 = padder.pad(images[:, :3], images[:, 3:])
            mean, std = self._norm_mean, self._norm_std
            image1 = (image1 - mean) / std
            image2 = (image2 - mean) / std

        persp, paral = _camera_pair(image1.shape[-2], image1.shape[-1], inputs['intrinsics'])
        pc1, pc2 = persp2paral_both(inputs['pcs'], persp, paral)     # one launch for both clouds on the product path

        # images padded at the bottom only (960x540 -> 544 rows): the up-sampler writes the 540 rows the un-padding would keep
        rows = images.shape[2] if (padder.bottom_only() and runtime.fused() and images.is_cuda
                                   and os.environ.get('CAMLI_UPSAMPLE_CROP', '1') == '1') else None
        flow_2d_preds, flow_3d_preds = self.core(image1, image2, pc1, pc2, paral, flow_rows=rows)
        flow_2d_preds = [padder.unpad(f) for f in flow_2d_preds]
        flow_3d_preds = flows_paral2persp(pc1, flow_3d_preds, persp, paral)

        return self.supervise(inputs, {'flow_2d': flow_2d_preds[-1], 'flow_3d': flow_3d_preds[-1]}, {
            'flow_2d': lambda target: calc_sequence_loss_2d(flow_2d_preds, target, cfgs=self.cfgs.loss2d),
            'flow_3d': lambda target: calc_sequence_loss_3d(flow_3d_preds, target, cfgs=self.cfgs.loss3d)})

    RANKED_BY = 'epe2d'


class CamLiRAFT_L(FlowModel):
    """Point-cloud-only variant (models/camliraft_l.py:7-81).  The cameras are those of a fixed
    540x960 sensor; ``cfgs.ids.enabled`` switches the inverse-depth-scaling transform; optional
    ``src_mean/src_std/dst_mean/dst_std`` inputs re-standardise the clouds around the core."""

    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.core = CamLiRAFT_L_Core(cfgs)

    def forward(self, inputs):
        pc1, pc2 = inputs['pcs'][:, :3], inputs['pcs'][:, 3:]
        use_ids = self.cfgs.ids.enabled
        persp, paral = _camera_pair(540, 960, inputs['intrinsics'])
        if use_ids:
            pc1, pc2 = persp2paral_both(inputs['pcs'], persp, paral)

        restandardise = 'src_mean' in inputs and 'dst_mean' in inputs
        if restandardise:
            src_mean, dst_mean = inputs['src_mean'][..., None], inputs['dst_mean'][..., None]
            src_std, dst_std = inputs['src_std'][..., None], inputs['dst_std'][..., None]

            def to_dst(pc):
                return ((pc - src_mean) / src_std) * dst_std + dst_mean

            def to_src(pc):
                return ((pc - dst_mean) / dst_std) * src_std + src_mean
            pc1, pc2 = to_dst(pc1), to_dst(pc2)

        flow_preds = self.core.forward(pc1, pc2)

        if restandardise:
            flow_preds = [to_src(pc1 + f) - to_src(pc1) for f in flow_preds]
            pc1 = to_src(pc1)
        if use_ids:
            flow_preds = flows_paral2persp(pc1, flow_preds, persp, paral)

        # the ground truth may carry a validity channel: this model trains and scores on the first three (camliraft_l.py:70)
        return self.supervise(inputs, {'flow_3d': flow_preds[-1]}, {
            'flow_3d': lambda target: calc_sequence_loss_3d(flow_preds, target, self.cfgs.loss)},
            targets={'flow_3d': inputs['flow_3d'][:, :3]} if 'flow_3d' in inputs else None)

    RANKED_BY = 'epe3d'

"""2-D PWC branch (counterpart of models/pwc_core.py): stride-2 residual feature pyramid, warped
local correlation (the ``correlation2d`` boundary operator), dense/lite flow estimator, dilated
context network, coarse-to-fine decoding with a convex-upsampled finest level.
"""
import torch
import torch.nn as nn
from torch.nn.functional import interpolate, leaky_relu

from ..csrc import wrapper as _ops
from .blocks import Conv2dNormRelu, flow_conv
from .geometry import backwarp_2d, convex_upsample

PYRAMID_CHANNELS_2D = [3, 16, 32, 64, 96, 128, 192]


def _conv3x3(cin, cout, norm, **kw):
    return Conv2dNormRelu(cin, cout, kernel_size=3, padding=kw.pop('padding', 1), norm=norm, **kw)


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, down_sample=True, norm=None):
        super().__init__()
        stride = 2 if down_sample else 1
        self.down0 = (Conv2dNormRelu(in_channels, out_channels, stride=2, norm=norm, act=None)
                      if down_sample else nn.Identity())
        self.conv0 = _conv3x3(in_channels, out_channels, norm, stride=stride)
        self.conv1 = _conv3x3(out_channels, out_channels, norm, stride=1, act=None)
        self.relu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def forward(self, x):
        return self.relu(self.conv1(self.conv0(x)) + self.down0(x))


class FeaturePyramid2D(nn.Module):
    def __init__(self, n_channels, norm=None):
        super().__init__()
        self.pyramid_convs = nn.ModuleList(ResidualBlock(a, b, norm=norm) for a, b in zip(n_channels[:-1], n_channels[1:]))

    def forward(self, x):
        outputs = []
        for conv in self.pyramid_convs:
            x = conv(x)
            outputs.append(x)
        return outputs


class _FlowEstimator2D(nn.Module):
    """Five 3x3 convs; ``inputs_of(i)`` says which earlier activations conv i+1 sees."""

    def _finish(self, flow_feat):
        if self.conv_last is None:
            return flow_feat
        return flow_feat, flow_conv(self.conv_last, flow_feat)

    def _make_last(self, conv_last):
        self.conv_last = nn.Conv2d(self.flow_feat_dim, 2, kernel_size=3, stride=1, padding=1) if conv_last else None


class FlowEstimatorLite2D(_FlowEstimator2D):
    """Each conv sees the two previous activations (pwc_core.py:48-75)."""

    def __init__(self, n_channels, norm=None, conv_last=True):
        super().__init__()
        c = n_channels
        self.conv1 = _conv3x3(c[0], c[1], norm)
        self.conv2 = _conv3x3(c[1], c[2], norm)
        self.conv3 = _conv3x3(c[1] + c[2], c[3], norm)
        self.conv4 = _conv3x3(c[2] + c[3], c[4], norm)
        self.conv5 = _conv3x3(c[3] + c[4], c[5], norm)
        self.flow_feat_dim = c[4] + c[5]
        self._make_last(conv_last)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(torch.cat([x1, x2], dim=1))
        x4 = self.conv4(torch.cat([x2, x3], dim=1))
        x5 = self.conv5(torch.cat([x3, x4], dim=1))
        return self._finish(torch.cat([x4, x5], dim=1))


class FlowEstimatorDense2D(_FlowEstimator2D):
    """DenseNet-style: every conv sees the input and all earlier activations (pwc_core.py:78-124)."""

    def __init__(self, n_channels, norm=None, conv_last=True):
        super().__init__()
        seen = 0
        for i in range(5):
            seen += n_channels[i]
            setattr(self, 'conv%d' % (i + 1), _conv3x3(seen, n_channels[i + 1], norm))
        self.flow_feat_dim = sum(n_channels)
        self._make_last(conv_last)

    def forward(self, x):
        for i in range(1, 6):
            x = torch.cat([getattr(self, 'conv%d' % i)(x), x], dim=1)
        return self._finish(x)


class ContextNetwork2D(nn.Module):
    def __init__(self, n_channels, dilations, norm=None):
        super().__init__()
        self.convs = nn.ModuleList(
            Conv2dNormRelu(a, b, kernel_size=3, padding=d, dilation=d, norm=norm)
            for a, b, d in zip(n_channels[:-1], n_channels[1:], dilations))
        self.conv_last = nn.Conv2d(n_channels[-1], 2, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return x, flow_conv(self.conv_last, x)


def pyramid_aligners(conv_cls):
    """Identity for level 0, then 1x1 convs bringing [32,64,96,128,192] channels to 64."""
    return nn.ModuleList([nn.Identity()] + [conv_cls(c, 64) for c in (32, 64, 96, 128, 192)])


def up_mask_head():
    return nn.Sequential(nn.Conv2d(32, 64, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=True),
                         nn.Conv2d(64, 4 * 4 * 9, kernel_size=1, stride=1, padding=0))


def upsample_flow_x2(flow):
    return interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=True)


def finalize_flows_2d(flows_2d, mask):
    """coarse-to-fine list -> fine-to-coarse, finest level convex-upsampled x4, the others bilinear x4
    (pwc_core.py:216-223)."""
    flows_2d = [f.float() for f in flows_2d][::-1]
    flows_2d[0] = convex_upsample(flows_2d[0], mask, scale_factor=4)
    for i in range(1, len(flows_2d)):
        flows_2d[i] = interpolate(flows_2d[i] * 4, scale_factor=4, mode='bilinear', align_corners=True)
    return flows_2d


class PWCCore(nn.Module):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        corr_channels = (cfgs.max_displacement * 2 + 1) ** 2
        self.feature_pyramid = FeaturePyramid2D(PYRAMID_CHANNELS_2D, norm=cfgs.norm.feature_pyramid)
        self.pyramid_feature_aligners = pyramid_aligners(Conv2dNormRelu)
        estimator = FlowEstimatorLite2D if cfgs.lite_estimator else FlowEstimatorDense2D
        self.flow_estimator = estimator([64 + corr_channels + 2, 128, 128, 96, 64, 32], norm=cfgs.norm.flow_estimator)
        self.context_network = ContextNetwork2D([self.flow_estimator.flow_feat_dim + 2, 128, 128, 128, 96, 64, 32],
                                                [1, 2, 4, 8, 16, 1], norm=cfgs.norm.context_network)
        self.up_mask_head = up_mask_head()

    def encode(self, image):
        return self.feature_pyramid(image)

    def decode(self, feats1_2d, feats2_2d):
        assert len(feats1_2d) == len(feats2_2d)
        flows_2d = []
        for level in range(len(feats1_2d) - 1, 0, -1):
            feat1, feat2 = feats1_2d[level], feats2_2d[level]
            bs, _, image_h, image_w = feat1.shape
            if not flows_2d:
                last_flow = torch.zeros([bs, 2, image_h, image_w], dtype=feat1.dtype, device=feat1.device)
                feat2_warp = feat2
            else:
                last_flow = upsample_flow_x2(flows_2d[-1])
                feat2_warp = backwarp_2d(feat2, last_flow, padding_mode='border')
            corr = leaky_relu(_ops.correlation2d(feat1, feat2_warp, self.cfgs.max_displacement), 0.1)
            x = torch.cat([corr, self.pyramid_feature_aligners[level](feat1), last_flow], dim=1)
            flow_feat, flow_delta = self.flow_estimator(x)
            flow = flow_delta + last_flow
            flow_feat, flow_delta = self.context_network(torch.cat([flow_feat, flow], dim=1))
            flows_2d.append(flow_delta + flow)
        return finalize_flows_2d(flows_2d, self.up_mask_head(flow_feat))

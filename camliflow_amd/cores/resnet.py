"""ResNet stem + first two stages, the image encoder trunk of the 2-D branch.

The reference takes this trunk from a third-party dependency that is not vendored:
``mmdet==2.14.0`` ``mmdet.models.backbones.ResNet(depth=50, num_stages=2, strides=(1, 2),
dilations=(1, 1), out_indices=(1,), norm_eval=True)`` (call site models/raft_core.py:10-25,
pin README.md:78-79).  This file restates the published architecture (He et al. 2016, "pytorch"
style: the stride sits on the 3x3 conv) with mmdet's parameter names -- ``conv1, bn1,
layer{1,2}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}`` -- so reference
checkpoints load.  Parity for this sub-module is UNPINNED (the dependency's source is absent from
the reference tree); it is plain conv/BN/ReLU/max-pool executed by MIOpen and is not a HIP-kernel
target.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import runtime

_STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
_TRUNK_CHANNELS_LAST = os.environ.get('CAMLI_TRUNK_NHWC', '1') == '1'


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, with_downsample=False):
        super().__init__()
        out_planes = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out_planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, out_planes, 1, stride=stride, bias=False),
                nn.BatchNorm2d(out_planes),
            )

    def forward(self, x, folded=None):
        if _foldable(self.bn1, x):
            if (self.downsample is None and _FORK and _is_channels_last(x) and x.dtype == torch.float32
                    and torch.is_grad_enabled() and x.requires_grad and not torch.is_autocast_enabled()):
                # identity block: conv1 and the shortcut leave through one node whose adjoint accumulates in place
                y, shortcut = _conv_bn(self.conv1, self.bn1, x, 'relu', folded, fork=True)
            else:
                shortcut = x if self.downsample is None else _conv_bn(self.downsample[0], self.downsample[1], x, None, folded)
                y = _conv_bn(self.conv1, self.bn1, x, 'relu', folded)
            y = _conv_bn(self.conv2, self.bn2, y, 'relu', folded)
            # bn3's bias, the shortcut and the block's closing ReLU in ONE pass over conv3's output (camli_bias_act_res_fwd);
            # as bias pass + add + relu they were 7 tensor streams over the largest activations of the model
            return _conv_bn(self.conv3, self.bn3, y, 'relu', folded, residual=shortcut)
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + shortcut)


class _PointwiseFork(torch.autograd.Function):
    """``conv1`` of an identity bottleneck on a channels-last map, returned together with the block input it shares with the
    shortcut: ``y, x = fork(x, w)``.  The point is the adjoint.  Autograd would compute the data gradient of the 1x1
    convolution into a fresh tensor and then ADD the shortcut's gradient to it (three passes over the block's widest
    tensor); here the data gradient is a GEMM with beta = 1 INTO the shortcut's gradient -- ``g_short += gy W`` on the
    [pixels, C] matrices the channels-last tensors are -- one read-modify-write.  The shortcut gradient is the tensor the
    closing epilogue's adjoint produced (``_BiasActNHWC.backward``); its other reader, conv3's adjoint, was enqueued on the
    same stream before this node runs, so overwriting it here is ordered after that read."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.args = ([1, 1], [0, 0], [1, 1], False, [0, 0], 1)
        y = torch.ops.aten.convolution(x, w, None, *ctx.args)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, g_short):
        x, w = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad
        gx = gw = None
        fused_ok = (need_x and g_short is not None and g_short.dtype == torch.float32 and gy.dtype == torch.float32
                    and _is_channels_last(g_short) and _is_channels_last(gy))
        if need_w or (need_x and not fused_ok):
            got = torch.ops.aten.convolution_backward(gy, x, w, None, *ctx.args, [need_x and not fused_ok, need_w, False])
            gx, gw = got[0], got[1]
        if need_x:
            if fused_ok:
                flat = g_short.permute(0, 2, 3, 1).reshape(-1, g_short.shape[1])            # [pixels, Cin]: a view
                flat.addmm_(gy.permute(0, 2, 3, 1).reshape(-1, gy.shape[1]), w.flatten(1))
                gx = g_short
            elif g_short is not None:
                gx = gx + g_short
        return gx, gw


def _is_channels_last(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


_FORK = os.environ.get('CAMLI_TRUNK_FORK', '1') != '0'


def _foldable(bn, x):
    """Frozen-statistics BatchNorm (norm_eval) on the product path: it is an affine map per channel
    and can be folded into the preceding convolution."""
    from .blocks import epilogue_ok
    return (not bn.training) and epilogue_ok(x)


def _conv_bn(conv, bn, x, act, folded=None, residual=None, fork=False):
    """conv -> BatchNorm(eval) -> act as ONE convolution with folded weights plus the fused bias /
    activation epilogue:  y = conv(x, w * s) + (beta - mean * s),  s = gamma / sqrt(var + eps).
    gamma / beta stay trainable (the fold is differentiated by autograd on the small tensors); no
    BatchNorm pass over the activations remains, forward or backward.  Equal to the unfolded form up
    to fp32 rounding.  ``folded`` = the trunk's table of (s, beta - mean * s) per BatchNorm, computed
    for all layers at once (ResNetTrunk._fold_table): same values, 5 launches instead of 5 per layer."""
    from ..csrc import fused
    if folded is not None and id(bn) in folded:
        scale, bias = folded[id(bn)]
    else:
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        bias = bn.bias - bn.running_mean * scale
    weight = conv.weight * scale.view(-1, 1, 1, 1)
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        weight = weight.contiguous(memory_format=torch.channels_last)
    if fork:        # 1x1, stride 1: (act(conv + bias), the input for the shortcut) -- see _PointwiseFork
        y, shortcut = _PointwiseFork.apply(x, weight)
        return fused.bias_act(y, bias, act), shortcut
    y = F.conv2d(x, weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if residual is not None:
        return fused.bias_act_res(y, bias, residual, act)
    return fused.bias_act(y, bias, act)


class ResNetTrunk(nn.Module):
    """conv1/bn1/maxpool + ``num_stages`` bottleneck stages; returns a 1-tuple like mmdet does."""

    def __init__(self, depth=50, num_stages=2, strides=(1, 2), norm_eval=True, **_unused):
        super().__init__()
        if depth not in _STAGE_BLOCKS:
            raise KeyError('invalid depth %s for the bottleneck trunk' % depth)
        self.norm_eval = norm_eval
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)

        inplanes = 64
        self.res_layers = []
        for i in range(num_stages):
            planes = 64 * 2 ** i
            blocks = []
            for j in range(_STAGE_BLOCKS[depth][i]):
                stride = strides[i] if j == 0 else 1
                need_ds = j == 0 and (stride != 1 or inplanes != planes * Bottleneck.expansion)
                blocks.append(Bottleneck(inplanes, planes, stride, need_ds))
                inplanes = planes * Bottleneck.expansion
            name = 'layer%d' % (i + 1)
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self.feat_dim = inplanes

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)  # mmdet default zero_init_residual=True

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def _fold_table(self):
        """{id(bn): (scale, bias)} for every BatchNorm of the trunk from ONE vectorised evaluation:
        gamma and beta are concatenated (autograd splits the gradient back), the frozen statistics
        (rsqrt(var + eps), mean) are concatenated once and reused until a buffer is written again."""
        bns = getattr(self, '_bn_list', None)
        if bns is None:
            bns = self._bn_list = [m for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        versions = tuple(t._version for bn in bns for t in (bn.running_var, bn.running_mean))
        stats = getattr(self, '_frozen_stats', None)
        if stats is None or stats[0] != versions or stats[1].device != bns[0].weight.device:
            with torch.no_grad():
                inv_std = torch.cat([torch.rsqrt(bn.running_var + bn.eps) for bn in bns])
                mean = torch.cat([bn.running_mean for bn in bns])
            stats = self._frozen_stats = (versions, inv_std, mean)
        scale = torch.cat([bn.weight for bn in bns]) * stats[1]
        bias = torch.cat([bn.bias for bn in bns]) - stats[2] * scale
        sizes = [bn.num_features for bn in bns]
        return {id(bn): pair for bn, pair in zip(bns, zip(torch.split(scale, sizes), torch.split(bias, sizes)))}

    def forward(self, x, keep_channels_last=False):
        """``keep_channels_last``: the caller's next op reads the channels-last map as it is (Encoder2D's 1x1 ``align``,
        a GEMM): the conversion back at the trunk's output is skipped."""
        if _foldable(self.bn1, x):
            folded = self._fold_table()
            from ..csrc import fused
            x = _conv_bn(self.conv1, self.bn1, x, 'relu', folded)
            pool = self.maxpool
            if (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False):
                x = fused.maxpool3x3s2(x)      # own kernels: torch's pooling adjoint alone took 1.5 ms of a step
            else:
                x = pool(x)
            if _TRUNK_CHANNELS_LAST:
                # the bottleneck stages run channels-last: MIOpen's implicit-GEMM kernels are NHWC kernels and, fed NCHW
                # tensors, wrap every call in layout transposes (profiles/r03_trunk_conv_layout_microbench.txt: 22.1 ->
                # 18.1 ms per forward+backward of the stages at batch 16).  The stem stays NCHW (its 7x7 weight gradient
                # is 2x slower channels-last); one conversion of the pooled map here, one back at the trunk's output.
                x = x.contiguous(memory_format=torch.channels_last)
            for name in self.res_layers:
                for block in getattr(self, name):
                    x = block(x, folded)
            if _TRUNK_CHANNELS_LAST and not keep_channels_last:
                x = x.contiguous()
            return (x,)
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for name in self.res_layers:
            x = getattr(self, name)(x)
        return (x,)

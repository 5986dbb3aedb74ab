"""1x1-convolution building blocks shared by every core (counterpart of models/mlp.py:1-162).

Parameter names (``conv_fn``, ``norm_fn``, ``convs.N``) are those of the reference so that its
checkpoints load; the classes are generated from one dimension-generic implementation.
"""
import os

import torch
import torch.nn as nn

from . import runtime


class _LayerNormCF(nn.Module):
    """LayerNorm over the channel axis of a channel-first tensor (mlp.py:5-40), eps inside sqrt."""
    extra_dims = 1

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        mean = x.mean(1, keepdim=True)
        var = (x - mean).pow(2).mean(1, keepdim=True)
        x = (x - mean) / torch.sqrt(var + self.eps)
        shape = (-1,) + (1,) * self.extra_dims
        return self.weight.view(shape) * x + self.bias.view(shape)


class LayerNormCF1d(_LayerNormCF):
    extra_dims = 1


class LayerNormCF2d(_LayerNormCF):
    extra_dims = 2


def make_activation(act):
    if act == 'relu':
        return nn.ReLU(inplace=True)
    if act == 'leaky_relu':
        return nn.LeakyReLU(negative_slope=0.1, inplace=True)
    if act == 'sigmoid':
        return nn.Sigmoid()
    if act is None:
        return nn.Identity()
    raise NotImplementedError('Unknown activation function: %s' % act)


def _make_norm(norm, channels, dims):
    bn, inorm, ln = ((nn.BatchNorm1d, nn.InstanceNorm1d, LayerNormCF1d) if dims == 1
                     else (nn.BatchNorm2d, nn.InstanceNorm2d, LayerNormCF2d))
    if norm == 'batch_norm':
        return bn(channels)
    if norm == 'instance_norm':
        return inorm(channels)
    if norm == 'instance_norm_affine':
        return inorm(channels, affine=True)
    if norm == 'layer_norm':
        return ln(channels)
    if norm is None:
        return nn.Identity()
    raise NotImplementedError('Unknown normalization function: %s' % norm)


class _ConvNormAct(nn.Module):
    dims = 1

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1,
                 norm=None, act='leaky_relu'):
        super().__init__()
        conv = nn.Conv1d if self.dims == 1 else nn.Conv2d
        self.conv_fn = conv(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                            dilation=dilation, groups=groups, bias=norm is None)
        self.norm_fn = _make_norm(norm, out_channels, self.dims)
        self.act_fn = make_activation(act)
        self._epilogue = act if (norm is None and act in (None, 'relu', 'leaky_relu', 'sigmoid')) else False

    def forward(self, x):
        if self._epilogue is not False and epilogue_ok(x):
            return conv_bias_act(self.conv_fn, x, self._epilogue)
        return self.act_fn(self.norm_fn(self.conv_fn(x)))


def epilogue_ok(x):
    """The fused epilogue runs on the product path for GPU tensors.  Under autocast the convolution
    yields bf16 / fp16: ``conv_bias_act`` then widens its output to fp32 first (the reference's
    activations stay in the low precision there; this path is at least as precise)."""
    return runtime.fused() and x.is_cuda and runtime.atomics_ok('bias_act')


_PW_DX_GEMM = os.environ.get('CAMLI_PW_DX', 'gemm') != 'lib'
_PW_FWD_BMM = os.environ.get('CAMLI_PW_FWD', 'bmm') == 'bmm'


def _is_nhwc(x):
    return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


class _PointwiseConv(torch.autograd.Function):
    """1x1, stride-1 convolution without bias.  The forward stays on the library (a GEMM, no layout
    change); the data gradient is a batched GEMM (see backward); the WEIGHT gradient is taken as a batched GEMM over the positions
    (dW = sum_b gy_b x_b^T), because the library's weight-gradient path for these shapes is an NHWC
    implicit-GEMM kernel wrapped in three layout transposes and a zero-fill (measured, B8 128->128 at
    68x120: 137 us for dx+dW against 55 + 39 us)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.param = runtime.deferral_target(w)      # identity of the Parameter, for deferral
        nd = x.dim() - 2
        ctx.conv_args = ([1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1)
        ctx.nhwc_in = _is_nhwc(x)
        if ctx.nhwc_in:
            # channels-last input (the ResNet trunk's output), channel-first output: y_b = W x_b with x_b read as the
            # [P, C] matrix it is in memory -- the layout change rides on the GEMM's transposed operand instead of a
            # 0.5 ms transposing copy of the [16, 512, 68, 120] map (and another one for its gradient)
            w2 = w.flatten(1).unsqueeze(0).expand(x.shape[0], -1, -1)
            y = torch.empty((x.shape[0], w.shape[0]) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
            torch.bmm(w2, x.flatten(2), out=y.flatten(2))       # y itself is returned: the epilogue works in place on it
            return y
        if _PW_FWD_BMM and x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous():
            # the forward as a strided-batched GEMM through at::cuda::blas (y_b = W x_b), where the shipped TunableOp table
            # picks the solution -- the library's 1x1 convolution path calls rocBLAS itself, out of TunableOp's reach
            w2 = w.flatten(1).unsqueeze(0).expand(x.shape[0], -1, -1)
            y = torch.empty((x.shape[0], w.shape[0]) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
            torch.bmm(w2, x.flatten(2), out=y.flatten(2))
            return y
        return torch.ops.aten.convolution(x, w, None, *ctx.conv_args)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        if gy.dtype != w.dtype or x.dtype != gy.dtype:
            # the forward ran under autocast (fp32 operands, reduced-precision output): the adjoints run in the
            # forward's compute type, as the library's own autocast backward does
            x, w = x.to(gy.dtype), w.to(gy.dtype)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if ctx.nhwc_in:
                # dx in the layout x came in: dx_b^T [P, C] = gy_b^T W, a channels-last tensor without a copy
                gx = torch.matmul(gy.flatten(2).transpose(1, 2), w.flatten(1))                    # [B, P, C]
                gx = gx.view(x.shape[0], *x.shape[2:], x.shape[1]).permute(0, 3, 1, 2)
            elif _PW_DX_GEMM:
                # dx_b = W^T gy_b: one strided-batched GEMM.  The library's data-gradient entry costs ~340 us of HOST
                # time per call here (solution lookup on every call + a zero-fill launch, 304 calls per step =
                # 100 ms of the step's enqueue time) for the same GEMM on the device.
                gy3 = gy.flatten(2)
                wt = w.flatten(1).t().unsqueeze(0).expand(gy3.shape[0], -1, -1)      # batch stride 0: no copy
                gx = torch.bmm(wt, gy3).view(x.shape)
            else:
                gx = torch.ops.aten.convolution_backward(gy, x, w, None, *ctx.conv_args, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gy3, xt3 = gy.flatten(2), x.flatten(2).transpose(1, 2)
            if gy3.dtype != torch.float32:       # fp32 accumulation of the weight gradient
                gy3, xt3 = gy3.float(), xt3.float()
            if ctx.param is not None:
                # one [B,Co,Ci] accumulator per parameter and pass, GEMM with beta = 1; summed over the batch and
                # moved into .grad once, at the end of backward()
                acc = runtime.PARAM_GRADS.slot(ctx.param, lambda: torch.zeros(gy3.shape[0], gy3.shape[1], xt3.shape[2],
                                                                            dtype=torch.float32, device=gy.device), True)
                acc.baddbmm_(gy3, xt3)
            else:
                gw = torch.bmm(gy3, xt3).sum(0).view_as(w).to(ctx.saved_tensors[1].dtype)
        return gx, gw


def _is_pointwise(conv):
    return (all(k == 1 for k in conv.kernel_size) and all(s == 1 for s in conv.stride)
            and all(p == 0 for p in conv.padding) and all(d == 1 for d in conv.dilation) and conv.groups == 1
            and not isinstance(conv.padding, str))


# CAMLI_CONV_CL=0: the GRU's half-step convolutions on NCHW tensors like every other one (the library transposes around its
# NHWC kernels); default: on explicitly channels-last operands (_CatConvCL).  The same treatment of the 3x3 call sites that
# gain in isolation (conv_c2 -123 us, flow-head / mask-head conv1 -9 .. -90 us fwd+bwd, tools/conv_cl_probe.py) was measured
# in the step and LOSES there: 221.7 against 215.9 ms (alternating runs on one box), so those stay with the library.
_CONV_CL = os.environ.get('CAMLI_CONV_CL', '1') != '0'
_CONVCL = os.environ.get('CAMLI_CONVCL', '1') != '0'


def conv_bias_act(conv, x, act, leave_bias=False):
    """``act(conv(x))`` with the bias add, the activation and (backward) the bias-gradient reduction
    fused into one pass over the convolution output (camli_bias_act_fwd/bwd).  ``leave_bias`` (act None only): return
    the bias-free convolution -- the caller's next kernel adds ``conv.bias`` itself (ConvexUpsampler2D)."""
    from ..csrc import fused
    assert not leave_bias or act is None
    if act is None and not leave_bias and fused.conv3x3_co2_supported(conv, x):
        # a flow head's last convolution (wide map -> 2 channels): own HBM-bound kernels, bias included
        return fused.conv3x3_co2(x, conv.weight, conv.bias)
    if not leave_bias and fused.wino_epilogue_ok(conv, x, act):
        # r6: bias + activation on the Winograd output transform, the ReLU adjoint on the backward's input transforms
        return fused.wino_conv_cat(x, conv, act)
    if _is_pointwise(conv) and x.dtype == torch.float32 and (x.is_contiguous() or _is_nhwc(x)):
        y = _PointwiseConv.apply(x, conv.weight)
    elif fused.wino_supported(conv, x):
        # r6: the update block's wide 3x3 convolutions as Winograd F(2x2,3x3) on the fp32 matrix cores (csrc/hip/winograd.hip)
        y = fused.conv3x3_wino(x, conv.weight)
    else:
        y = conv._conv_forward(x, conv.weight, None)
    if y.dtype != torch.float32:      # autocast: a fresh fp32 copy the epilogue may overwrite in place
        y = y.float()
    if leave_bias:
        return y
    if conv.bias is None:
        return y if act is None else fused.bias_act(y, torch.zeros(y.shape[1], device=y.device), act)
    return fused.bias_act(y, conv.bias, act)


class _CatConvCL(torch.autograd.Function):
    """``conv2d(cat(parts, 1), w, padding=padding)`` (stride 1, no bias) on explicitly channels-last operands.

    The library runs these shapes (1x5 / 5x1 / 7x7, wide maps) on NHWC implicit-GEMM kernels and, given NCHW tensors,
    wraps every call in layout transposes of its own: x and y in the forward, gy and gx in the data gradient, x AND gy
    again in the weight gradient -- six volume-sized passes per layer and step.  Here the concatenation writes straight
    into a channels-last buffer (it was a copy anyway), that buffer is kept for the weight gradient, and the output
    gradient is transposed once for both adjoints: three passes less per layer."""

    @staticmethod
    def forward(ctx, w, padding, *parts):
        b, _, hh, ww = parts[0].shape
        widths = [p.shape[1] for p in parts]
        hip = runtime.fused() and w.is_cuda
        if hip:
            from ..csrc import fused
        x_cl = torch.empty((b, sum(widths), hh, ww), dtype=w.dtype, device=w.device, memory_format=torch.channels_last)
        c0 = 0
        for p, c in zip(parts, widths):
            if hip:
                fused.nchw_into_channels_last(p, x_cl, c0)
            else:
                x_cl[:, c0:c0 + c].copy_(p)
            c0 += c
        ctx.conv_args = ([1, 1], list(padding), [1, 1], False, [0, 0], 1)
        ctx.widths, ctx.hip = widths, hip
        # r5: the contraction itself on this repo's channels-last kernels (csrc/hip/convcl.hip: forward 0.76-0.80 of the fp32
        # MFMA peak where the library's NHWC implicit GEMM runs at 0.66-0.70; both adjoints); CAMLI_CONVCL=0 -> the library
        # (the own kernels write an H x W output with zeros outside the image: "same" padding only -- any other padding
        # changes the output size, which the library path below handles)
        ctx.own = (hip and _CONVCL and w.dtype == torch.float32 and fused.convcl_supported(w.shape[1], w.shape[0])
                   and w.shape[2] * w.shape[3] <= 32 and 2 * padding[0] == w.shape[2] - 1 and 2 * padding[1] == w.shape[3] - 1)
        if ctx.own:
            wp, ctx.wpt = fused.convcl_pack(w)
            ctx.taps = (w.shape[2], w.shape[3], padding[0], padding[1])
            y = fused.convcl([x_cl.permute(0, 2, 3, 1)], wp, fused.convcl_taps(*ctx.taps))
            ctx.save_for_backward(x_cl, w)
            return fused.channels_last_to_nchw(y.permute(0, 3, 1, 2), 0, y.shape[3])
        w_cl = w.contiguous(memory_format=torch.channels_last)
        y_cl = torch.ops.aten.convolution(x_cl, w_cl, None, *ctx.conv_args)
        ctx.save_for_backward(x_cl, w_cl)
        if hip:
            return fused.channels_last_to_nchw(y_cl, 0, y_cl.shape[1])
        return y_cl.contiguous()

    @staticmethod
    def backward(ctx, gy):
        x_cl, w_cl = ctx.saved_tensors
        if ctx.hip:
            from ..csrc import fused
            gy_cl = torch.empty(gy.shape, dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
            fused.nchw_into_channels_last(gy, gy_cl, 0)
        else:
            gy_cl = gy.contiguous(memory_format=torch.channels_last)
        need_x = any(ctx.needs_input_grad[2:])
        if ctx.own:
            gy_n, x_n = gy_cl.permute(0, 2, 3, 1), x_cl.permute(0, 2, 3, 1)
            gx_cl = gw_cl = None
            if need_x:
                gx_cl = fused.convcl([gy_n], ctx.wpt, fused.convcl_taps(*ctx.taps, negate=True)).permute(0, 3, 1, 2)
            if ctx.needs_input_grad[0]:
                gw_cl = fused.convcl_wrw([x_n], gy_n, fused.convcl_taps(*ctx.taps), ctx.taps[:2])
        else:
            gx_cl, gw_cl, _ = torch.ops.aten.convolution_backward(gy_cl, x_cl, w_cl, None, *ctx.conv_args,
                                                                  [need_x, ctx.needs_input_grad[0], False])
        gparts, c0 = [], 0
        for i, c in enumerate(ctx.widths):
            if not ctx.needs_input_grad[2 + i]:
                gparts.append(None)
            elif ctx.hip:
                gparts.append(fused.channels_last_to_nchw(gx_cl, c0, c))
            else:
                gparts.append(gx_cl[:, c0:c0 + c].contiguous())
            c0 += c
        return (gw_cl.contiguous() if gw_cl is not None else None, None, *gparts)


def cat_conv_cl(parts, w, padding):
    return _CatConvCL.apply(w, tuple(padding), *parts)


def flow_conv(conv, x):
    """A bare ``nn.Conv2d`` that ends a flow head (wide map -> 2 channels, pwc_core.py / raft_core.py:169-181): on the
    product path the two-channel 3x3 kernels of csrc/hip/smallconv.hip, otherwise the module itself."""
    if runtime.fused() and x.is_cuda:
        from ..csrc import fused
        if fused.conv3x3_co2_supported(conv, x):
            return fused.conv3x3_co2(x, conv.weight, conv.bias)
    return conv(x)


class Conv1dNormRelu(_ConvNormAct):
    dims = 1


class Conv2dNormRelu(_ConvNormAct):
    dims = 2


class _MLP(nn.Module):
    layer = Conv1dNormRelu

    def __init__(self, in_channels, mlp_channels, norm=None, act='leaky_relu'):
        super().__init__()
        assert isinstance(in_channels, int) and isinstance(mlp_channels, list)
        widths = [in_channels] + mlp_channels
        self.convs = nn.ModuleList(self.layer(a, b, norm=norm, act=act) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return x


class MLP1d(_MLP):
    layer = Conv1dNormRelu


class MLP2d(_MLP):
    layer = Conv2dNormRelu

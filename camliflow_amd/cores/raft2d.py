"""2-D RAFT branch (counterpart of models/raft_core.py): image encoder, all-pairs cost-volume
pyramid with radius-4 lookup, conv-GRU update block, flow head and convex upsampler.
"""
import math
import os

import torch
import torch.nn as nn
from torch.nn.functional import avg_pool2d, grid_sample

from . import runtime
from .blocks import _CONV_CL, Conv2dNormRelu, cat_conv_cl, conv_bias_act, epilogue_ok
from .geometry import convex_upsample, mesh_grid
from .resnet import ResNetTrunk


class Encoder2D(ResNetTrunk):
    """ResNet-50 stem + stages 1-2 (stride 8, 512 ch) followed by a 1x1 ``align`` conv to 128 ch
    (raft_core.py:10-38).  BatchNorm layers of the trunk always run in eval mode (norm_eval=True)."""

    def __init__(self, depth=50, pretrained=None):
        super().__init__(depth=depth, num_stages=2, strides=(1, 2), norm_eval=True)
        self.align = Conv2dNormRelu(self.feat_dim, 128)
        self.init_weights()
        if pretrained is not None:
            import os
            if os.path.exists(pretrained):
                state = torch.load(pretrained, map_location='cpu')
                self.load_state_dict(state.get('state_dict', state), strict=False)

    def forward(self, x):
        # on the product path ``align`` (1x1, fp32) takes the trunk's channels-last map directly
        direct = (runtime.fused() and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
                  and self.align._epilogue is not False and os.environ.get('CAMLI_ALIGN_NHWC', '1') == '1')
        return self.align(super().forward(x, keep_channels_last=direct)[0])


def _window_offsets(radius, device):
    return torch.linspace(-radius, radius, 2 * radius + 1, device=device)


class Correlation2D(nn.Module):
    """All-pairs correlation volume + pooled pyramid, looked up in a (2r+1)^2 window per level
    (raft_core.py:41-107).  The pyramid is module state between ``build`` and the lookups, exactly
    like the reference (not re-entrant)."""

    def __init__(self, num_levels=4, radius=4):
        super().__init__()
        self.num_levels = num_levels
        self.radius = radius
        self.fnet_aligner = nn.Conv2d(128, 256, kernel_size=1)
        self.cost_volume_pyramid = None

    def build_cost_volume_pyramid(self, fmap1, fmap2):
        fmap1 = self.fnet_aligner(fmap1.float())
        fmap2 = self.fnet_aligner(fmap2.float())
        bs, dim, h, w = fmap1.shape
        if runtime.fused():
            from ..csrc import fused
            self.cost_volume_pyramid = fused.allpairs_pyramid(fmap1, fmap2, self.num_levels)
            return
        volume = torch.matmul(fmap1.view(bs, dim, h * w).transpose(1, 2), fmap2.view(bs, dim, h * w))
        volume = (volume / torch.sqrt(torch.tensor(dim))).reshape(bs * h * w, 1, h, w)
        self.cost_volume_pyramid = [volume]
        for _ in range(self.num_levels - 1):
            volume = avg_pool2d(volume, 2, stride=2)
            self.cost_volume_pyramid.append(volume)

    def release(self):
        """Let go of the pass's pyramid once its last lookup is enqueued (the lookup nodes keep what their backward needs).
        The reference leaves it in the module until the next build overwrites it; here that would keep the PREVIOUS step's
        autograd graph -- through the pyramid's token, back to the encoders' parameters -- alive while the next forward
        pass runs, and torch then reuses that step's AccumulateGrad nodes with the streams they were created on."""
        self.cost_volume_pyramid = None

    def forward(self, coords):
        """coords [B,2,h,w] (x,y) at 1/8 resolution -> [B, levels*(2r+1)^2, h, w].
        Channel l*81 + i*9 + j samples level l at (x/2^l + d[i], y/2^l + d[j]) -- the transposed
        window of the reference (raft_core.py:79-85)."""
        if runtime.fused():
            from ..csrc import fused
            return fused.allpairs_lookup(self.cost_volume_pyramid, coords.float(), self.radius)
        coords = coords.permute(0, 2, 3, 1).float()
        bs, h, w, _ = coords.shape
        r = self.radius
        d = _window_offsets(r, coords.device)
        delta = torch.stack(torch.meshgrid(d, d, indexing='ij'), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
        out = []
        for lvl, volume in enumerate(self.cost_volume_pyramid):
            centre = coords.reshape(bs * h * w, 1, 1, 2) / 2 ** lvl
            sampled = self.bilinear_sampler(volume, centre + delta)
            out.append(sampled.view(bs, h, w, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def bilinear_sampler(feat, coords):
        """grid_sample in pixel units, align_corners=True, zeros padding (raft_core.py:96-107)."""
        h, w = feat.shape[-2:]
        xgrid, ygrid = coords.split([1, 1], dim=-1)
        grid = torch.cat([2 * xgrid / (w - 1) - 1, 2 * ygrid / (h - 1) - 1], dim=-1)
        return grid_sample(feat, grid, align_corners=True)


class GRU2D(nn.Module):
    """Separable (1x5 then 5x1) convolutional GRU (raft_core.py:110-139)."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for suffix, ksize, padding in (('1', (1, 5), (0, 2)), ('2', (5, 1), (2, 0))):
            for gate in 'zrq':
                setattr(self, 'conv%s%s' % (gate, suffix),
                        nn.Conv2d(hidden_dim + input_dim, hidden_dim, ksize, padding=padding))

    def _half_step(self, h, x, convz, convr, convq):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(convz(hx))
        r = torch.sigmoid(convr(hx))
        q = torch.tanh(convq(torch.cat([r * h, x], dim=1)))
        return (1 - z) * h + z * q

    def forward(self, h, x):
        h = self._half_step(h, x, self.convz1, self.convr1, self.convq1)
        h = self._half_step(h, x, self.convz2, self.convr2, self.convq2)
        return torch.nan_to_num(h)

    # ---- iteration-invariant hoisting (used by the cores under the 'hip' backend) -------------
    # x = cat([context, motion]); the context half never changes across GRU iterations
    # (raft_core.py:236-259: `x` is computed once from cnet).  A convolution is linear in its input
    # channels, so conv(cat[h, context, motion]) = conv_{h,motion}(cat[h, motion]) + conv_context(context):
    # the context term (+ bias) is evaluated ONCE per pass for all six gates, and z / r share one
    # convolution.  One third of the GRU's convolution work (forward, data-gradient and
    # weight-gradient) disappears; values agree with the literal form up to fp32 summation order.
    def prepare(self, context):
        """context [B,128,h,w] -> per-pass state for ``step``."""
        hd = self.convz1.weight.shape[0]
        cd = context.shape[1]
        state = {}
        for suffix, padding in (('1', (0, 2)), ('2', (2, 0))):
            gates = [getattr(self, 'conv%s%s' % (g, suffix)) for g in 'zrq']
            w_ctx = torch.cat([g.weight[:, hd:hd + cd] for g in gates], dim=0)
            bias = torch.cat([g.bias for g in gates], dim=0)
            if torch.is_autocast_enabled() and runtime.own_kernels_allowed():
                with torch.autocast('cuda', enabled=False):        # the hoisted context terms feed the fp32 own kernels
                    ctx = torch.nn.functional.conv2d(context.float(), w_ctx.float(), bias.float(), padding=padding)
            else:
                ctx = torch.nn.functional.conv2d(context, w_ctx, bias, padding=padding)
            keep = [torch.cat([g.weight[:, :hd], g.weight[:, hd + cd:]], dim=1) for g in gates]
            # contiguous once per pass: the gate kernels would otherwise copy these channel slices every iteration
            state[suffix] = (torch.cat(keep[:2], dim=0), keep[2], ctx[:, :2 * hd].contiguous(),
                             ctx[:, 2 * hd:].contiguous(), padding)
            if (runtime.fused() and context.is_cuda and ctx.dtype == torch.float32 and runtime.own_kernels_allowed()
                    and os.environ.get('CAMLI_GRU_CL', '1') != '0'):
                # r5 (default): the whole update as ONE node on channels-last tensors, convolutions + gate arithmetic on this
                # repo's matrix-core kernels (csrc/hip/convcl.hip, fused._GRU2DStepCL); the hoisted context terms are laid out
                # NHWC once per pass
                state['cl' + suffix] = (ctx[:, :2 * hd].permute(0, 2, 3, 1).contiguous(), ctx[:, 2 * hd:].permute(0, 2, 3, 1).contiguous())
        if 'cl1' in state and 'cl2' in state:
            from ..csrc import fused
            state['hub'] = fused.GRU2DPass((state['1'][0], state['1'][1], state['2'][0], state['2'][1]),
                                           (state['cl1'][0], state['cl1'][1], state['cl2'][0], state['cl2'][1]))
        return state

    def step(self, h, motion, state):
        """One GRU update with the hoisted context terms; the elementwise halves run as two fused
        kernels each way (camli_gru_gates / camli_gru_blend) when the channel plane allows 16-byte
        accesses, otherwise as plain torch ops."""
        from ..csrc import fused
        conv2d = torch.nn.functional.conv2d
        hd = h.shape[1]
        if 'hub' in state and runtime.own_kernels_allowed() and fused.gru2d_step_supported(h, motion, state['1'][0]):
            return fused.gru2d_step_cl(h, motion, state['hub'])
        fusable = h.is_cuda and (hd * h.shape[2] * h.shape[3]) % 4 == 0
        for suffix in ('1', '2'):
            w_zr, w_q, ctx_zr, ctx_q, padding = state[suffix]
            # CAMLI_CONV_CL (default on): the two convolutions of a half step on explicitly channels-last operands
            # (blocks._CatConvCL: the cat writes the channels-last input, kept for the weight gradient)
            # (channel counts in multiples of 16 only: on an 8 + 16-channel NHWC input the library's kernels read past the end
            # of the tensor -- "Memory access fault by GPU" whenever the allocation ends a segment, 2 of 3 full test runs late
            # in round 5; the models' GRUs carry 128-channel maps)
            cl = _CONV_CL and fusable and runtime.fused() and h.dtype == torch.float32 and motion.dtype == torch.float32 \
                and not torch.is_autocast_enabled() and hd % 16 == 0 and motion.shape[1] % 16 == 0
            pre_zr = cat_conv_cl([h, motion], w_zr, padding) if cl else conv2d(torch.cat([h, motion], dim=1), w_zr, None, padding=padding)
            if not fusable and h.is_cuda:
                runtime.fallback('GRU2D', 'hidden plane is not a multiple of 4 elements')
            if fusable:
                z, rh = fused.gru_gates(pre_zr, ctx_zr, h)
                # the GRU's closing nan_to_num (raft_core.py:138) rides on the second half-step's blend kernel
                pre_q = cat_conv_cl([rh, motion], w_q, padding) if cl else conv2d(torch.cat([rh, motion], dim=1), w_q, None, padding=padding)
                h = fused.gru_blend(pre_q, ctx_q, z, h, nan_to_num=(suffix == '2'))
            else:
                zr = torch.sigmoid(pre_zr + ctx_zr)
                z, r = zr[:, :hd], zr[:, hd:]
                q = torch.tanh(conv2d(torch.cat([r * h, motion], dim=1), w_q, None, padding=padding) + ctx_q)
                h = (1 - z) * h + z * q
        return h if fusable else torch.nan_to_num(h)


def _conv(cin, cout, ksize):
    return nn.Conv2d(cin, cout, kernel_size=ksize, padding=ksize // 2)


class MotionEncoder2D(nn.Module):
    """(flow, correlation window) -> 126 motion channels + the flow itself (raft_core.py:142-166)."""

    def __init__(self, corr_levels, corr_radius):
        super().__init__()
        corr_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.conv_c1 = _conv(corr_planes, 256, 1)
        self.conv_c2 = _conv(256, 192, 3)
        self.conv_f1 = _conv(2, 128, 7)
        self.conv_f2 = _conv(128, 64, 3)
        self.conv = _conv(64 + 192, 128 - 2, 3)
        self.relu = nn.ReLU(inplace=True)

    def begin(self, flow):
        """Issue the flow branch (conv_f1 7x7 2->128, conv_f2 3x3) on the auxiliary stream as soon as the flow estimate
        exists -- it does not need the correlation features, and a 2-channel convolution leaves most of the chip to the
        lookup / fusion / conv_c* kernels issued meanwhile.  Returns a handle for ``forward(..., flow_branch=handle)``;
        None where the fused epilogue does not apply (the caller then runs the plain forward)."""
        if not epilogue_ok(flow):
            return None
        branch = runtime.Branch(flow, slot=0)
        raw = self._cat_free(flow)
        with branch:
            f = conv_bias_act(self.conv_f1, flow, 'relu')
            # cat-free form: the second convolution leaves its bias + ReLU to the kernel that writes the concatenation
            f = conv_bias_act(self.conv_f2, f, None, leave_bias=True) if raw else conv_bias_act(self.conv_f2, f, 'relu')
        return branch, f, raw

    @staticmethod
    def _cat_free(t):
        """bias + activation epilogues write straight into the concatenated tensors (fused.bias_act_cat): planes of 4k
        elements, fp32 outside autocast; CAMLI_BIAS_CAT=0 restores epilogue + torch.cat."""
        return (os.environ.get('CAMLI_BIAS_CAT', '1') == '1' and (t.shape[2] * t.shape[3]) % 4 == 0
                and runtime.own_kernels_allowed())

    def forward(self, flow, corr, flow_branch=None):
        if epilogue_ok(corr) and self._cat_free(flow) and (flow_branch is None or (len(flow_branch) > 2 and flow_branch[2])):
            from ..csrc import fused
            c1 = conv_bias_act(self.conv_c1, corr, 'relu')
            wino = fused.wino_epilogue_ok(self.conv_c2, c1, 'relu')
            c_raw = None if wino else conv_bias_act(self.conv_c2, c1, None, leave_bias=True)
            if flow_branch is not None:
                branch, f_raw = flow_branch[0], flow_branch[1]
                branch.join(f_raw)
            else:
                f_raw = conv_bias_act(self.conv_f2, conv_bias_act(self.conv_f1, flow, 'relu'), None, leave_bias=True)
            if wino:
                # r6: conv_c2 / conv as Winograd convolutions whose output transform writes bias + ReLU (+ nan_to_num) straight
                # into the concatenations (fused.wino_conv_cat): no pass over their outputs, forward or backward
                x = fused.wino_conv_cat(c1, self.conv_c2, 'relu', others=[(f_raw, self.conv_f2.bias, 'relu')])
                return fused.wino_conv_cat(x, self.conv, 'relu_nan_to_num', tail=flow)
            x = fused.bias_act_cat([(c_raw, self.conv_c2.bias, 'relu'), (f_raw, self.conv_f2.bias, 'relu')])
            joint_raw = conv_bias_act(self.conv, x, None, leave_bias=True)
            # relu + nan_to_num (bias_act code 5) and the flow channels appended, in the pass that writes the motion features
            return fused.bias_act_cat([(joint_raw, self.conv.bias, 'relu_nan_to_num')], tail=flow)
        if epilogue_ok(corr):
            c = conv_bias_act(self.conv_c2, conv_bias_act(self.conv_c1, corr, 'relu'), 'relu')
            if flow_branch is not None:
                branch, f = flow_branch[0], flow_branch[1]
                branch.join(f)
                if len(flow_branch) > 2 and flow_branch[2]:      # a raw handle (bias and ReLU left to the caller), the state changed in between
                    f = self.relu(f + self.conv_f2.bias.view(1, -1, 1, 1))
            else:
                f = conv_bias_act(self.conv_f2, conv_bias_act(self.conv_f1, flow, 'relu'), 'relu')
            x = torch.cat([c, f], dim=1)
            if (x.shape[2] * x.shape[3]) % 4 == 0:      # relu + nan_to_num in the epilogue pass (bias_act code 5)
                joint = conv_bias_act(self.conv, x, 'relu_nan_to_num')
            else:
                joint = torch.nan_to_num(conv_bias_act(self.conv, x, 'relu'))
            return torch.cat([joint, flow], dim=1)
        c = self.relu(self.conv_c1(corr))
        c = self.relu(self.conv_c2(c))
        if flow_branch is not None:          # issued by begin(): same values, join before use
            flow_branch[0].join(flow_branch[1])
            f = flow_branch[1]
            if len(flow_branch) > 2 and flow_branch[2]:      # begin() left conv_f2's bias + ReLU to the epilogue branch: finish it here
                f = self.relu(f + self.conv_f2.bias.view(1, -1, 1, 1))
        else:
            f = self.relu(self.conv_f1(flow))
            f = self.relu(self.conv_f2(f))
        joint = torch.nan_to_num(self.relu(self.conv(torch.cat([c, f], dim=1))))
        return torch.cat([joint, flow], dim=1)


class FlowHead2D(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = _conv(input_dim, hidden_dim, 3)
        self.conv2 = _conv(hidden_dim, 2, 3)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if epilogue_ok(x):
            delta = conv_bias_act(self.conv2, conv_bias_act(self.conv1, x, 'relu'), None)
        else:
            delta = self.conv2(self.relu(self.conv1(x)))
        return torch.nan_to_num(delta.float())


class ConvexUpsampler2D(nn.Module):
    """8x convex up-sampling; the mask head's output is scaled by 0.25 "to balance gradients"
    (raft_core.py:183-197)."""

    def __init__(self, input_dim):
        super().__init__()
        self.mask = nn.Sequential(_conv(input_dim, 256, 3), nn.ReLU(inplace=True), _conv(256, 64 * 9, 1))

    def begin(self, h):
        """Issue the mask head (3x3 + 1x1 convolution of the hidden state) on the auxiliary stream; the flow head, which
        reads the same hidden state, follows on the current stream and the two chains meet in ``finish``.  Returns a
        handle, or None where the fused path does not apply."""
        if not (epilogue_ok(h) and runtime.atomics_ok('convex_upsample')):
            return None
        branch = runtime.Branch(h, slot=1)
        with branch:
            raw = conv_bias_act(self.mask[2], conv_bias_act(self.mask[0], h.float(), 'relu'), None, leave_bias=True)
        return branch, raw

    def finish(self, handle, h, flow, out_rows=None):
        if handle is None:
            return self.forward(h, flow, out_rows)
        branch, raw = handle
        branch.join(raw)
        return convex_upsample(flow, raw, mask_scale=0.25, mask_bias=self.mask[2].bias, out_rows=out_rows)

    def forward(self, h, flow, out_rows=None):
        """``out_rows``: produce the first out_rows of the 8h fine rows only -- the caller's un-padding of a bottom-padded
        image done by the up-sampling kernel (no slice + copy of every prediction, forward and backward)."""
        if epilogue_ok(h) and runtime.atomics_ok('convex_upsample'):
            # the last convolution's bias is added inside the up-sampling kernel: one pass less over [B,576,h,w]
            raw = conv_bias_act(self.mask[2], conv_bias_act(self.mask[0], h.float(), 'relu'), None, leave_bias=True)
            return convex_upsample(flow, raw, mask_scale=0.25, mask_bias=self.mask[2].bias, out_rows=out_rows)
        if epilogue_ok(h):
            mask = conv_bias_act(self.mask[2], conv_bias_act(self.mask[0], h.float(), 'relu'), None)
        else:
            mask = self.mask(h.float())
        return convex_upsample(flow, mask, mask_scale=0.25, out_rows=out_rows)


class RAFTCore(nn.Module):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.hidden_dim = 128
        self.context_dim = 128
        self.corr_levels = 4
        self.corr_radius = 4
        self.fnet = Encoder2D(cfgs.backbone.depth, cfgs.backbone.pretrained)
        self.cnet = Encoder2D(cfgs.backbone.depth, cfgs.backbone.pretrained)
        self.cnet_aligner = nn.Conv2d(128, 256, kernel_size=1)
        self.correlation = Correlation2D(self.corr_levels, self.corr_radius)
        self.motion_encoder = MotionEncoder2D(self.corr_levels, self.corr_radius)
        self.gru = GRU2D(hidden_dim=self.hidden_dim, input_dim=self.hidden_dim + 128)
        self.flow_head = FlowHead2D(self.hidden_dim)
        self.convex_upsampler = ConvexUpsampler2D(self.hidden_dim)

    def forward(self, image1, image2):
        """image-only RAFT (raft_core.py:226-270): encode, build the all-pairs pyramid, iterate."""
        self.correlation.build_cost_volume_pyramid(self.fnet(image1), self.fnet(image2))
        state = self.cnet_aligner(self.cnet(image1))
        hidden, context = torch.tanh(state[:, :self.hidden_dim]), torch.relu(state[:, self.hidden_dim:])

        bs, _, image_h, image_w = image1.shape
        grid = mesh_grid(bs, image_h // 8, image_w // 8, device=image1.device)
        flow = torch.zeros_like(grid)
        n_iters = self.cfgs.n_iters_train if self.training else self.cfgs.n_iters_eval
        hoisted = self.gru.prepare(context) if runtime.fused() else None

        predictions = []
        for _ in range(n_iters):
            flow = flow.detach()
            motion = self.motion_encoder(flow, self.correlation(grid + flow))
            if hoisted is not None:
                hidden = self.gru.step(hidden, motion, hoisted)
            else:
                hidden = self.gru(hidden, torch.cat([context, motion], dim=1))
            flow = flow + self.flow_head(hidden)
            predictions.append(self.convex_upsampler(hidden, flow))
        self.correlation.release()
        return predictions

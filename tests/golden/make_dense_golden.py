"""Golden vectors for oracle/dense.py from the REFERENCE's own modules, run unchanged with autograd on seeded inputs:

  dense_cost_mlp        models/camliraft_l_core.py Correlation3D: cost_mlp (MLP2d 4 -> 32 -> 32, relu) + sum over the k
                        neighbours (:96-98) for four levels, concatenated as forward() does (:86-93)
  dense_flow_head       models/raft_core.py FlowHead2D (:169-182): input / output of its two-channel conv2, gradients
  dense_allpairs_{even,odd}  models/raft_core.py Correlation2D.build_cost_volume_pyramid (:52-68): the aligned feature
                        maps, the four levels, gradients back to the aligned maps
  dense_point_volume    models/camliraft_l_core.py Correlation3D.build_cost_volume_pyramid (:51-60): features, the
                        neighbour tables it computed, the four levels, gradients back to the features; and
                        models/utils.py batch_indexing(layout='channel_last') (:85-104), rank-3 and rank-2 data
  dense_gru2d           models/raft_core.py GRU2D (:110-140): weights, inputs (h, x = [context | motion]), the input and
                        output of each of its six 1x5 / 5x1 convolutions, the new hidden state
  dense_gru2d_wide      the same module at the product's widths (hidden 128, x = 128 context + 128 motion channels), two updates
                        with autograd: inputs, output, input gradients, fingerprints of the weight gradients (name-hashed weights)
  dense_update_block    models/raft_core.py MotionEncoder2D (:142-166), FlowHead2D (:169-181) and the mask head of ConvexUpsampler2D
                        (:184-190) at the product's widths with autograd: inputs, outputs, input gradients, fingerprints of every
                        parameter gradient (name-hashed weights) -- the module-level pin of the 3x3 convolutions the product
                        runs as Winograd F(2x2,3x3) (csrc/hip/winograd.hip)
  dense_resnet_glue     the stem max pooling and the bottleneck epilogue of the ResNet trunk the reference instantiates
                        through mmdet (README.md:78-79, models/raft_core.py:10-38; mmdet itself is not under
                        /root/reference -- SURVEY 8c): nn.MaxPool2d(3, 2, 1) and relu(bn-bias + conv + identity) from torch

Run in the build container only:  python tests/golden/make_dense_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import refmodels  # noqa: E402

refmodels.install(native_semantics=True)
from models.camliraft_l_core import Correlation3D  # noqa: E402
from models.raft_core import GRU2D, ConvexUpsampler2D, Correlation2D, FlowHead2D, MotionEncoder2D  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def fill(module, g, scale):
    for p in module.parameters():
        p.data = torch.randn(p.shape, generator=g) * scale


def golden_cost_mlp():
    g = torch.Generator().manual_seed(21)
    b, n, k, levels = 2, 40, 16, 4          # the fused kernel takes point counts that are multiples of 8
    corr = Correlation3D(out_channels=128, k=k)
    fill(corr.cost_mlp, g, 0.4)
    lookups = [torch.randn(b, 4, n, k, generator=g, requires_grad=True) for _ in range(levels)]
    # calc_matching_cost's tail (:96-98) per level, forward()'s concatenation (:93)
    costs = torch.cat([torch.sum(corr.cost_mlp(x), dim=-1) for x in lookups], dim=1)
    gout = torch.randn(costs.shape, generator=g)
    costs.backward(gout)
    convs = [m for m in corr.cost_mlp.modules() if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 2
    save('dense_cost_mlp', lookup=torch.cat([x.detach() for x in lookups], dim=-1), levels=levels,
         w1=convs[0].weight.detach().reshape(32, 4), b1=convs[0].bias.detach(), w2=convs[1].weight.detach().reshape(32, 32),
         b2=convs[1].bias.detach(), out=costs.detach(), gout=gout,
         glookup=torch.cat([x.grad for x in lookups], dim=-1),
         gw1=convs[0].weight.grad.reshape(32, 4), gb1=convs[0].bias.grad, gw2=convs[1].weight.grad.reshape(32, 32),
         gb2=convs[1].bias.grad)


def golden_flow_head():
    g = torch.Generator().manual_seed(22)
    head = FlowHead2D(input_dim=12, hidden_dim=20)
    fill(head, g, 0.2)
    store = {}

    def pre(_m, args):
        args[0].retain_grad()
        store['x'] = args[0]

    def post(_m, _a, out):
        out.retain_grad()
        store['y'] = out
    head.conv2.register_forward_pre_hook(pre)
    head.conv2.register_forward_hook(post)
    x = torch.randn(2, 12, 9, 11, generator=g)
    out = head(x)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    save('dense_flow_head', x=store['x'].detach(), w=head.conv2.weight.detach(), b=head.conv2.bias.detach(), y=store['y'].detach(),
         gy=store['y'].grad, gx=store['x'].grad, gw=head.conv2.weight.grad, gb=head.conv2.bias.grad, head_out=out.detach())


def golden_allpairs(tag, hh, ww):
    g = torch.Generator().manual_seed(23 + hh)
    corr = Correlation2D(num_levels=4, radius=4)
    fill(corr, g, 0.1)
    store = []

    def post(_m, _a, out):
        out.retain_grad()
        store.append(out)
    corr.fnet_aligner.register_forward_hook(post)
    fmap1 = torch.randn(2, 128, hh, ww, generator=g)
    fmap2 = torch.randn(2, 128, hh, ww, generator=g)
    corr.build_cost_volume_pyramid(fmap1, fmap2)
    pyr = corr.cost_volume_pyramid
    gpyr = [torch.randn(p.shape, generator=g) for p in pyr]
    sum((p * q).sum() for p, q in zip(pyr, gpyr)).backward()
    arrays = {'f1': store[0].detach(), 'f2': store[1].detach(), 'gf1': store[0].grad, 'gf2': store[1].grad}
    for lvl, (p, q) in enumerate(zip(pyr, gpyr)):
        arrays['pyr%d' % lvl] = p.detach().squeeze(1)
        arrays['gpyr%d' % lvl] = q.squeeze(1)
    save('dense_allpairs_' + tag, **arrays)


def golden_point_volume():
    """Correlation3D.build_cost_volume_pyramid (:51-60) on an FPS-like nested target pyramid; the neighbour tables it
    computes internally are recorded by wrapping the module's k_nearest_neighbor; batch_indexing's channel-last form
    (models/utils.py:85-104) on the same data."""
    import models.camliraft_l_core as core
    from models.utils import batch_indexing
    g = torch.Generator().manual_seed(26)
    b, c, n = 2, 24, 72
    sizes = [72, 40, 20, 12]
    corr = Correlation3D(out_channels=128, k=16)
    feat1 = torch.randn(b, c, n, generator=g, requires_grad=True)
    feat2 = torch.randn(b, c, sizes[0], generator=g, requires_grad=True)
    cloud = torch.randn(b, 3, sizes[0], generator=g)
    xyzs2 = [cloud[:, :, :m].contiguous() for m in sizes]
    tables = []
    inner = core.k_nearest_neighbor

    def recording(*args, **kwargs):
        out = inner(*args, **kwargs)
        tables.append(out.detach().clone())
        return out
    core.k_nearest_neighbor = recording
    try:
        corr.build_cost_volume_pyramid(feat1, feat2, xyzs2, k=3)
    finally:
        core.k_nearest_neighbor = inner
    pyr = corr.cost_volume_pyramid
    gpyr = [torch.randn(p.shape, generator=g) for p in pyr]
    sum((p * q).sum() for p, q in zip(pyr, gpyr)).backward()
    arrays = {'f1': feat1.detach(), 'f2': feat2.detach(), 'gf1': feat1.grad, 'gf2': feat2.grad}
    for lvl, (p, q) in enumerate(zip(pyr, gpyr)):
        arrays['pyr%d' % lvl] = p.detach()
        arrays['gpyr%d' % lvl] = q
        arrays['xyz%d' % lvl] = xyzs2[lvl]
    for lvl, t in enumerate(tables):
        arrays['parents%d' % lvl] = t
    # channel-last batch_indexing: rank-3 rows and the rank-2 form calc_matching_cost uses (:70-74)
    rows = torch.randn(b, 30, 7, generator=g, requires_grad=True)
    picks = torch.randint(0, 30, (b, 11, 5), generator=g)
    got = batch_indexing(rows, picks, layout='channel_last')
    grow = torch.randn(got.shape, generator=g)
    got.backward(grow)
    flat = torch.randn(b * 9, 30, generator=g, requires_grad=True)
    fpicks = torch.randint(0, 30, (b * 9, 16), generator=g)
    fgot = batch_indexing(flat, fpicks, layout='channel_last')
    fg = torch.randn(fgot.shape, generator=g)
    fgot.backward(fg)
    arrays.update(cl_data=rows.detach(), cl_idx=picks, cl_out=got.detach(), cl_gout=grow, cl_gdata=rows.grad,
                  cl2_data=flat.detach(), cl2_idx=fpicks, cl2_out=fgot.detach(), cl2_gout=fg, cl2_gdata=flat.grad)
    save('dense_point_volume', **arrays)


def golden_gru2d():
    g = torch.Generator().manual_seed(25)
    hd, cd, md = 16, 8, 24                       # hidden, context and motion channels: x = cat([context, motion])
    gru = GRU2D(hidden_dim=hd, input_dim=cd + md)
    fill(gru, g, 0.15)
    store = {}
    for name in ('convz1', 'convr1', 'convq1', 'convz2', 'convr2', 'convq2'):
        mod = getattr(gru, name)
        mod.register_forward_hook(lambda _m, args, out, name=name: store.update({name + '_in': args[0].detach(), name + '_out': out.detach()}))
    h0 = torch.tanh(torch.randn(2, hd, 11, 13, generator=g))
    x = torch.randn(2, cd + md, 11, 13, generator=g)
    with torch.no_grad():
        out = gru(h0, x)
    arrays = {'h0': h0, 'x': x, 'out': out, 'hidden': hd, 'context': cd}
    for name in ('convz1', 'convr1', 'convq1', 'convz2', 'convr2', 'convq2'):
        arrays[name + '_w'] = getattr(gru, name).weight.detach()
        arrays[name + '_b'] = getattr(gru, name).bias.detach()
    arrays.update(store)
    save('dense_gru2d', **arrays)


def golden_gru2d_wide():
    """The reference's GRU2D at the product's widths (hidden 128, x = [128 context | 128 motion]) with autograd: the shapes
    camli_convcl_gru_gates / _blend take.  Weights are name-hashed (tests/modelutils.hashed_fill_: none are stored); the
    weight gradients are recorded as fingerprints (L2 norm and the projection on a name-seeded random direction)."""
    import zlib
    from modelutils import hashed_fill_
    g = torch.Generator().manual_seed(26)
    gru = hashed_fill_(GRU2D(hidden_dim=128, input_dim=256))
    h0 = torch.tanh(torch.randn(1, 128, 10, 14, generator=g)).requires_grad_()
    x = torch.randn(1, 256, 10, 14, generator=g).requires_grad_()
    out = gru(gru(h0, x), x)                     # two updates, as two GRU iterations share the weights
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    arrays = {'h0': h0.detach(), 'x': x.detach(), 'out': out.detach(), 'gout': gout, 'gh0': h0.grad, 'gx': x.grad}
    for name, p in gru.named_parameters():
        d = torch.randn(p.shape, generator=torch.Generator().manual_seed(zlib.crc32(('dir.' + name).encode())))
        arrays['fp_' + name] = torch.stack([p.grad.double().norm(), (p.grad.double() * d.double()).sum()])
    save('dense_gru2d_wide', **arrays)


def _fingerprints(module, arrays, prefix):
    import zlib
    for name, p in module.named_parameters():
        d = torch.randn(p.shape, generator=torch.Generator().manual_seed(zlib.crc32(('dir.' + prefix + name).encode())))
        arrays['fp_' + prefix + name] = torch.stack([p.grad.double().norm(), (p.grad.double() * d.double()).sum()])


def golden_update_block():
    """The reference's MotionEncoder2D, FlowHead2D and mask head (raft_core.py:142-190) at the product's widths (4 levels x 81
    correlation channels, hidden 128) on a small odd-sized map, with autograd.  Weights name-hashed (modelutils.hashed_fill_),
    parameter gradients as fingerprints (L2 norm, projection on a name-seeded direction)."""
    from modelutils import hashed_fill_
    g = torch.Generator().manual_seed(27)
    enc = hashed_fill_(MotionEncoder2D(4, 4))
    head = hashed_fill_(FlowHead2D(128, 256))
    up = hashed_fill_(ConvexUpsampler2D(128))
    b, hh, ww = 1, 7, 12
    flow = (torch.randn(b, 2, hh, ww, generator=g) * 2).requires_grad_()
    corr = torch.randn(b, 324, hh, ww, generator=g).requires_grad_()
    hidden = torch.tanh(torch.randn(b, 128, hh, ww, generator=g)).requires_grad_()
    motion = enc(flow, corr)
    gmotion = torch.randn(motion.shape, generator=g)
    motion.backward(gmotion)
    delta = head(hidden)
    mask = up.mask(hidden)
    gdelta, gmask = torch.randn(delta.shape, generator=g), torch.randn(mask.shape, generator=g) * 0.1
    (delta * gdelta).sum().add((mask * gmask).sum()).backward()
    arrays = {'flow': flow.detach(), 'corr': corr.detach(), 'hidden': hidden.detach(), 'motion': motion.detach(), 'gmotion': gmotion,
              'gflow': flow.grad, 'gcorr': corr.grad, 'delta': delta.detach(), 'mask': mask.detach(), 'gdelta': gdelta, 'gmask': gmask,
              'ghidden': hidden.grad}
    _fingerprints(enc, arrays, 'enc.')
    _fingerprints(head, arrays, 'head.')
    _fingerprints(up, arrays, 'up.')
    save('dense_update_block', **arrays)


def golden_resnet_glue():
    g = torch.Generator().manual_seed(24)
    x = torch.relu(torch.randn(2, 5, 13, 18, generator=g)).requires_grad_(True)     # post-ReLU stem output: zeros tie
    pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    y = pool(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    conv_out = torch.randn(2, 6, 7, 9, generator=g, requires_grad=True)
    identity = torch.randn(2, 6, 7, 9, generator=g, requires_grad=True)
    bias = torch.randn(6, generator=g, requires_grad=True)
    out = torch.relu(conv_out + bias[None, :, None, None] + identity)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    save('dense_resnet_glue', pool_x=x.detach(), pool_y=y.detach(), pool_gy=gy, pool_gx=x.grad,
         conv_out=conv_out.detach(), identity=identity.detach(), bias=bias.detach(), out=out.detach(), gout=gout,
         gconv=conv_out.grad, gbias=bias.grad)


if __name__ == '__main__':
    torch.set_num_threads(4)
    golden_cost_mlp()
    golden_flow_head()
    golden_allpairs('even', 8, 12)
    golden_allpairs('odd', 9, 15)
    golden_gru2d()
    golden_gru2d_wide()
    golden_update_block()
    golden_resnet_glue()
    golden_point_volume()

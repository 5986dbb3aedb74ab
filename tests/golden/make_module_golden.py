"""Module-level golden vectors from the REFERENCE's own modules, unchanged, with the oracle-backed
operators substituted for models.csrc (= the native kernels' index semantics, tests/refmodels.py), name-hashed
weights and seeded inputs.  Besides each module's output (and gradients) a few INTERNAL tensors are captured with
hooks -- they pin oracle functions that no op-level fixture reaches:
    PointConv        input of `linear`           -> oracle_pointconv_mix_fwd
    Correlation3D    input of `cost_mlp` (RAFT)  -> oracle_corr3d_gather_fwd
    PointConvDW      grad of weight_net's output -> oracle_pointconv_dw_bwd
    knn_interpolation autograd                   -> oracle_knn_interp_bwd / _bwd_xyz
    convex_upsample                              -> oracle_convex_upsample_fwd

Run in the build container only:  python tests/golden/make_module_golden.py
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
from modelutils import hashed_fill_  # noqa: E402
from models import utils as ref_utils  # noqa: E402
from models.camliraft_l_core import Correlation3D as RefCorr3DRaft, FlowHead3D, GRU3D, MotionEncoder3D  # noqa: E402
from models.camlipwc_l_core import Correlation3D as RefCorr3DPwc  # noqa: E402
from models.clfm import CLFM  # noqa: E402
from models.point_conv import PointConv, PointConvDW  # noqa: E402
from oracle import torch_ops  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def gen(seed):
    return torch.Generator().manual_seed(seed)


def params(mod):
    return {'p_' + n.replace('.', '__'): t for n, t in mod.state_dict().items()}


def grad_norms(mod):
    named = [(n, p) for n, p in mod.named_parameters() if p.grad is not None]
    return {'gn_names': np.array([n for n, _ in named]), 'gn_values': np.array([p.grad.double().norm().item() for _, p in named])}


def cloud(g, b, n, scale=4.0):
    return torch.rand(b, 3, n, generator=g) * scale


def golden_pointconv():
    g = gen(301)
    mod = hashed_fill_(PointConv(13, 24, norm=None, k=16)).eval()
    xyz, feat = cloud(g, 2, 300), torch.randn(2, 13, 300, generator=g, requires_grad=True)
    sampled = xyz[:, :, :150].contiguous()
    captured = {}
    h1 = mod.linear.register_forward_pre_hook(lambda m, i: captured.update(mixed=i[0].detach()))
    h2 = mod.weight_net.register_forward_hook(lambda m, i, o: captured.update(wgt=o.detach()))
    out = mod(xyz, feat, sampled)
    h1.remove(), h2.remove()
    knn = torch_ops.k_nearest_neighbor(xyz, sampled, 16)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('module_pointconv', xyz=xyz, feat=feat, sampled=sampled, knn=knn, out=out, mixed=captured['mixed'],
         wgt=captured['wgt'], grad_out=go, gfeat=feat.grad, **params(mod), **grad_norms(mod))


def golden_pointconv_dw_bwd():
    g = gen(307)
    mod = hashed_fill_(PointConvDW(20, 32, k=16))
    xyz, feat = cloud(g, 2, 256), torch.randn(2, 20, 256, generator=g, requires_grad=True)
    knn = torch_ops.k_nearest_neighbor(xyz, xyz, 32)
    captured = {}

    def hook(m, i, o):
        o.retain_grad()
        captured['weight'] = o
    h = mod.weight_net.register_forward_hook(hook)
    def hook2(m, i, o):
        o.retain_grad()
        captured['mlp_out'] = o
    h2 = mod.mlp.register_forward_hook(hook2)
    out = mod(xyz, feat, knn_indices=knn)
    h.remove(), h2.remove()
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('module_pointconv_dw_bwd', xyz=xyz, feat=feat, knn=knn, k=16, out=out, grad_out=go, weight=captured['weight'],
         gweight=captured['weight'].grad, mlp_out=captured['mlp_out'], gmlp_out=captured['mlp_out'].grad, gfeat=feat.grad,
         **params(mod), **grad_norms(mod))


def golden_corr3d_raft():
    g = gen(311)
    mod = hashed_fill_(RefCorr3DRaft(out_channels=128, k=16))
    n = 256
    xyz1 = cloud(g, 2, n)
    base2 = xyz1 + torch.randn(2, 3, n, generator=g) * 0.2
    xyzs2 = [base2[:, :, :m].contiguous() for m in (256, 128, 64, 32)]        # nested prefixes, like the FPS pyramid
    f1 = torch.randn(2, 128, n, generator=g, requires_grad=True)
    f2 = torch.randn(2, 128, n, generator=g, requires_grad=True)
    mod.build_cost_volume_pyramid(f1, f2, xyzs2)
    captured = {}
    def pre_hook(m, i):            # returns None: a pre-hook's return value would replace the input
        if 'lookup0' not in captured:
            captured['lookup0'] = i[0].detach()
    h = mod.cost_mlp.register_forward_pre_hook(pre_hook)
    out = mod(xyz1, xyzs2)
    h.remove()
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    cross0 = torch_ops.k_nearest_neighbor(xyzs2[0], xyz1, 16)
    save('module_corr3d_raft', xyz1=xyz1, xyz2=base2, f1=f1, f2=f2, out=out, grad_out=go, gf1=f1.grad, gf2=f2.grad,
         lookup0=captured['lookup0'], cross0=cross0, level3=mod.cost_volume_pyramid[3], **params(mod), **grad_norms(mod))


def golden_corr3d_pwc():
    g = gen(313)
    mod = hashed_fill_(RefCorr3DPwc(32, 32, 64))
    n = 300
    xyz1 = cloud(g, 2, n)
    xyz2 = (xyz1 + torch.randn(2, 3, n, generator=g) * 0.2).requires_grad_(True)
    f1 = torch.randn(2, 32, n, generator=g, requires_grad=True)
    f2 = torch.randn(2, 32, n, generator=g, requires_grad=True)
    own = torch_ops.k_nearest_neighbor(xyz1, xyz1, 16)
    out = mod(xyz1, f1, xyz2, f2, own)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('module_corr3d_pwc', xyz1=xyz1, xyz2=xyz2, f1=f1, f2=f2, own=own, out=out, grad_out=go, gf1=f1.grad, gf2=f2.grad,
         gxyz2=xyz2.grad, **params(mod), **grad_norms(mod))


def golden_clfm():
    g = gen(317)
    mod = hashed_fill_(CLFM(32, 32, fusion_fn='sk', norm=None))
    b, h, w, n = 2, 12, 20, 300
    uv = torch.rand(b, 2, n, generator=g) * torch.tensor([w - 1.0, h - 1.0]).view(1, 2, 1)
    f2d = torch.randn(b, 32, h, w, generator=g, requires_grad=True)
    f3d = torch.randn(b, 32, n, generator=g, requires_grad=True)
    o2d, o3d = mod(uv, f2d, f3d)
    g2, g3 = torch.randn(o2d.shape, generator=g), torch.randn(o3d.shape, generator=g)
    torch.autograd.backward([o2d, o3d], [g2, g3])
    save('module_clfm', uv=uv, f2d=f2d, f3d=f3d, out2d=o2d, out3d=o3d, g2d=g2, g3d=g3, gf2d=f2d.grad, gf3d=f3d.grad,
         **params(mod), **grad_norms(mod))


def golden_update_blocks():
    g = gen(331)
    n = 256
    xyz = cloud(g, 2, n)
    knn = torch_ops.k_nearest_neighbor(xyz, xyz, 32)
    gru = hashed_fill_(GRU3D(input_dim=64, hidden_dim=32))
    h = torch.tanh(torch.randn(2, 32, n, generator=g)).requires_grad_(True)
    x = torch.randn(2, 64, n, generator=g, requires_grad=True)
    out = gru(xyz, h, x, knn)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('module_gru3d', xyz=xyz, knn=knn, h=h, x=x, out=out, grad_out=go, gh=h.grad, gx=x.grad, **params(gru), **grad_norms(gru))

    me = hashed_fill_(MotionEncoder3D(corr_dim=128))
    flow = (torch.randn(2, 3, n, generator=g) * 0.1).requires_grad_(True)
    corr = torch.randn(2, 128, n, generator=g, requires_grad=True)
    out = me(xyz, flow, corr, knn)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('module_motion3d', xyz=xyz, knn=knn, flow=flow, corr=corr, out=out, grad_out=go, gflow=flow.grad, gcorr=corr.grad,
         **params(me), **grad_norms(me))

    fh = hashed_fill_(FlowHead3D(input_dim=128))
    feat = torch.randn(2, 128, n, generator=g, requires_grad=True)
    out = fh(xyz, feat, knn)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('module_flowhead3d', xyz=xyz, knn=knn, feat=feat, out=out, grad_out=go, gfeat=feat.grad, **params(fh), **grad_norms(fh))


def golden_functional():
    g = gen(337)
    flow = torch.randn(2, 2, 6, 9, generator=g)
    mask = torch.randn(2, 9 * 64, 6, 9, generator=g)
    save('convex_upsample', flow=flow, mask=mask, out8=ref_utils.convex_upsample(flow, mask, scale_factor=8),
         mask4=mask[:, :9 * 16], out4=ref_utils.convex_upsample(flow, mask[:, :9 * 16], scale_factor=4))

    in_xyz = torch.randn(2, 3, 200, generator=g).requires_grad_(True)
    feat = torch.randn(2, 7, 200, generator=g, requires_grad=True)
    q_xyz = torch.randn(2, 3, 90, generator=g)
    q_xyz[:, :, :5] = in_xyz.detach()[:, :, :5]          # coincident points: clamp / norm-at-zero subgradients
    q_xyz.requires_grad_(True)
    knn = torch_ops.k_nearest_neighbor(in_xyz, q_xyz, 3)
    out = ref_utils.knn_interpolation(in_xyz, feat, q_xyz, k=3)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save('knn_interpolation_grad', in_xyz=in_xyz, feat=feat, q_xyz=q_xyz, knn=knn, out=out, grad_out=go, gfeat=feat.grad,
         g_in_xyz=in_xyz.grad, g_q_xyz=q_xyz.grad)


def golden_input_side():
    """the reference's own preprocessing: InputPadder + ImageNet normalisation (camliraft.py:38-46) and the
    inverse-depth-scaling transform (ids.py:4-33)"""
    from models.ids import persp2paral as ref_persp2paral
    g = gen(347)
    b, h, w, n = 2, 13, 22, 400
    images = torch.randint(0, 256, (b, 6, h, w), generator=g).float()
    padder = ref_utils.InputPadder(images.shape, x=8)
    image1, image2 = padder.pad(images[:, :3], images[:, 3:])
    mean = torch.tensor([123.675, 116.280, 103.530]).reshape(1, 3, 1, 1)
    std = torch.tensor([58.395, 57.120, 57.375]).reshape(1, 3, 1, 1)
    image1, image2 = (image1 - mean) / std, (image2 - mean) / std
    intr = torch.tensor([[1050.0, 479.5, 269.5], [721.5, 609.6, 172.9]])
    persp = {'projection_mode': 'perspective', 'sensor_h': 544, 'sensor_w': 960, 'f': intr[:, 0], 'cx': intr[:, 1], 'cy': intr[:, 2]}
    paral = {'projection_mode': 'parallel', 'sensor_h': 17, 'sensor_w': 30, 'cx': 14.5, 'cy': 8.0}
    z = torch.rand(b, 2, n, generator=g) * 30 + 5
    u = torch.rand(b, 2, n, generator=g) * 959
    v = torch.rand(b, 2, n, generator=g) * 543
    x = (u - intr[:, 1].view(b, 1, 1)) * z / intr[:, 0].view(b, 1, 1)
    y = (v - intr[:, 2].view(b, 1, 1)) * z / intr[:, 0].view(b, 1, 1)
    pcs = torch.stack([x[:, 0], y[:, 0], z[:, 0], x[:, 1], y[:, 1], z[:, 1]], dim=1)
    save('input_side', images=images, pad=padder._pad, image1=image1, image2=image2, pcs=pcs, intrinsics=intr,
         persp_hw=[544, 960], paral_hw=[17, 30], pc1=ref_persp2paral(pcs[:, :3], persp, paral),
         pc2=ref_persp2paral(pcs[:, 3:], persp, paral))


def golden_projection():
    """the reference's project_pc2image (utils.py:234-259) under both cameras, followed by the feature-grid rescale its
    callers apply in place (camliraft_core.py:51-56)"""
    g = gen(911)
    b, n = 2, 500
    intr = torch.tensor([[1050.0, 479.5, 269.5], [721.5, 609.6, 172.9]])
    persp = {'projection_mode': 'perspective', 'sensor_h': 544, 'sensor_w': 960, 'f': intr[:, 0], 'cx': intr[:, 1], 'cy': intr[:, 2]}
    paral = {'projection_mode': 'parallel', 'sensor_h': 17, 'sensor_w': 30, 'cx': 14.5, 'cy': 8.0}
    z = torch.rand(b, n, generator=g) * 30 + 5
    pc_persp = torch.stack([(torch.rand(b, n, generator=g) - 0.5) * z, (torch.rand(b, n, generator=g) - 0.5) * z * 0.6, z], dim=1)
    pc_paral = torch.stack([torch.rand(b, n, generator=g) * 29 - 14.5, torch.rand(b, n, generator=g) * 16 - 8.0,
                            torch.rand(b, n, generator=g) * 60 + 40], dim=1)
    out = {}
    for name, pc, cam in (('persp', pc_persp, persp), ('paral', pc_paral, paral)):
        uv = ref_utils.project_pc2image(pc, cam)
        out['uv_' + name] = uv.clone()
        grid_h, grid_w = 68, 120
        uv[:, 0] *= (grid_w - 1) / (cam['sensor_w'] - 1)
        uv[:, 1] *= (grid_h - 1) / (cam['sensor_h'] - 1)
        out['uv_grid_' + name] = uv
        out['pc_' + name] = pc
    save('project_pc2image', intrinsics=intr, persp_hw=[544, 960], paral_hw=[17, 30], paral_c=[14.5, 8.0], grid_hw=[68, 120], **out)


if __name__ == '__main__':
    torch.set_num_threads(8)
    if len(sys.argv) > 1:       # regenerate single fixtures:  python make_module_golden.py golden_projection ...
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    golden_pointconv()
    golden_pointconv_dw_bwd()
    golden_corr3d_raft()
    golden_corr3d_pwc()
    golden_clfm()
    golden_update_blocks()
    golden_functional()
    golden_input_side()

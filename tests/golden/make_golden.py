"""Generate the op-level golden vectors from the REFERENCE's own Python path.

Run in the build container only:  python tests/golden/make_golden.py
It imports /root/reference through tests/refshim.py (pure-torch fallbacks of models/csrc/wrapper.py,
models/utils.py, models/raft_core.py) and writes small .npz files next to this script.  The files
hold inputs + expected outputs only; no reference source travels.

KNN note: the reference's importable path is ``squared_distance + topk`` (wrapper.py:115-117), which
uses |a|^2+|b|^2-2ab and so disagrees with the native kernel's direct-difference distances on
near-ties.  Each KNN fixture therefore carries ``safe`` -- a mask of queries whose k+1 smallest
fp64 distances are separated by a relative gap > 1e-4 -- and the check is exact on those queries.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import refshim  # noqa: E402

refshim.install()
from models.csrc import wrapper as ref_ops  # noqa: E402
from models import utils as ref_utils  # noqa: E402
from models.raft_core import Correlation2D as RefCorrelation2D  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('%-40s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def golden_correlation():
    for tag, (b, c, h, w, md) in {'a': (2, 32, 12, 20, 4), 'b': (1, 96, 9, 15, 4), 'c': (1, 8, 5, 6, 2)}.items():
        g = gen(100 + ord(tag))
        x1 = torch.randn(b, c, h, w, generator=g, requires_grad=True)
        x2 = torch.randn(b, c, h, w, generator=g, requires_grad=True)
        go = torch.randn(b, (2 * md + 1) ** 2, h, w, generator=g)
        out = ref_ops.correlation2d(x1, x2, md, cpp_impl=False)
        out.backward(go)
        save('corr2d_' + tag, input1=x1, input2=x2, md=md, grad_output=go, output=out, grad1=x1.grad, grad2=x2.grad)


def golden_fps():
    g = gen(7)
    xyz = torch.rand(2, 1024, 3, generator=g) * 10
    save('fps_a', xyz=xyz, n_samples=256, indices=ref_ops.furthest_point_sampling(xyz, 256, cpp_impl=False))
    xyz = torch.randn(1, 4100, 3, generator=g) * 3
    save('fps_b', xyz=xyz, n_samples=4096, indices=ref_ops.furthest_point_sampling(xyz, 4096, cpp_impl=False))
    # duplicated points: exact ties in the arg-max (25 % of the cloud repeated)
    xyz = torch.rand(1, 600, 3, generator=g)
    xyz[0, 450:] = xyz[0, :150]
    save('fps_dup', xyz=xyz, n_samples=500, indices=ref_ops.furthest_point_sampling(xyz, 500, cpp_impl=False))


def knn_safe_mask(inp, query, k, rel=1e-4):
    d = ((query.double()[:, :, None, :] - inp.double()[:, None, :, :]) ** 2).sum(-1)
    top = torch.sort(d, dim=-1).values[..., :min(k + 1, d.shape[-1])]
    gap = top[..., 1:] - top[..., :-1]
    return (gap > rel * top[..., 1:].clamp_min(1e-12)).all(-1)


def golden_knn():
    g = gen(11)
    for dim in (2, 3):
        for (m, n) in ((512, 256), (256, 512)):
            inp = torch.rand(2, m, dim, generator=g) * 8
            qry = torch.rand(2, n, dim, generator=g) * 8
            for k in (1, 3, 16, 32):
                idx = ref_ops.k_nearest_neighbor(inp, qry, k, cpp_impl=False)
                idx_cf = ref_ops.k_nearest_neighbor(inp.transpose(1, 2), qry.transpose(1, 2), k, cpp_impl=False)
                assert torch.equal(idx, idx_cf)
                save('knn_d%d_m%d_n%d_k%d' % (dim, m, n, k), input=inp, query=qry, k=k, indices=idx,
                     safe=knn_safe_mask(inp, qry, k))
    # self query (distance 0 to itself)
    pts = torch.randn(1, 300, 3, generator=g)
    idx = ref_ops.k_nearest_neighbor(pts, pts, 8, cpp_impl=False)
    save('knn_self', input=pts, query=pts, k=8, indices=idx, safe=knn_safe_mask(pts, pts, 8))


def golden_indexing():
    g = gen(13)
    data = torch.randn(2, 5, 40, generator=g)
    idx = torch.randint(0, 40, (2, 17, 3), generator=g)
    save('batch_indexing', data=data, indices=idx,
         out_cf=ref_utils.batch_indexing(data, idx),
         out_cl=ref_utils.batch_indexing(data.transpose(1, 2).contiguous(), idx, layout='channel_last'))
    in_xyz = torch.randn(2, 3, 200, generator=g)
    feat = torch.randn(2, 7, 200, generator=g)
    q_xyz = torch.randn(2, 3, 90, generator=g)
    q_xyz[:, :, :5] = in_xyz[:, :, :5]  # coincident points: exercises clamp(1e-8)
    knn = ref_ops.k_nearest_neighbor(in_xyz, q_xyz, 3, cpp_impl=False)
    out = ref_utils.knn_interpolation(in_xyz, feat, q_xyz, k=3)
    save('knn_interpolation', in_xyz=in_xyz, feat=feat, q_xyz=q_xyz, knn=knn, out=out)


def golden_allpairs():
    for tag, (b, h, w) in {'even': (1, 16, 16), 'odd': (1, 17, 18)}.items():
        g = gen(17 + len(tag))
        corr = RefCorrelation2D(num_levels=4, radius=4)
        with torch.no_grad():
            for p in corr.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        f1 = torch.randn(b, 128, h, w, generator=g, requires_grad=True)
        f2 = torch.randn(b, 128, h, w, generator=g, requires_grad=True)
        corr.build_cost_volume_pyramid(f1, f2)
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                indexing='ij')
        coords = torch.stack([xs, ys], 0)[None].repeat(b, 1, 1, 1)
        coords = coords + torch.randn(b, 2, h, w, generator=g) * 3.0
        coords[:, :, 0, 0] = -7.5            # far outside
        coords[:, 0, 1, 1] = w + 2.25
        coords[:, :, 2, 2] = torch.tensor([3.0, 4.0])  # exactly integral
        out = corr(coords)
        go = torch.randn(out.shape, generator=g)
        levels = [v.detach().clone() for v in corr.cost_volume_pyramid]
        gl = torch.autograd.grad(out, corr.cost_volume_pyramid, go, retain_graph=True)
        gf1, gf2 = torch.autograd.grad(out, [f1, f2], go)
        sd = corr.state_dict()
        extra = {}
        if tag == 'even':  # the pyramid itself (and its gradient) only for the small case
            for l in range(4):
                extra['level%d' % l] = levels[l]
                extra['glevel%d' % l] = gl[l]
        save('allpairs_' + tag, fmap1=f1, fmap2=f2, aligner_weight=sd['fnet_aligner.weight'],
             aligner_bias=sd['fnet_aligner.bias'], coords=coords, out=out, grad_out=go,
             gfmap1=gf1, gfmap2=gf2, **extra)


def golden_pointconv_dw():
    """The reference's PointConvDW (models/point_conv.py:102-130) with the name-hashed fill: inputs,
    the weight_net parameters and output (captured by a forward hook), and the module output."""
    from models.point_conv import PointConvDW as RefPointConvDW
    from modelutils import hashed_fill_
    for tag, (cin, cout, k, n) in {'a': (24, 40, 16, 96), 'b': (6, 125, 4, 50)}.items():
        g = gen(31 + ord(tag))
        mod = hashed_fill_(RefPointConvDW(cin, cout, k=k)).eval()
        xyz = torch.rand(2, 3, n, generator=g) * 4
        feat = torch.randn(2, cin, n, generator=g)
        knn = ref_ops.k_nearest_neighbor(xyz, xyz, 32, cpp_impl=False)
        captured = {}
        hook = mod.weight_net.register_forward_hook(lambda m, i, o: captured.update(offset=i[0], weight=o))
        with torch.no_grad():
            out = mod(xyz, feat, knn_indices=knn)
        hook.remove()
        sd = mod.state_dict()
        save('pointconv_dw_' + tag, xyz=xyz, feat=feat, knn=knn, k=k, out=out, knn_offset=captured['offset'],
             weight=captured['weight'], **{'p_' + name.replace('.', '__'): t for name, t in sd.items()})


def golden_grid_sample():
    """grid_sample_wrapper of the reference (models/utils.py:262-269): interior, border-crossing, far-out
    and exactly-integer positions."""
    g = gen(41)
    feat = torch.randn(2, 5, 9, 13, generator=g)
    uv = torch.rand(2, 2, 64, generator=g) * torch.tensor([16.0, 12.0]).view(1, 2, 1) - 2.0
    uv[:, :, :8] = torch.randint(0, 9, (2, 2, 8), generator=g).float()          # exact pixel centres
    uv[:, :, 8:12] = torch.tensor([[-50.0, 1e6, 12.0, 11.999], [3.0, -1e6, 8.0, 7.5]]).unsqueeze(0)
    save('grid_sample', feat=feat, uv=uv, out=ref_utils.grid_sample_wrapper(feat, uv))


def golden_ids_flow():
    """paral2persp(pc1 + flow) - paral2persp(pc1) with the reference's models/ids.py (camliraft.py:108-110)."""
    from models.ids import paral2persp as ref_paral2persp, persp2paral as ref_persp2paral
    g = gen(53)
    b, n = 2, 300
    intr = torch.tensor([[1050.0, 479.5, 269.5], [721.5, 609.6, 172.9]])
    persp = {'projection_mode': 'perspective', 'sensor_h': 544, 'sensor_w': 960, 'f': intr[:, 0], 'cx': intr[:, 1], 'cy': intr[:, 2]}
    paral = {'projection_mode': 'parallel', 'sensor_h': 17, 'sensor_w': 30, 'cx': 14.5, 'cy': 8.0}
    z = torch.rand(b, n, generator=g) * 30 + 5
    u = torch.rand(b, n, generator=g) * 959
    v = torch.rand(b, n, generator=g) * 543
    pc = torch.stack([(u - intr[:, 1:2]) * z / intr[:, 0:1], (v - intr[:, 2:3]) * z / intr[:, 0:1], z], dim=1)
    pc1 = ref_persp2paral(pc, persp, paral)
    flow = torch.randn(b, 3, n, generator=g) * 0.3
    origin = ref_paral2persp(pc1, persp, paral)
    out = ref_paral2persp(pc1 + flow, persp, paral) - origin
    save('ids_flow', pc1=pc1, flow=flow, origin=origin, out=out, intrinsics=intr, persp_hw=[544, 960], paral_hw=[17, 30])


if __name__ == '__main__':
    torch.set_num_threads(8)
    golden_correlation()
    golden_fps()
    golden_knn()
    golden_indexing()
    golden_allpairs()
    golden_pointconv_dw()
    golden_grid_sample()
    golden_ids_flow()

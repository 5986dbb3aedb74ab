"""A/B of the point cost-volume lookup of one GRU iteration on one GPU: gather launch + cost-MLP launch (round 3) against
the one-launch form with the gather folded into the MLP kernels (round 4), forward and adjoint, HIP-graph replayed.

  python tools/ab_corr3d.py [--batch 8] [--points 2048]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ab_knn import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    from camliflow_amd.csrc import fused, wrapper
    b, n = args.batch, args.points
    sizes = [n, n // 2, n // 4, n // 8]
    gen = torch.Generator(device='cuda').manual_seed(0)
    xyz1 = torch.rand(b, 3, n, device='cuda', generator=gen) * 4
    xyz2 = xyz1 + torch.randn(b, 3, n, device='cuda', generator=gen) * 0.2
    tables = wrapper.k_nearest_neighbor_prefixes(xyz2.transpose(1, 2).contiguous(), xyz1.transpose(1, 2).contiguous(), sizes, 16)
    levels = [torch.randn(b, n, m, device='cuda', generator=gen) for m in sizes]
    c1, c2 = torch.nn.Conv2d(4, 32, 1).cuda(), torch.nn.Conv2d(32, 32, 1).cuda()
    gout = torch.randn(b, 128, n, device='cuda', generator=gen)
    pyr = fused.Corr3DPyramid(levels)
    pyr.grads = [torch.zeros_like(lvl) for lvl in levels]
    lib = fused._lib.load()
    import ctypes
    ws = torch.empty(lib.camli_corr3d_mlp_bwd_workspace_bytes(b, n) // 4, device='cuda')
    gparams = [torch.zeros_like(t) for t in (c1.weight, c1.bias, c2.weight, c2.bias)]
    szs = (ctypes.c_int * 4)(*sizes)
    stream = lambda: fused._stream_ptr(xyz1)       # noqa: E731
    lookup = torch.empty(b, 4, n, 64, device='cuda')
    glookup = torch.empty_like(lookup)
    out = torch.empty(b, 128, n, device='cuda')
    P = fused._ptr_array

    def split_fwd():
        lib.camli_corr3d_gather_levels_fwd(xyz1.data_ptr(), xyz2.data_ptr(), P(levels), P(tables), szs, 4, lookup.data_ptr(), b, n,
                                           n, 16, stream())
        lib.camli_corr3d_mlp_fwd(lookup.data_ptr(), c1.weight.data_ptr(), c1.bias.data_ptr(), c2.weight.data_ptr(),
                                 c2.bias.data_ptr(), out.data_ptr(), b, n, 4, 16, 32, stream())

    def fold_fwd():
        lib.camli_corr3d_cost_levels_fwd(xyz1.data_ptr(), xyz2.data_ptr(), P(levels), P(tables), szs, c1.weight.data_ptr(),
                                         c1.bias.data_ptr(), c2.weight.data_ptr(), c2.bias.data_ptr(), out.data_ptr(), b, n, n, 4,
                                         16, 32, stream())

    def split_bwd():
        glookup[:, :3].zero_()
        lib.camli_corr3d_mlp_bwd(lookup.data_ptr(), gout.data_ptr(), c1.weight.data_ptr(), c1.bias.data_ptr(),
                                 c2.weight.data_ptr(), c2.bias.data_ptr(), glookup.data_ptr(), gparams[0].data_ptr(),
                                 gparams[1].data_ptr(), gparams[2].data_ptr(), gparams[3].data_ptr(), ws.data_ptr(), b, n, 4, 16,
                                 32, stream())
        lib.camli_corr3d_gather_levels_bwd(glookup.data_ptr(), P(tables), P(pyr.grads), szs, 4, b, n, n, 16, stream())

    def fold_bwd():
        lib.camli_corr3d_cost_levels_bwd(xyz1.data_ptr(), xyz2.data_ptr(), P(levels), P(tables), szs, gout.data_ptr(),
                                         c1.weight.data_ptr(), c1.bias.data_ptr(), c2.weight.data_ptr(), c2.bias.data_ptr(),
                                         P(pyr.grads), gparams[0].data_ptr(), gparams[1].data_ptr(), gparams[2].data_ptr(),
                                         gparams[3].data_ptr(), ws.data_ptr(), b, n, n, 4, 16, 32, stream())

    row = {'case': 'B%d N%d 4x16' % (b, n)}
    for name, fn in (('split_fwd', split_fwd), ('fold_fwd', fold_fwd), ('split_bwd', split_bwd), ('fold_bwd', fold_bwd)):
        row[name + '_us'] = round(timed(fn, args.reps), 2)
    print(json.dumps(row))


if __name__ == '__main__':
    main()

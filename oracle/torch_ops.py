"""Oracle-backed stand-ins for the four boundary operators, taking/returning CPU torch tensors.

TEST INFRASTRUCTURE: tests (and bench.py's cpu_baseline leg) patch these over
``camliflow_amd.csrc.wrapper`` to run the host-side model mirror on CPU with the exact index
semantics of the HIP kernels.  Never imported by the package itself.
"""
import numpy as np
import torch

from . import binding as _b


def k_nearest_neighbor(input_xyz, query_xyz, k, cpp_impl=True):
    if input_xyz.shape[1] <= 3:
        assert query_xyz.shape[1] == input_xyz.shape[1]
        input_xyz = input_xyz.transpose(1, 2).contiguous()
        query_xyz = query_xyz.transpose(1, 2).contiguous()
    out = _b.knn(input_xyz.detach().float().numpy(), query_xyz.detach().float().numpy(), k)
    return torch.from_numpy(out)


def furthest_point_sampling(xyz, n_samples, cpp_impl=True):
    assert xyz.shape[2] == 3 and xyz.shape[1] > n_samples
    return torch.from_numpy(_b.fps(xyz.detach().float().contiguous().numpy(), n_samples))


class _Corr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in1, in2, md):
        ctx.save_for_backward(in1, in2)
        ctx.md = md
        return torch.from_numpy(_b.corr2d_fwd(in1.numpy(), in2.numpy(), md))

    @staticmethod
    def backward(ctx, g):
        in1, in2 = ctx.saved_tensors
        g1, g2 = _b.corr2d_bwd(g.contiguous().numpy(), in1.numpy(), in2.numpy(), ctx.md)
        return torch.from_numpy(g1), torch.from_numpy(g2), None


def correlation2d(input1, input2, max_displacement, cpp_impl=True):
    in1 = input1.permute(0, 2, 3, 1).contiguous().float()
    in2 = input2.permute(0, 2, 3, 1).contiguous().float()
    return _Corr.apply(in1, in2, max_displacement)

"""Point set-convolutions (counterpart of models/point_conv.py:7-70 and :102-130).

``PointConv``    weighted set-conv:  W = MLP2d(3->8->16)(dxyz);  Linear(16*(C+3) -> Cout)(W @ F_knn)
``PointConvDW``  depth-wise set-conv: max_k( gather(MLP1d(F)) * MLP2d(3->8->32->Cout, relu)(dxyz) )
(the reference's ``PointNet2`` is never instantiated and is not restated.)
"""
import contextlib

import torch
import torch.nn as nn

from ..csrc import wrapper as _ops
from . import runtime
from .blocks import LayerNormCF1d, MLP1d, MLP2d, make_activation
from .geometry import batch_indexing

# Iteration-invariant state of one forward pass.  Inside ``pass_cache()`` the fused PointConvDW
# path computes ``weight_net(knn_offset)`` once per (module, xyz, centres, knn_indices, k) and
# reuses it for every GRU iteration; outside it nothing is cached.
_pass_cache = None


@contextlib.contextmanager
def pass_cache():
    global _pass_cache
    prev, _pass_cache = _pass_cache, {}
    try:
        yield
    finally:
        _pass_cache = prev


def _neighbourhood(xyz, sampled_xyz, knn_indices, k):
    """Shared prologue: resolve centres, slice / compute the k nearest, centre the neighbours.
    A precomputed index tensor may be wider than k: KNN output is ascending, so its first k columns
    are the k nearest (point_conv.py:50-55,116-120)."""
    if sampled_xyz is None:
        sampled_xyz = xyz
    if knn_indices is None:
        # inside a pass the pyramid clouds are fixed tensors: the feature and context encoders of frame 1 ask for the
        # same neighbour table (camliraft_core.py:45-47), which the pass cache then serves once
        from .geometry import knn_channel_first
        knn_indices = knn_channel_first(xyz, sampled_xyz, k, invariant_input=True, invariant_query=True)
    else:
        bs, n_samples = sampled_xyz.shape[0], sampled_xyz.shape[-1]
        assert knn_indices.shape[:2] == torch.Size([bs, n_samples])
        assert knn_indices.shape[2] >= k
        knn_indices = knn_indices[:, :, :k]
    knn_offset = batch_indexing(xyz, knn_indices) - sampled_xyz[:, :, :, None]  # [B,3,n,k]
    return sampled_xyz, knn_indices, knn_offset


class PointConv(nn.Module):
    def __init__(self, in_channels, out_channels, norm=None, act='leaky_relu', k=16):
        super().__init__()
        self.k = k
        self.weight_net = MLP2d(3, [8, 16], act=act)
        self.linear = nn.Linear(16 * (in_channels + 3), out_channels)
        if norm == 'batch_norm':
            self.norm_fn = nn.BatchNorm1d(out_channels, affine=True)
        elif norm == 'instance_norm':
            self.norm_fn = nn.InstanceNorm1d(out_channels, affine=True)
        elif norm == 'layer_norm':
            self.norm_fn = LayerNormCF1d(out_channels)
        elif norm is None:
            self.norm_fn = nn.Identity()
        else:
            raise NotImplementedError('Unknown normalization function: %s' % norm)
        if act not in ('relu', 'leaky_relu', None):
            raise NotImplementedError('Unknown activation function: %s' % act)
        self.act_fn = make_activation(act)

    def forward(self, xyz, features, sampled_xyz=None, knn_indices=None):
        """xyz [B,3,N], features [B,C,N], sampled_xyz [B,3,n] -> [B,Cout,n]"""
        sampled_xyz, knn_indices, knn_offset = _neighbourhood(xyz, sampled_xyz, knn_indices, self.k)
        bs, n_samples = sampled_xyz.shape[0], sampled_xyz.shape[-1]
        points_cl = torch.cat([xyz, features], dim=1).transpose(1, 2)         # [B,N,3+C]
        # k = 16 (every model): atomic-free adjoint.  Any other k has a float-atomic scatter in its adjoint, which a
        # request for deterministic algorithms sends to the torch composition (runtime.atomics_ok)
        if runtime.fused() and (self.k == 16 or runtime.atomics_ok('PointConv(k=%d)' % self.k)):
            from ..csrc import fused     # gather + per-point matmul in one kernel, no [B,n,k,3+C] tensor
            mixed = fused.pointconv_mix(points_cl, self.weight_net(knn_offset), knn_indices, self.k)
            mixed = mixed.view(bs, n_samples, -1)
        else:
            weights = self.weight_net(knn_offset).transpose(1, 2)             # [B,n,16,k]
            knn_points = batch_indexing(points_cl, knn_indices, layout='channel_last')  # [B,n,k,3+C]
            mixed = torch.matmul(weights, knn_points).view(bs, n_samples, -1)     # [B,n,16*(3+C)]
        out = self.linear(mixed).transpose(1, 2)
        return self.act_fn(self.norm_fn(out))


class PointConvDW(nn.Module):
    def __init__(self, in_channels, out_channels, norm=None, act='leaky_relu', k=16):
        super().__init__()
        self.k = k
        self.mlp = MLP1d(in_channels, [out_channels], norm, act)
        self.weight_net = MLP2d(3, [8, 32, out_channels], act='relu')

    def forward(self, xyz, features, sampled_xyz=None, knn_indices=None):
        # round 4: under torch.use_deterministic_algorithms(True) the adjoint's scatter runs on the ordered row kernel (no float
        # atomics, camli_pointconv_dw_bwd_ordered) while a feature row and its two side arrays fit 64 KB of LDS -- up to 5461
        # source points, every PointConvDW of the models; longer rows would need global float atomics and leave HIP
        if runtime.fused() and xyz.is_cuda and (features.shape[-1] * 12 <= 64 * 1024 or runtime.atomics_ok('PointConvDW')):
            return self._forward_fused(xyz, features, sampled_xyz, knn_indices)
        sampled_xyz, knn_indices, knn_offset = _neighbourhood(xyz, sampled_xyz, knn_indices, self.k)
        features = batch_indexing(self.mlp(features), knn_indices)            # [B,Cout,n,k]
        features = features * self.weight_net(knn_offset)
        return torch.max(features, dim=-1)[0]

    def _weightnet_on_matrix_cores(self):
        ok = getattr(self, '_wn_ok', None)
        if ok is None:
            from ..csrc import fused
            ok = self._wn_ok = fused.weightnet_supported(self.weight_net, self.weight_net.convs[-1].conv_fn.out_channels)
        return ok

    def _forward_fused(self, xyz, features, sampled_xyz, knn_indices):
        """Same math through camli_pointconv_dw_{fwd,bwd}: no [B,C,n,k] gather / product tensors,
        one atomic per output element in the backward, neighbour weights shared across the pass."""
        from ..csrc import fused
        centres = xyz if sampled_xyz is None else sampled_xyz
        if knn_indices is None:
            knn_indices = _ops.k_nearest_neighbor(xyz, centres, self.k)
        key = (id(self), xyz.data_ptr(), centres.data_ptr(), knn_indices.data_ptr(), self.k,
               tuple(knn_indices.shape), torch.is_grad_enabled())
        entry = _pass_cache.get(key) if _pass_cache is not None else None
        if entry is None:
            if self._weightnet_on_matrix_cores() and not (xyz.requires_grad or centres.requires_grad):
                # offsets + 3 -> 8 -> 32 -> C in one launch, the wide layer on MFMA (camli_weightnet_fwd/bwd)
                # k-major weights [B,C,k,N] where the forward kernel covers the shape: k coalesced rows per lane and
                # channel instead of an LDS transposition of every weight chunk
                k_major = fused.dw_k_major_ok(self.k, xyz.shape[2])
                weight = fused.weightnet(xyz, centres, knn_indices, self.k, self.weight_net, k_major=k_major)
            else:
                k_major = False
                runtime.fallback('weightnet', 'weight_net is not MLP2d(3,[8,32,C<=128],relu) or the coordinates are differentiable')
                _, _, knn_offset = _neighbourhood(xyz, sampled_xyz, knn_indices, self.k)
                weight = self.weight_net(knn_offset)
            shared = fused.SharedSetConvWeights(weight, k_major=k_major)
            if _pass_cache is not None:
                # the key holds raw addresses: keep the keyed tensors alive so an address is never recycled
                _pass_cache[key] = (shared, (xyz, centres, knn_indices))
        else:
            shared = entry[0]
        return fused.pointconv_dw(self.mlp(features), shared, knn_indices, self.k)

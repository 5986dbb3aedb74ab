"""Index / interpolation / projection helpers of the hot path (counterpart of models/utils.py and
models/ids.py).  Every function keeps the reference's name, argument order and tensor layout.
"""
import torch
from torch.nn.functional import grid_sample, interpolate, pad, softmax, unfold

from ..csrc import wrapper as _ops
from . import runtime


class InputPadder:
    """Replicate-pad H and W up to a multiple of ``x`` (utils.py:7-20): width split left/right,
    height padded at the bottom only."""

    def __init__(self, dims, x=8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // x) + 1) * x - self.ht) % x
        pad_wd = (((self.wd // x) + 1) * x - self.wd) % x
        self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        return [pad(x, self._pad, mode='replicate').contiguous() for x in inputs]

    def bottom_only(self):
        """the padding is rows at the bottom only (always true for the height, true for the width when it is a multiple
        of x already): the cores can then produce un-padded predictions directly (``flow_rows``)."""
        return self._pad[0] == 0 and self._pad[1] == 0 and self._pad[2] == 0

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        if (ht, wd) == (self.ht, self.wd):      # already produced at the original size
            return x
        left, right, top, bottom = self._pad
        return x[..., top:ht - bottom, left:wd - right]


def batch_indexing(batched_data, batched_indices, layout='channel_first'):
    """Gather along the point axis (utils.py:61-104).

    channel_first: data [B,C,N], indices [B,I1..Im] -> [B,C,I1..Im]
    channel_last : data [B,N,C] (or [B,N]), indices [B,I1..Im] -> [B,I1..Im,C]
    """
    assert batched_data.shape[0] == batched_indices.shape[0]
    bs = batched_data.shape[0]
    idx_shape = list(batched_indices.shape[1:])
    if layout == 'channel_first':
        if runtime.fused() and batched_data.is_cuda:
            if batched_data.dim() == 3 and batched_data.is_floating_point():
                from ..csrc import fused
                return fused.gather_points(batched_data, batched_indices)
            runtime.fallback('batch_indexing', 'channel-first data of rank %d / dtype %s' % (batched_data.dim(), batched_data.dtype))
        n_channels = batched_data.shape[1]
        flat = batched_indices.reshape(bs, 1, -1).expand(bs, n_channels, -1).to(torch.int64)
        return torch.gather(batched_data, 2, flat).view([bs, n_channels] + idx_shape)
    if layout == 'channel_last':
        if runtime.fused() and batched_data.is_cuda:
            if batched_data.dim() in (2, 3) and batched_data.is_floating_point():
                from ..csrc import fused
                return fused.gather_rows(batched_data, batched_indices)
            runtime.fallback('batch_indexing', 'channel-last data of rank %d / dtype %s' % (batched_data.dim(), batched_data.dtype))
        rows = torch.arange(bs, dtype=torch.long, device=batched_data.device)
        rows = rows.view([bs] + [1] * len(idx_shape)).expand([bs] + idx_shape)
        if batched_data.dim() == 2:
            return batched_data[rows, batched_indices.to(torch.long)]
        return batched_data[rows, batched_indices.to(torch.long), :]
    raise ValueError(layout)


def build_pc_pyramid(pc1, pc2, n_samples_list):
    """One FPS over both clouds; level l keeps the first n_l picks (utils.py:107-127)."""
    batch_size, _, n_points = pc1.shape
    both = torch.cat([pc1, pc2], dim=0).transpose(1, 2)
    picks = _ops.furthest_point_sampling(both, max(n_samples_list))
    picks1, picks2 = picks[:batch_size], picks[batch_size:]

    identity = torch.arange(n_points, device=pc1.device)[None, :].expand(batch_size, n_points)
    xyzs1, xyzs2, sample_indices1, sample_indices2 = [pc1], [pc2], [identity], [identity]
    for n_samples in n_samples_list:
        sample_indices1.append(picks1[:, :n_samples])
        sample_indices2.append(picks2[:, :n_samples])
        xyzs1.append(batch_indexing(pc1, picks1[:, :n_samples]))
        xyzs2.append(batch_indexing(pc2, picks2[:, :n_samples]))
    return xyzs1, xyzs2, sample_indices1, sample_indices2


def _channel_last(xyz, invariant):
    """[B,3,N] -> contiguous [B,N,3] for the KNN kernel; an operand that is the same tensor in every GRU
    iteration (``invariant``) takes its copy from the pass cache instead of a transpose kernel per call."""
    from . import setconv
    cache = setconv._pass_cache if invariant else None
    if cache is None or xyz.requires_grad:
        return xyz.transpose(1, 2).contiguous()
    key = ('channel_last', xyz.data_ptr(), tuple(xyz.shape))
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = (xyz.detach().transpose(1, 2).contiguous(), xyz)     # keeps xyz alive: the key is its address
    return hit[0]


def knn_channel_first(input_xyz, query_xyz, k, invariant_input=False, invariant_query=False):
    """k_nearest_neighbor on channel-first clouds [B,3,*] with cached layout conversion (and, when BOTH
    operands are iteration-invariant, a cached result: the reference recomputes e.g. the up-sampling
    neighbours of every iterate, camliraft_core.py:142-143 -> utils.py:132)."""
    from . import setconv
    cache = setconv._pass_cache
    if not (runtime.fused() and input_xyz.is_cuda and input_xyz.shape[1] <= 3):
        return _ops.k_nearest_neighbor(input_xyz, query_xyz, k)
    both = invariant_input and invariant_query and cache is not None
    if both:
        key = ('knn', input_xyz.data_ptr(), query_xyz.data_ptr(), tuple(input_xyz.shape), tuple(query_xyz.shape), k)
        hit = cache.get(key)
        if hit is not None:
            return hit[0]
    indices = _ops.k_nearest_neighbor(_channel_last(input_xyz, invariant_input), _channel_last(query_xyz, invariant_query), k)
    if both:
        cache[key] = (indices, input_xyz, query_xyz)
    return indices


def knn_interpolation(input_xyz, input_features, query_xyz, k=3, invariant_input=False, invariant_query=False):
    """Inverse-distance interpolation from the k nearest inputs (utils.py:130-146).
    [B,3,M] x [B,C,M] x [B,3,Nq] -> [B,C,Nq]; gradients flow to features AND coordinates."""
    knn_indices = knn_channel_first(input_xyz, query_xyz, k, invariant_input, invariant_query)
    if runtime.fused() and input_xyz.is_cuda and runtime.atomics_ok('knn_interpolation'):
        if k <= 8:
            from ..csrc import fused
            return fused.knn_interpolate(input_xyz, input_features, query_xyz, knn_indices, k,
                                         invariant=invariant_input and invariant_query)
        runtime.fallback('knn_interpolation', 'k = %d > 8' % k)
    knn_xyz = batch_indexing(input_xyz, knn_indices)
    knn_dists = torch.linalg.norm(knn_xyz - query_xyz[..., None], dim=1).clamp(1e-8)
    knn_weights = 1.0 / knn_dists
    knn_weights = knn_weights / torch.sum(knn_weights, dim=-1, keepdim=True)
    knn_features = batch_indexing(input_features, knn_indices)
    return torch.sum(knn_features * knn_weights[:, None, :, :], dim=-1)


def backwarp_3d(xyz1, xyz2, flow12, k=3):
    """Warp cloud 2 towards cloud 1 with the interpolated inverse flow (utils.py:149-159)."""
    flow21 = knn_interpolation(xyz1 + flow12, -flow12, query_xyz=xyz2, k=k, invariant_query=True)
    return xyz2 + flow21


def backwarp_3d_levels(xyz1, xyz2_levels, flow12, k=3, nested=False):
    """``[backwarp_3d(xyz1, level, flow12) for level in xyz2_levels]`` with the level-independent parts
    (warped source cloud, negated flow, their layout conversion) evaluated once.

    ``nested``: the caller guarantees that level l+1 is the first n_{l+1} points of level l (the FPS pyramid of
    build_pc_pyramid, models/utils.py:121-125).  Every warped point depends on its own position only, so the warped
    coarser levels are prefixes of the warped level 0: one search + one interpolation instead of one per level, and
    the returned levels are views of one tensor."""
    if not (runtime.fused() and xyz1.is_cuda and runtime.atomics_ok('backwarp_3d')):
        return [backwarp_3d(xyz1, level, flow12, k) for level in xyz2_levels]
    from ..csrc import fused
    warped, inverse = xyz1 + flow12, -flow12
    warped_cl = warped.detach().transpose(1, 2).contiguous()
    if nested:
        level0 = xyz2_levels[0]
        knn_indices = _ops.k_nearest_neighbor(warped_cl, _channel_last(level0, True), k)
        warped0 = level0 + fused.knn_interpolate(warped, inverse, level0, knn_indices, k)
        return [warped0] + [warped0[:, :, :lvl.shape[2]] for lvl in xyz2_levels[1:]]
    out = []
    for level in xyz2_levels:
        knn_indices = _ops.k_nearest_neighbor(warped_cl, _channel_last(level, True), k)
        out.append(level + fused.knn_interpolate(warped, inverse, level, knn_indices, k))
    return out


_grid_cache = {}


def mesh_grid(n, h, w, device, channel_first=True):
    """Pixel-coordinate grid [n,2,h,w] (x then y), cached per shape/device (utils.py:161-173)."""
    key = (n, h, w, str(device), channel_first)
    grid = _grid_cache.get(key)
    if grid is None:
        xs = torch.arange(0, w, dtype=torch.float32, device=device).view(1, 1, w).expand(n, h, w)
        ys = torch.arange(0, h, dtype=torch.float32, device=device).view(1, h, 1).expand(n, h, w)
        grid = torch.stack([xs, ys], 1)
        if not channel_first:
            grid = grid.permute(0, 2, 3, 1)
        _grid_cache[key] = grid
    return grid


def backwarp_2d(x, flow12, padding_mode):
    """Bilinear backward warp, align_corners=True (utils.py:176-188)."""
    assert x.size()[-2:] == flow12.size()[-2:]
    batch_size, _, image_h, image_w = x.size()
    target = mesh_grid(batch_size, image_h, image_w, device=x.device) + flow12
    norm = torch.zeros_like(target)
    norm[:, 0] = 2.0 * target[:, 0] / (image_w - 1) - 1.0
    norm[:, 1] = 2.0 * target[:, 1] / (image_h - 1) - 1.0
    return grid_sample(x, norm.permute(0, 2, 3, 1), padding_mode=padding_mode, align_corners=True)


def convex_upsample(flow, mask, scale_factor=8, mask_scale=1.0, mask_bias=None, out_rows=None):
    """RAFT convex upsampling: softmax over the 3x3 neighbourhood (utils.py:191-204).  ``mask_scale``
    is the factor the caller would otherwise multiply the mask by (RAFT: 0.25); ``mask_bias`` [9*S*S] is added to the
    mask first (a caller that leaves the bias of the mask head's last convolution to this op saves one pass over the
    mask tensor on the product path)."""
    if runtime.fused() and flow.is_cuda and runtime.atomics_ok('convex_upsample'):
        if scale_factor in (4, 8):
            from ..csrc import fused
            return fused.convex_upsample(flow, mask, scale_factor, mask_scale, mask_bias, out_rows)
        runtime.fallback('convex_upsample', 'scale factor %d (kernels exist for 4 and 8)' % scale_factor)
    if mask_bias is not None:
        mask = mask + mask_bias.view(1, -1, 1, 1)
    if mask_scale != 1.0:
        mask = mask_scale * mask
    batch_size, _, image_h, image_w = flow.shape
    mask = softmax(mask.float().view(batch_size, 1, 9, scale_factor, scale_factor, image_h, image_w), dim=2)
    patches = unfold(flow.float() * scale_factor, [3, 3], padding=1).view(batch_size, 2, 9, 1, 1, image_h, image_w)
    up = torch.sum(mask * patches, dim=2).permute(0, 1, 4, 2, 5, 3)
    up = up.reshape(batch_size, 2, image_h * scale_factor, image_w * scale_factor)
    return up if out_rows is None else up[:, :, :out_rows]


def resize_flow2d(flow, target_h, target_w):
    origin_h, origin_w = flow.shape[2:]
    if target_h == origin_h and target_w == origin_w:
        return flow
    flow = interpolate(flow, size=(target_h, target_w), mode='bilinear', align_corners=True)
    flow[:, 0] *= target_w / origin_w
    flow[:, 1] *= target_h / origin_h
    return flow


def resize_to_64x(inputs, target, x=64):
    """Bilinear resize up to the next multiple of 64 (utils.py:217-231); flow targets are rescaled."""
    n, c, h, w = inputs.shape
    if h % x == 0 and w % x == 0:
        return inputs, target
    resized_h, resized_w = ((h + x - 1) // x) * x, ((w + x - 1) // x) * x
    inputs = interpolate(inputs, size=(resized_h, resized_w), mode='bilinear', align_corners=True)
    if target is not None:
        target = interpolate(target, size=(resized_h, resized_w), mode='bilinear', align_corners=True)
        target[:, 0] *= resized_w / w
        target[:, 1] *= resized_h / h
    return inputs, target


def project_pc2image(pc, camera_info, grid_hw=None):
    """[B,3,N] -> pixel coordinates [B,2,N] under a perspective or parallel camera (utils.py:234-259).
    ``grid_hw`` = (h, w): additionally rescale to a feature grid of that size, uv * (grid - 1) / (sensor - 1), as every
    caller in the cores does right after the projection (camliraft_core.py:51-56, camlipwc_core.py:112-114); on the
    product path both steps are one launch (camli_project_pc2image)."""
    assert pc.shape[1] == 3
    scale = (1.0, 1.0)
    if grid_hw is not None:
        scale = ((grid_hw[1] - 1) / (camera_info['sensor_w'] - 1), (grid_hw[0] - 1) / (camera_info['sensor_h'] - 1))
    mode = camera_info['projection_mode']
    if runtime.fused() and pc.is_cuda and mode in ('perspective', 'parallel'):
        scalar_camera = not any(isinstance(camera_info[k], torch.Tensor) for k in ('cx', 'cy'))
        tensor_camera = all(isinstance(camera_info.get(k), torch.Tensor) and camera_info[k].dim() == 1
                            and camera_info[k].device == pc.device and camera_info[k].shape[0] == pc.shape[0]
                            for k in ('f', 'cx', 'cy'))       # the kernel reads one value per batch element from device memory
        if (torch.is_grad_enabled() and pc.requires_grad) or not (scalar_camera if mode == 'parallel' else tensor_camera):
            runtime.fallback('project_pc2image', 'differentiable cloud, mixed scalar / tensor camera constants, or constants not [B] on the cloud\'s device')
        else:
            from ..csrc import fused
            return fused.project_pc2image(pc, camera_info, scale)
    batch_size, n_points = pc.shape[0], pc.shape[-1]
    cx, cy = camera_info['cx'], camera_info['cy']
    if isinstance(cx, torch.Tensor):
        cx = cx[:, None].expand(batch_size, n_points)
        cy = cy[:, None].expand(batch_size, n_points)
    if mode == 'perspective':
        f = camera_info['f'][:, None].expand(batch_size, n_points)
        image_x = cx + (f / pc[:, 2, :]) * pc[:, 0, :]
        image_y = cy + (f / pc[:, 2, :]) * pc[:, 1, :]
    elif mode == 'parallel':
        image_x = pc[:, 0, :] + cx
        image_y = pc[:, 1, :] + cy
    else:
        raise NotImplementedError(mode)
    uv = torch.cat([image_x[:, None, :], image_y[:, None, :]], dim=1)
    if grid_hw is not None:
        uv[:, 0] *= scale[0]
        uv[:, 1] *= scale[1]
    return uv


def grid_sample_wrapper(feat_2d, uv):
    """Bilinear sample of [B,C,H,W] at pixel coordinates uv [B,2,N] -> [B,C,N], fp32 (utils.py:262-269)."""
    if runtime.fused() and feat_2d.is_cuda:
        if min(feat_2d.shape[2:]) < 2:
            runtime.fallback('grid_sample_wrapper', 'degenerate %dx%d feature map' % tuple(feat_2d.shape[2:]))
        elif torch.is_grad_enabled() and (feat_2d.requires_grad or uv.requires_grad):
            runtime.fallback('grid_sample_wrapper', 'differentiable inputs (the fusion path detaches them)')
        else:
            from ..csrc import fused
            return fused.bilinear_sample(feat_2d.detach(), uv.detach())       # one gather kernel (camli_bilinear_sample_fwd)
    with torch.autocast(device_type=feat_2d.device.type, enabled=False):
        image_h, image_w = feat_2d.shape[2:]
        new_x = 2.0 * uv[:, 0] / (image_w - 1) - 1.0
        new_y = 2.0 * uv[:, 1] / (image_h - 1) - 1.0
        new_xy = torch.cat([new_x[:, :, None, None], new_y[:, :, None, None]], dim=-1)
        return grid_sample(feat_2d.float(), new_xy, 'bilinear', align_corners=True)[..., 0]


# ----------------------------------------------------------------------------------------------
# inverse depth scaling (models/ids.py): perspective camera <-> low-resolution parallel camera
# ----------------------------------------------------------------------------------------------
def _ids_scales(persp, paral):
    ratio_w = (paral['sensor_w'] - 1) / (persp['sensor_w'] - 1)
    ratio_h = (paral['sensor_h'] - 1) / (persp['sensor_h'] - 1)
    return ratio_w, ratio_h


def persp2paral(xyz, perspect_camera_info, parallel_camera_info):
    """ids.py:4-33: project, take f*log(z)+1 as depth, rescale to the parallel sensor."""
    x, y, z = xyz[:, 0, :], xyz[:, 1, :], xyz[:, 2, :]
    shape = x.shape
    f = perspect_camera_info['f'][:, None].expand(shape)
    cx = perspect_camera_info['cx'][:, None].expand(shape)
    cy = perspect_camera_info['cy'][:, None].expand(shape)
    u = cx + (f / z) * x
    v = cy + (f / z) * y
    depth = f * torch.log(z) + 1
    ratio_w, ratio_h = _ids_scales(perspect_camera_info, parallel_camera_info)
    paral_w, paral_h = parallel_camera_info['sensor_w'], parallel_camera_info['sensor_h']
    return torch.cat([
        u[:, None, :] * ratio_w - (paral_w - 1) / 2,
        v[:, None, :] * ratio_h - (paral_h - 1) / 2,
        depth[:, None, :] * min(ratio_w, ratio_h),
    ], dim=1)


def persp2paral_both(pcs, perspect_camera_info, parallel_camera_info):
    """pcs [B,6,N] (cloud 1 | cloud 2) -> (persp2paral(cloud 1), persp2paral(cloud 2)).  Product path: one launch
    for both clouds (camli_persp2paral, same expression order as ids.py:4-33, bit-identical to the composition)."""
    f = perspect_camera_info['f']
    if (runtime.fused() and pcs.is_cuda and pcs.shape[1] == 6 and not pcs.requires_grad and torch.is_tensor(f)
            and f.dim() == 1 and all(torch.is_tensor(perspect_camera_info[k]) for k in ('cx', 'cy'))):
        from ..csrc import fused
        intrinsics = torch.stack([f, perspect_camera_info['cx'], perspect_camera_info['cy']], dim=1)
        return fused.persp2paral_pair(pcs, intrinsics, perspect_camera_info, parallel_camera_info)
    if pcs.is_cuda:
        runtime.fallback('persp2paral', 'non-tensor intrinsics or differentiable clouds')
    return (persp2paral(pcs[:, :3], perspect_camera_info, parallel_camera_info),
            persp2paral(pcs[:, 3:], perspect_camera_info, parallel_camera_info))


def paral2persp(xyz, perspect_camera_info, parallel_camera_info):
    """ids.py:36-67: inverse of persp2paral."""
    ratio_w, ratio_h = _ids_scales(perspect_camera_info, parallel_camera_info)
    paral_w, paral_h = parallel_camera_info['sensor_w'], parallel_camera_info['sensor_h']
    u = (xyz[:, 0, :] + (paral_w - 1) / 2) / ratio_w
    v = (xyz[:, 1, :] + (paral_h - 1) / 2) / ratio_h
    depth = xyz[:, 2, :] / min(ratio_w, ratio_h)
    shape = u.shape
    f = perspect_camera_info['f'][:, None].expand(shape)
    cx = perspect_camera_info['cx'][:, None].expand(shape)
    cy = perspect_camera_info['cy'][:, None].expand(shape)
    z = torch.exp((depth - 1) / f)
    x = (u - cx) * z / f
    y = (v - cy) * z / f
    return torch.cat([x[:, None, :], y[:, None, :], z[:, None, :]], dim=1)


def flows_paral2persp(pc1, flows, perspect_camera_info, parallel_camera_info):
    """``[paral2persp(pc1 + f) - paral2persp(pc1) for f in flows]`` (camliraft.py:108-110): one fused
    kernel per iterate (and one for its backward) on the product path instead of ~22 pointwise launches."""
    origin = paral2persp(pc1, perspect_camera_info, parallel_camera_info)
    f = perspect_camera_info['f']
    if (runtime.fused() and pc1.is_cuda and torch.is_tensor(f) and f.dim() == 1 and not pc1.requires_grad
            and all(torch.is_tensor(perspect_camera_info[k]) for k in ('cx', 'cy'))):
        from ..csrc import fused
        return [fused.ids_flow(flow, pc1, origin, perspect_camera_info, parallel_camera_info) for flow in flows]
    if pc1.is_cuda:
        runtime.fallback('flows_paral2persp', 'non-tensor intrinsics or differentiable pc1')
    return [paral2persp(pc1 + flow, perspect_camera_info, parallel_camera_info) - origin for flow in flows]

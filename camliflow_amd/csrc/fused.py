"""Composite operators of the cores that have a fused gfx950 kernel (internal API, used when
``cores.runtime.backend() == 'hip'``).  Each function documents the reference composition it
replaces; the torch-composed twin lives next to its call site in ``camliflow_amd/cores``.

Every autograd Function here runs with autocast DISABLED and fp32 inputs (``custom_fwd(cast_inputs=
torch.float32)``): the kernels take raw fp32 device pointers, and the reference keeps these ops in
fp32 under AMP as well (``.float()`` at raft_core.py:53-54, clfm.py:31-32, camliraft_l_core.py:52).
"""
import ctypes
import os
import math
import weakref

import torch
from torch.nn.functional import avg_pool2d

from . import _lib
from ..cores import runtime as _runtime
from .wrapper import _on_device, _require_cuda, _stream_ptr, _zero_slice


# ------------------------------------------------------------------------------------------------
# all-pairs cost-volume pyramid (models/raft_core.py:52-107)
# ------------------------------------------------------------------------------------------------
# CAMLI_ALLPAIRS_SPLITK=0: the pyramid adjoint's g_f2 GEMMs unsplit (A/B switch; default: coarse levels split over K)
_BUILD_SPLITK = os.environ.get('CAMLI_ALLPAIRS_SPLITK', '1') != '0'
# CAMLI_ALLPAIRS_MARKS=0: the pyramid adjoint examines every gradient tile instead of following the lookups' visit marks
_USE_MARKS = os.environ.get('CAMLI_ALLPAIRS_MARKS', '1') != '0'
_ALLPAIRS_FWD_LIB = os.environ.get('CAMLI_ALLPAIRS_FWD', 'hip') == 'lib'


# CAMLI_ALLPAIRS_KEEP=0: allocate and zero-fill the gradient pyramid in every backward pass (the round-3 behaviour)
_KEEP_GRAD_PYRAMID = os.environ.get('CAMLI_ALLPAIRS_KEEP', '1') != '0'
# (device, level shapes) -> [grads, marks, event]: a gradient pyramid and its visit marks, ALL ZERO, kept between steps.  A pass
# takes it (pop), its lookups dirty a band of blocks, the build adjoint reads it and then cleans exactly the marked blocks
# (camli_allpairs_clear_marked) before putting it back; `event` orders that cleaning before the next pass's first write.
# A pass that dies between the two never returns its buffers (the next one allocates fresh zeros), so what sits here is
# always clean.  One entry per shape: a second pass alive at the same time allocates its own.
_clean_grad_pyramids = {}


def release_cached_buffers():
    """Drop what this module keeps between passes: the clean all-pairs gradient pyramids (2.8 GB at batch 8) and the inverse
    index maps.  Nothing needs them -- the next pass allocates / rebuilds -- call it when a model is done with the device."""
    _clean_grad_pyramids.clear()
    _inverse_maps.clear()
    _interp_weights.clear()


class AllPairsPyramid:
    """The 4-level all-pairs volume of one forward pass plus the state its backward needs.

    Autograd design: the levels are plain (non-differentiable) tensors.  Every lookup depends on a
    1-element ``token`` produced by the build node; each lookup's backward ADDS its window
    gradients into ONE persistent gradient pyramid (allocated + zeroed once) and hands a dummy
    gradient to the token, so the build node's backward runs after the last lookup and turns the
    accumulated volume gradient into feature-map gradients with two GEMMs.  The reference instead
    materialises a volume-sized gradient per level per GRU iteration and sums them.
    """

    def __init__(self):
        self.levels = None      # list of [B*P, h_l, w_l] fp32
        self.grads = None       # same shapes, lazily allocated by the first lookup backward
        self.marks = None       # per level [B, ceil(P/32), ceil(P_l/32)] uint8: blocks of the gradient the lookups wrote
        self.token = None
        self.shape = None       # (B, h, w)

    def _level_args(self, tensors):
        n = len(tensors)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
        hs = (ctypes.c_int * n)(*[t.shape[-2] for t in tensors])
        ws = (ctypes.c_int * n)(*[t.shape[-1] for t in tensors])
        return n, ptrs, hs, ws


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _BuildPyramid(torch.autograd.Function):
    """All levels straight from the feature maps on the fp32 matrix cores (camli_allpairs_build_fwd/bwd):
    avg_pool2d acts on the target pixel only and is linear, so level l = f1^T . pool_l(f2) / sqrt(C) -- every volume
    element is written once and never re-read, and the adjoint needs no volume-sized fold (csrc/hip/allpairs.hip)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, fmap1, fmap2, num_levels, pyr):
        lib = _lib.load()
        bs, dim, h, w = fmap1.shape
        p = h * w
        f2_levels = [fmap2]
        for _ in range(num_levels - 1):
            if min(f2_levels[-1].shape[-2:]) < 2:
                break
            f2_levels.append(avg_pool2d(f2_levels[-1], 2, stride=2))     # tiny: [B,C,h_l,w_l]
        sizes = [(t.shape[-2], t.shape[-1]) for t in f2_levels]
        p_levels = (ctypes.c_int * len(sizes))(*[a * b for a, b in sizes])
        levels = [torch.empty((bs * p, a, b), dtype=torch.float32, device=fmap1.device) for a, b in sizes]
        total = sum(a * b for a, b in sizes)
        if _ALLPAIRS_FWD_LIB:
            # A/B switch (CAMLI_ALLPAIRS_FWD=lib): the same four GEMMs through the library (hipBLASLt reaches 0.81 of the fp32
            # MFMA peak on the level-0 shape with 256-wide macro tiles, this repo's 128 x 128 kernel 0.58)
            a = fmap1.flatten(2).transpose(1, 2)
            for lvl, f2l in zip(levels, f2_levels):
                out = lvl.view(bs, p, -1)
                torch.baddbmm(out, a, f2l.flatten(2), beta=0.0, alpha=1.0 / math.sqrt(dim), out=out)
        else:
            with _on_device(fmap1):
                _lib.launch('camli_allpairs_build_fwd', lib.camli_allpairs_build_fwd, fmap1.data_ptr(), _ptr_array(f2_levels),
                            _ptr_array(levels), p_levels, len(sizes), bs, dim, p, 1.0 / math.sqrt(dim), _stream_ptr(fmap1),
                            work=(4.0 * bs * p * total + 4.0 * bs * dim * (p + total), 'B'), flop=2.0 * bs * p * total * dim)
        pyr.levels = levels
        pyr.shape = (bs, h, w)
        ctx.save_for_backward(fmap1, *f2_levels)
        # weak: pyr.token is this node's output, a strong link would close a reference cycle
        # (pyr -> token -> grad_fn -> ctx -> pyr) and keep the 2.8 GB pyramid alive until the cyclic GC runs.
        # Every lookup node holds pyr strongly, so it lives exactly as long as the graph needs it.
        ctx.pyr = weakref.ref(pyr)
        ctx.dims = (bs, dim, h, w)
        ctx.sizes = sizes
        return fmap1.new_zeros(1)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, _gtoken):
        lib = _lib.load()
        pyr = ctx.pyr()
        fmap1, *f2_levels = ctx.saved_tensors
        bs, dim, h, w = ctx.dims
        p = h * w
        grads = marks = keep_key = None
        if pyr is not None:
            grads, pyr.grads = pyr.grads, None
            marks, pyr.marks = pyr.marks, None
            keep_key = getattr(pyr, 'keep_key', None)
        if grads is None:
            return torch.zeros_like(fmap1), torch.zeros_like(fmap1), None, None
        sizes = ctx.sizes
        p_levels = (ctypes.c_int * len(sizes))(*[a * b for a, b in sizes])
        total = sum(a * b for a, b in sizes)
        g1 = torch.empty_like(fmap1)
        g2_levels = [torch.empty_like(t) for t in f2_levels]
        work = dict(work=(4.0 * bs * p * total + 4.0 * bs * dim * 2 * (p + total), 'B'), flop=4.0 * bs * p * total * dim)
        with _on_device(fmap1):
            ws_bytes = int(lib.camli_allpairs_build_bwd_workspace_bytes(p_levels, len(sizes), bs, dim, p)) if _BUILD_SPLITK else 0
            if marks is not None and ws_bytes > 0:
                # ... and the coarse levels' g_f2 GEMMs (few tiles, the longest K loops) are split over K into a workspace
                ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=fmap1.device)
                _lib.launch('camli_allpairs_build_bwd', lib.camli_allpairs_build_bwd_splitk, fmap1.data_ptr(), _ptr_array(f2_levels),
                            _ptr_array(grads), p_levels, len(sizes), g1.data_ptr(), _ptr_array(g2_levels), bs, dim, p,
                            1.0 / math.sqrt(dim), _ptr_array(marks), ws.data_ptr(), ws_bytes, _stream_ptr(fmap1), **work)
            elif marks is not None:
                # the lookups marked the 32x32 blocks they wrote: the GEMMs never load the rest of the volume
                _lib.launch('camli_allpairs_build_bwd', lib.camli_allpairs_build_bwd_marked, fmap1.data_ptr(), _ptr_array(f2_levels),
                            _ptr_array(grads), p_levels, len(sizes), g1.data_ptr(), _ptr_array(g2_levels), bs, dim, p,
                            1.0 / math.sqrt(dim), _ptr_array(marks), _stream_ptr(fmap1), **work)
            else:       # a gradient pyramid that did not come from the lookups (tests): every tile is examined
                _lib.launch('camli_allpairs_build_bwd', lib.camli_allpairs_build_bwd, fmap1.data_ptr(), _ptr_array(f2_levels),
                            _ptr_array(grads), p_levels, len(sizes), g1.data_ptr(), _ptr_array(g2_levels), bs, dim, p,
                            1.0 / math.sqrt(dim), _stream_ptr(fmap1), **work)
            if marks is not None and keep_key is not None and keep_key not in _clean_grad_pyramids:
                # put the pyramid back to all-zero -- only the marked blocks were ever written -- and keep it for the next pass
                _lib.launch('camli_allpairs_clear_marked', lib.camli_allpairs_clear_marked, _ptr_array(grads), p_levels, len(sizes),
                            _ptr_array(marks), bs, p, _stream_ptr(fmap1), work=(0.2 * 4.0 * bs * p * total, 'B'))
                cleaned = torch.cuda.Event()
                cleaned.record(torch.cuda.current_stream(fmap1.device))
                while len(_clean_grad_pyramids) >= 2:        # other shapes (another batch size): keep the two latest
                    _clean_grad_pyramids.pop(next(iter(_clean_grad_pyramids)))
                _clean_grad_pyramids[keep_key] = [grads, marks, cleaned]
        # adjoint of the avg_pool2d chain on the SMALL maps ([B,C,h_l,w_l]): fold the coarse levels into level 0
        g2 = g2_levels[-1]
        for lvl in range(len(g2_levels) - 2, -1, -1):
            h2, w2 = g2.shape[-2:]
            up = g2.repeat_interleave(2, dim=-2).repeat_interleave(2, dim=-1)
            g2_levels[lvl][:, :, :2 * h2, :2 * w2] += up * 0.25
            g2 = g2_levels[lvl]
        return g1, g2, None, None


class _Lookup(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, token, coords, radius, pyr):
        lib = _lib.load()
        bs, h, w = pyr.shape
        assert all(lvl.dtype == torch.float32 and lvl.is_contiguous() for lvl in pyr.levels), 'pyramid must be fp32'
        assert coords.dtype == torch.float32 and coords.is_contiguous()
        n, ptrs, hs, ws = pyr._level_args(pyr.levels)
        d = 2 * radius + 1
        out = torch.empty((bs, n * d * d, h, w), dtype=torch.float32, device=coords.device)
        with _on_device(coords):
            _lib.launch('camli_allpairs_lookup_fwd', lib.camli_allpairs_lookup_fwd, ptrs, hs, ws, n, coords.data_ptr(), out.data_ptr(),
                                                     bs, h, w, radius, _stream_ptr(coords),
                        work=(4.0 * bs * h * w * (n * d * d + n * (d + 1) ** 2 + 2), 'B'))
        ctx.save_for_backward(coords)
        ctx.pyr, ctx.radius = pyr, radius
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        (coords,) = ctx.saved_tensors
        pyr = ctx.pyr
        bs, h, w = pyr.shape
        if pyr.grads is None:
            kept = None
            if _USE_MARKS and _KEEP_GRAD_PYRAMID and not torch.cuda.is_current_stream_capturing():
                # (under graph capture the pass keeps the allocate-and-fill form: a captured replay owns its buffers)
                pyr.keep_key = (coords.device, tuple(tuple(lvl.shape) for lvl in pyr.levels))
                kept = _clean_grad_pyramids.pop(pyr.keep_key, None)
            if kept is not None:        # all zero since the last pass cleaned up after itself
                pyr.grads, pyr.marks, cleaned = kept
                torch.cuda.current_stream(coords.device).wait_event(cleaned)
            else:
                pyr.grads = [torch.zeros_like(lvl) for lvl in pyr.levels]
                sb = (h * w + 31) // 32
                if _USE_MARKS:
                    pyr.marks = [torch.zeros((bs, sb, (lvl.shape[-2] * lvl.shape[-1] + 31) // 32), dtype=torch.uint8,
                                             device=lvl.device) for lvl in pyr.levels]
        gout = gout.contiguous().float()
        n, ptrs, hs, ws = pyr._level_args(pyr.grads)
        work = (4.0 * bs * h * w * (n * (2 * ctx.radius + 1) ** 2 + 2 * n * (2 * ctx.radius + 2) ** 2 + 2), 'B')
        with _on_device(coords):
            if pyr.marks is not None:
                _lib.launch('camli_allpairs_lookup_bwd', lib.camli_allpairs_lookup_bwd_marked, ptrs, hs, ws, n, coords.data_ptr(),
                            gout.data_ptr(), bs, h, w, ctx.radius, _ptr_array(pyr.marks), _stream_ptr(coords), work=work)
            else:       # gradient pyramid supplied from outside: no marks to maintain
                _lib.launch('camli_allpairs_lookup_bwd', lib.camli_allpairs_lookup_bwd, ptrs, hs, ws, n, coords.data_ptr(),
                            gout.data_ptr(), bs, h, w, ctx.radius, _stream_ptr(coords), work=work)
        # no gradient for the token: the engine still runs the token's producer once every consumer has run (it hands it
        # materialised zeros), and a None costs no accumulation -- a shared zero tensor here was one add launch per lookup
        return None, None, None, None


def allpairs_pyramid(fmap1, fmap2, num_levels=4):
    """fmap1/fmap2 [B,C,h,w] (already through ``fnet_aligner``) -> AllPairsPyramid."""
    _require_cuda('allpairs_pyramid', fmap1, fmap2)
    if not (torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)) and _clean_grad_pyramids:
        # a pass that will not run backward (validation between training steps, another model's inference): the gradient
        # pyramid kept clean for the NEXT backward (2.8 GB at batch 8) would sit outside the allocator's reach for its whole
        # duration -- let it go, the next training step zero-fills a fresh one
        _clean_grad_pyramids.clear()
    pyr = AllPairsPyramid()
    pyr.token = _BuildPyramid.apply(fmap1.float().contiguous(), fmap2.float().contiguous(), num_levels, pyr)
    return pyr


def allpairs_lookup(pyr, coords, radius=4):
    """coords [B,2,h,w] -> [B, L*(2r+1)^2, h, w] (raft_core.py:70-94 in one launch)."""
    _require_cuda('allpairs_lookup', coords)
    if coords.requires_grad and torch.is_grad_enabled():
        # the reference's grid_sample is differentiable wrt the coordinates; its RAFT loops detach the flow first
        # (raft_core.py:248, camliraft_core.py:105-106), and this kernel has no coordinate adjoint
        raise _lib.CamliHipError('allpairs_lookup: coordinates that require grad are not supported (detach the flow '
                                 'as the reference loops do, or select the composed backend)')
    return _Lookup.apply(pyr.token, coords.detach().float().contiguous(), radius, pyr)


# ------------------------------------------------------------------------------------------------
# depth-wise set-conv core (models/point_conv.py:122-128)
# ------------------------------------------------------------------------------------------------
class SharedSetConvWeights:
    """The neighbour weights ``weight_net(knn_offset)`` [B,C,N,k] of one PointConvDW instance,
    shared by every call of one forward pass (they depend only on xyz / knn_indices, which are
    fixed across the GRU iterations: models/camliraft_core.py:88,119,127,140).

    Autograd: ``weight`` enters a ``_ShareWeights`` node that hands out a 1-element token.  Every
    fused call depends on the token; its backward leaves a COMPACT record (one value + one slot per
    output element, because max() passes gradient to a single neighbour).  The node's backward
    expands the records of all calls into the dense weight gradient in one pass and returns it, so
    ``weight_net`` is differentiated once per pass instead of once per iteration.
    """

    def __init__(self, weight, k_major=False):
        self.weight = weight.detach().float().contiguous()     # fp32 even when weight_net ran under autocast
        self.k_major = bool(k_major)   # [B,C,k,N] instead of [B,C,N,k]: what weightnet(..., k_major=True) produces
        self.records = []          # [(gwsel [B,C,N] fp32, arg [B,C,N] uint8)] appended by the backward calls
        self.token = _ShareWeights.apply(weight, self)
        self._idx_kn = None        # (knn_indices kept alive, k, int32 [B,k,N]) for the k-major forward

    def idx_kn(self, knn_indices, k):
        """The neighbour table as int32 [B,k,N] (k coalesced rows per lane), built once per pass and table."""
        hit = self._idx_kn
        if hit is None or hit[0] is not knn_indices or hit[1] != k:
            table = knn_indices[:, :, :k].to(torch.int32).permute(0, 2, 1).contiguous()
            hit = self._idx_kn = (knn_indices, k, table)
        return hit[2]

    @property
    def n_points(self):
        return self.weight.shape[3 if self.k_major else 2]

    @property
    def k(self):
        return self.weight.shape[2 if self.k_major else 3]


class _ShareWeights(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, weight, shared):
        # weak for the same reason as _BuildPyramid: shared.token is this node's output; the fused set-conv
        # nodes of the pass hold `shared` strongly
        ctx.shared = weakref.ref(shared)
        ctx.wshape, ctx.wdevice = tuple(weight.shape), weight.device
        return weight.new_zeros(1)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, _gtoken):
        lib = _lib.load()
        shared = ctx.shared()
        if shared is None or not shared.records:
            return torch.zeros(ctx.wshape, dtype=torch.float32, device=ctx.wdevice), None
        records, shared.records = shared.records, []
        weight = shared.weight
        b, c = weight.shape[:2]
        n, k = shared.n_points, shared.k
        grad = torch.empty_like(weight)
        with _on_device(weight):
            for start in range(0, len(records), 64):       # the kernel takes <= 64 calls at a time
                chunk = records[start:start + 64]
                part = grad if start == 0 else torch.empty_like(weight)
                gptrs = (ctypes.c_void_p * len(chunk))(*[g.data_ptr() for g, _ in chunk])
                aptrs = (ctypes.c_void_p * len(chunk))(*[a.data_ptr() for _, a in chunk])
                _lib.launch('camli_pointconv_dw_expand', lib.camli_pointconv_dw_expand, gptrs, aptrs, len(chunk),
                            part.data_ptr(), b, c, n, k, int(shared.k_major), _stream_ptr(weight),
                        work=(4.0 * b * c * n * k + 5.0 * len(chunk) * b * c * n, 'B'))
                if start:
                    grad += part
        return grad, None


class _PointConvDW(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, feat, token, knn_indices, k, shared):
        lib = _lib.load()
        weight = shared.weight
        b, c, m = feat.shape
        if shared.k_major and feat.data_ptr() % 16:
            feat = feat.clone()          # the k-major kernel stages feature rows with 16-byte loads
        n = shared.n_points
        out = torch.empty((b, c, n), dtype=torch.float32, device=feat.device)
        arg = torch.empty((b, c, n), dtype=torch.uint8, device=feat.device)
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]   # grad mode is off inside forward()
        wsel = torch.empty((b, c, n), dtype=torch.float32, device=feat.device) if need_grad else None
        msel = torch.empty((b, c, n), dtype=torch.int32, device=feat.device) if need_grad else None
        work = (4.0 * b * c * n * k + 4.0 * b * c * m + (4.0 if shared.k_major else 8.0) * b * n * k + 13.0 * b * c * n, 'B')
        with _on_device(feat):
            if shared.k_major:
                _lib.launch('camli_pointconv_dw_fwd', lib.camli_pointconv_dw_fwd_kmajor, feat.data_ptr(), weight.data_ptr(),
                            shared.idx_kn(knn_indices, k).data_ptr(), out.data_ptr(), arg.data_ptr(),
                            wsel.data_ptr() if need_grad else None, msel.data_ptr() if need_grad else None,
                            b, c, m, n, k, _stream_ptr(feat), work=work)
            else:
                _lib.launch('camli_pointconv_dw_fwd', lib.camli_pointconv_dw_fwd, feat.data_ptr(), weight.data_ptr(),
                            knn_indices.data_ptr(), knn_indices.stride(1), out.data_ptr(), arg.data_ptr(),
                            wsel.data_ptr() if need_grad else None, msel.data_ptr() if need_grad else None,
                            b, c, m, n, k, _stream_ptr(feat), work=work)
        if need_grad:
            ctx.save_for_backward(feat, wsel, msel, arg)
        ctx.shared = shared
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        feat, wsel, msel, arg = ctx.saved_tensors
        shared = ctx.shared
        b, c, m = feat.shape
        n = wsel.shape[2]
        gout = gout.float()
        gfeat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None   # fully written by the kernel
        gwsel = torch.empty_like(wsel)
        # torch.use_deterministic_algorithms(True): the ordered form (no float atomics, fixed summation order; 1.7x the time)
        ordered = torch.are_deterministic_algorithms_enabled() and 12 * m <= 64 * 1024
        # a channel slice of a wider gradient (the outputs are concatenated downstream) is read in place by the LDS row kernel
        gout_bs = None if (ordered or 8 * m > 64 * 1024) else _batch_strided(gout)
        work = dict(work=(16.0 * b * c * n + 8.0 * b * c * m, 'B'))
        with _on_device(feat):
            if gout_bs is not None and gout_bs != c * n:
                _lib.launch('camli_pointconv_dw_bwd', lib.camli_pointconv_dw_bwd_strided, gout.data_ptr(), gout_bs, feat.data_ptr(),
                            wsel.data_ptr(), msel.data_ptr(), gfeat.data_ptr() if gfeat is not None else None,
                            gwsel.data_ptr(), b, c, m, n, _stream_ptr(feat), **work)
            else:
                gout = gout.contiguous()
                _lib.launch('camli_pointconv_dw_bwd', lib.camli_pointconv_dw_bwd_ordered if ordered else lib.camli_pointconv_dw_bwd,
                            gout.data_ptr(), feat.data_ptr(),
                            wsel.data_ptr(), msel.data_ptr(), gfeat.data_ptr() if gfeat is not None else None,
                            gwsel.data_ptr(), b, c, m, n, _stream_ptr(feat), **work)
        shared.records.append((gwsel, arg))
        return gfeat, None, None, None, None


def pointconv_dw(feat, shared, knn_indices, k):
    """feat [B,C,M] (already through the 1x1 ``mlp``), shared weights [B,C,N,k], knn_indices int64
    [B,N,>=k] (row-contiguous) -> [B,C,N] = max_j feat[:, :, idx_j] * weight[..., j]."""
    _require_cuda('pointconv_dw', feat, knn_indices)
    assert knn_indices.dtype == torch.int64 and knn_indices.stride(2) == 1 and knn_indices.shape[2] >= k
    assert knn_indices.stride(0) == knn_indices.shape[1] * knn_indices.stride(1)
    assert shared.weight.shape[0] == feat.shape[0] and shared.weight.shape[1] == feat.shape[1]
    assert shared.k == k
    return _PointConvDW.apply(feat.float().contiguous(), shared.token, knn_indices, k, shared)


# ------------------------------------------------------------------------------------------------
# selective-kernel fusion, full-size part (models/clfm.py:170-213)
# ------------------------------------------------------------------------------------------------
class SkState:
    """Links the two autograd nodes of one SKFusion call: the mix node leaves (g, w) here and returns no
    gradient for a / b; the pool node -- which autograd necessarily runs later, because the gate that
    produced w hangs off its output -- then writes ga = g*w0 + gs/P and gb = g*w1 + gs/P in ONE pass."""
    __slots__ = ('g', 'w', 'deferred')

    def __init__(self):
        self.g = self.w = None
        self.deferred = False


def _rows(t):
    b, c = t.shape[0], t.shape[1]
    return b, c, t.numel() // max(b * c, 1)


class _SkPool(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, a, b, state):
        lib = _lib.load()
        bs, c, p = _rows(a)
        s = torch.empty((bs, c), dtype=torch.float32, device=a.device)
        with _on_device(a):
            _lib.launch('camli_sk_pool_fwd', lib.camli_sk_pool_fwd, a.data_ptr(), b.data_ptr(), s.data_ptr(), bs, c, p,
                        _stream_ptr(a), work=(8.0 * bs * c * p, 'B'))
        ctx.state, ctx.shape = state, a.shape
        return s

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gs):
        lib = _lib.load()
        st = ctx.state
        bs, c = ctx.shape[0], ctx.shape[1]
        p = 1
        for d in ctx.shape[2:]:
            p *= d
        gs = gs.contiguous().float()
        if st.g is None:        # the mixed output took no part in the loss: only the pooled term
            ga = (gs / p).view(bs, c, *([1] * (len(ctx.shape) - 2))).expand(ctx.shape).contiguous()
            return ga, ga.clone(), None
        g, w = st.g, st.w
        st.g = st.w = None
        ga = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        gb = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        with _on_device(g):
            _lib.launch('camli_sk_mix_bwd_x', lib.camli_sk_mix_bwd_x, g.data_ptr(), w.data_ptr(), gs.data_ptr(),
                        ga.data_ptr(), gb.data_ptr(), bs, c, p, _stream_ptr(g), work=(12.0 * bs * c * p, 'B'))
        return ga, gb, None


class _SkMix(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, a, b, w, state):
        lib = _lib.load()
        bs, c, p = _rows(a)
        w = w.contiguous()
        out = torch.empty_like(a)
        with _on_device(a):
            _lib.launch('camli_sk_mix_fwd', lib.camli_sk_mix_fwd, a.data_ptr(), b.data_ptr(), w.data_ptr(),
                        out.data_ptr(), bs, c, p, _stream_ptr(a), work=(12.0 * bs * c * p, 'B'))
        # a / b gradients are deferred to the pool node only if that node is certain to run: w is
        # differentiable (so the gate, hence the pooled vector, is on the backward path)
        state.deferred = bool(ctx.needs_input_grad[2] and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        ctx.state = state
        ctx.save_for_backward(a, b, w)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        lib = _lib.load()
        a, b, w = ctx.saved_tensors
        bs, c, p = _rows(a)
        g = g.contiguous().float()
        gw = None
        with _on_device(g):
            if ctx.needs_input_grad[2]:
                gw = torch.empty_like(w)
                _lib.launch('camli_sk_mix_bwd_w', lib.camli_sk_mix_bwd_w, g.data_ptr(), a.data_ptr(), b.data_ptr(),
                            gw.data_ptr(), bs, c, p, _stream_ptr(g), work=(12.0 * bs * c * p, 'B'))
            if ctx.state.deferred:
                ctx.state.g, ctx.state.w = g, w
                return None, None, gw, None
            ga, gb = torch.empty_like(a), torch.empty_like(b)
            _lib.launch('camli_sk_mix_bwd_x', lib.camli_sk_mix_bwd_x, g.data_ptr(), w.data_ptr(), None,
                        ga.data_ptr(), gb.data_ptr(), bs, c, p, _stream_ptr(g), work=(12.0 * bs * c * p, 'B'))
        return ga, gb, gw, None


class _SkGate(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, s, wmid, wout):
        lib = _lib.load()
        bs, c = s.shape
        r = wmid.shape[0]
        s = s.contiguous()
        m = torch.empty((bs, r), dtype=torch.float32, device=s.device)
        z = torch.empty((bs, 2 * c), dtype=torch.float32, device=s.device)
        w = torch.empty((bs, c, 2), dtype=torch.float32, device=s.device)
        with _on_device(s):
            _lib.launch('camli_sk_gate_fwd', lib.camli_sk_gate_fwd, s.data_ptr(), wmid.data_ptr(), wout.data_ptr(),
                        m.data_ptr(), z.data_ptr(), w.data_ptr(), bs, c, r, _stream_ptr(s),
                        work=(4.0 * (bs * (4 * c + r) + 3 * c * r), 'B'))
        ctx.save_for_backward(s, m, z, w, wmid, wout)
        ctx.params = [_runtime.deferral_target(t) for t in (wmid, wout)]
        return w

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gw):
        lib = _lib.load()
        s, m, z, w, wmid, wout = ctx.saved_tensors
        bs, c = s.shape
        r = wmid.shape[0]
        gw = gw.contiguous().float()
        gs = torch.empty_like(s)
        grads = []
        deferred = []
        for param, like in zip(ctx.params, (wmid, wout)):
            if param is not None:
                grads.append(_runtime.PARAM_GRADS.slot(param, lambda like=like: torch.zeros_like(like), False))
                deferred.append(True)
            else:
                grads.append(torch.zeros_like(like))
                deferred.append(False)
        with _on_device(s):
            _lib.launch('camli_sk_gate_bwd', lib.camli_sk_gate_bwd, gw.data_ptr(), s.data_ptr(), m.data_ptr(), z.data_ptr(),
                        w.data_ptr(), wmid.data_ptr(), wout.data_ptr(), gs.data_ptr(), grads[0].data_ptr(),
                        grads[1].data_ptr(), bs, c, r, _stream_ptr(s), work=(4.0 * (bs * (7 * c + r) + 9 * c * r), 'B'))
        return gs, (None if deferred[0] else grads[0]), (None if deferred[1] else grads[1])


def sk_gate(s, wmid, wout):
    """[B,C] pooled vector -> [B,C,2] branch weights (two bias-free Linear layers, ReLU, Sigmoid, softmax over
    the pair) in one launch each way."""
    _require_cuda('sk_gate', s, wmid, wout)
    assert s.dim() == 2 and wmid.shape[1] == s.shape[1] and wout.shape == (2 * s.shape[1], wmid.shape[0])
    return _SkGate.apply(s.float(), wmid.float().contiguous(), wout.float().contiguous())


def sk_pool(a, b, state):
    """mean over the positions of (a + b): [B,C,...] x2 -> [B,C]"""
    _require_cuda('sk_pool', a, b)
    assert a.shape == b.shape and a.dim() >= 3
    return _SkPool.apply(a.float().contiguous(), b.float().contiguous(), state)


def sk_mix(a, b, w, state):
    """a * w[...,0] + b * w[...,1] with per-(batch, channel) weights w [B,C,2]"""
    _require_cuda('sk_mix', a, b, w)
    assert a.shape == b.shape and w.shape == (a.shape[0], a.shape[1], 2)
    return _SkMix.apply(a.float().contiguous(), b.float().contiguous(), w.float(), state)


# ------------------------------------------------------------------------------------------------
# nearest-point feature x score of FusionAwareInterp (models/clfm.py:70-76, k = 1)
# ------------------------------------------------------------------------------------------------
class _GatherScale(torch.autograd.Function):
    @staticmethod
    def _run(data, scale, idx):
        lib = _lib.load()
        bs, c, m = data.shape
        p = scale.shape[2]
        out = torch.empty_like(scale)
        with _on_device(data):
            _lib.launch('camli_gather_scale_fwd', lib.camli_gather_scale_fwd, data.data_ptr(), scale.data_ptr(),
                        idx.data_ptr(), out.data_ptr(), bs, c, m, p, _stream_ptr(data),
                        work=(12.0 * bs * c * p + 8.0 * bs * p, 'B'))
        return out

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, scale, data, idx):
        scale = scale.contiguous()
        ctx.save_for_backward(data, idx)
        return _GatherScale._run(data, scale, idx)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        data, idx = ctx.saved_tensors
        return _GatherScale._run(data, gout.contiguous().float(), idx), None, None


def gather_scale(data, scale, idx):
    """scale [B,C,P] * data[B,C,M] gathered at idx [B,P] -> [B,C,P]; differentiable wrt scale only."""
    _require_cuda('gather_scale', data, scale, idx)
    assert not data.requires_grad and idx.dtype == torch.int64 and idx.shape == (scale.shape[0], scale.shape[2])
    assert data.shape[:2] == scale.shape[:2]
    return _GatherScale.apply(scale.float(), data.float().contiguous(), idx.contiguous())


# ------------------------------------------------------------------------------------------------
# masked end-point-error sums of the sequence losses (models/losses.py:64-119)
# ------------------------------------------------------------------------------------------------
class _MaskedL2Sums(torch.autograd.Function):
    """[sum over the mask of ||pred_i - target||_2 for every iterate i]: one kernel per iterate each way."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, target, n_channels, *preds):
        lib = _lib.load()
        bs, tc = target.shape[0], target.shape[1]
        p = target.numel() // (bs * tc)
        preds = [q.contiguous() for q in preds]
        sums = torch.zeros(len(preds), dtype=torch.float32, device=target.device)
        with _on_device(target):
            for i, q in enumerate(preds):
                _lib.launch('camli_masked_l2_fwd', lib.camli_masked_l2_fwd, q.data_ptr(), target.data_ptr(), tc,
                            sums.data_ptr() + 4 * i, bs, n_channels, p, _stream_ptr(target),
                            work=(4.0 * bs * p * (n_channels + tc), 'B'))
        ctx.save_for_backward(target, *preds)
        ctx.n_channels = n_channels
        return sums

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gsums):
        lib = _lib.load()
        target, *preds = ctx.saved_tensors
        bs, tc = target.shape[0], target.shape[1]
        p = target.numel() // (bs * tc)
        gsums = gsums.contiguous().float()
        grads = []
        with _on_device(target):
            for i, q in enumerate(preds):
                if not ctx.needs_input_grad[2 + i]:
                    grads.append(None)
                    continue
                gq = torch.empty_like(q)
                _lib.launch('camli_masked_l2_bwd', lib.camli_masked_l2_bwd, q.data_ptr(), target.data_ptr(), tc,
                            gsums.data_ptr() + 4 * i, gq.data_ptr(), bs, ctx.n_channels, p, _stream_ptr(target),
                            work=(4.0 * bs * p * (2 * ctx.n_channels + tc), 'B'))
                grads.append(gq)
        return (None, None, *grads)


def masked_l2_sums(preds, target, n_channels):
    """preds: list of [B,C,...], target [B,C or C+1,...] -> [len(preds)] masked sums of the per-position L2 error."""
    _require_cuda('masked_l2_sums', target, *preds)
    assert n_channels in (2, 3) and target.shape[1] in (n_channels, n_channels + 1) and not target.requires_grad
    assert all(q.shape[1] == n_channels and q.shape[2:] == target.shape[2:] for q in preds)
    return _MaskedL2Sums.apply(target.float().contiguous(), n_channels, *[q.float() for q in preds])


# ------------------------------------------------------------------------------------------------
# flow read-out of the inverse-depth-scaling wrapper (models/ids.py:36-67, camliraft.py:108-110)
# ------------------------------------------------------------------------------------------------
class _IdsFlow(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, flow, pc1, origin, f, cx, cy, consts):
        lib = _lib.load()
        bs, _, n = pc1.shape
        flow = flow.contiguous()
        out = torch.empty_like(pc1)
        with _on_device(pc1):
            _lib.launch('camli_ids_flow_fwd', lib.camli_ids_flow_fwd, pc1.data_ptr(), flow.data_ptr(), origin.data_ptr(),
                        f.data_ptr(), cx.data_ptr(), cy.data_ptr(), out.data_ptr(), *consts, bs, n, _stream_ptr(pc1),
                        work=(48.0 * bs * n, 'B'))
        ctx.save_for_backward(flow, pc1, f, cx, cy)
        ctx.consts = consts
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        flow, pc1, f, cx, cy = ctx.saved_tensors
        bs, _, n = pc1.shape
        gout = gout.contiguous().float()
        gflow = torch.empty_like(flow)
        with _on_device(pc1):
            _lib.launch('camli_ids_flow_bwd', lib.camli_ids_flow_bwd, pc1.data_ptr(), flow.data_ptr(), gout.data_ptr(),
                        f.data_ptr(), cx.data_ptr(), cy.data_ptr(), gflow.data_ptr(), *ctx.consts, bs, n,
                        _stream_ptr(pc1), work=(48.0 * bs * n, 'B'))
        return gflow, None, None, None, None, None, None


def ids_flow(flow, pc1, origin, persp, paral):
    """paral2persp(pc1 + flow) - origin in one launch (differentiable wrt flow); persp / paral as in
    cores.geometry.paral2persp."""
    _require_cuda('ids_flow', flow, pc1, origin)
    assert not (pc1.requires_grad or origin.requires_grad)
    ratio_w = (paral['sensor_w'] - 1) / (persp['sensor_w'] - 1)
    ratio_h = (paral['sensor_h'] - 1) / (persp['sensor_h'] - 1)
    consts = (float(ratio_w), float(ratio_h), float(min(ratio_w, ratio_h)), float((paral['sensor_w'] - 1) / 2),
              float((paral['sensor_h'] - 1) / 2))
    return _IdsFlow.apply(flow.float(), pc1.float().contiguous(), origin.float().contiguous(),
                          persp['f'].float().contiguous(), persp['cx'].float().contiguous(),
                          persp['cy'].float().contiguous(), consts)


# ------------------------------------------------------------------------------------------------
# bilinear sampling of image features at projected points (models/utils.py:262-269)
# ------------------------------------------------------------------------------------------------
def bilinear_sample(feat_2d, uv):
    """feat_2d [B,C,H,W], uv [B,2,N] pixel coordinates -> [B,C,N] fp32, no autograd (the reference detaches
    both the input and the output of this op on the fusion path, clfm.py:187-190)."""
    _require_cuda('bilinear_sample', feat_2d, uv)
    assert not (feat_2d.requires_grad or uv.requires_grad), 'forward-only op: detach the inputs'
    lib = _lib.load()
    feat_2d, uv = feat_2d.float().contiguous(), uv.float().contiguous()
    bs, c, h, w = feat_2d.shape
    n = uv.shape[2]
    out = torch.empty((bs, c, n), dtype=torch.float32, device=feat_2d.device)
    with _on_device(feat_2d):
        _lib.launch('camli_bilinear_sample_fwd', lib.camli_bilinear_sample_fwd, feat_2d.data_ptr(), uv.data_ptr(),
                    out.data_ptr(), bs, c, h, w, n, _stream_ptr(feat_2d),
                    work=(4.0 * bs * c * n * 5 + 8.0 * bs * n, 'B'))
    return out


# ------------------------------------------------------------------------------------------------
# neighbour-weight network of PointConvDW on the matrix cores (models/point_conv.py:110-121)
# ------------------------------------------------------------------------------------------------
class _WeightNet(torch.autograd.Function):
    """weight_net(knn_offset) in one launch; the backward is one launch as well: it recomputes the hidden
    layers and produces the gradients of all six parameters (every contraction on the matrix cores)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, xyz, centres, knn_indices, k, w1, b1, w2, b2, w3, b3, k_major=False):
        lib = _lib.load()
        bs, _, m = xyz.shape
        n, c = centres.shape[2], w3.shape[0]
        params = [t.reshape(t.shape[0], -1).contiguous() for t in (w1, b1, w2, b2, w3, b3)]
        out = torch.empty((bs, c, k, n) if k_major else (bs, c, n, k), dtype=torch.float32, device=xyz.device)
        ctx.k_major = bool(k_major)
        with _on_device(xyz):
            _lib.launch('camli_weightnet_fwd', lib.camli_weightnet_fwd, xyz.data_ptr(), centres.data_ptr(),
                        knn_indices.data_ptr(), knn_indices.stride(1), *[t.data_ptr() for t in params],
                        out.data_ptr(), bs, c, m, n, k, int(ctx.k_major), _stream_ptr(xyz),
                        work=(4.0 * bs * c * n * k + 8.0 * bs * n * k + 12.0 * bs * (m + n), 'B'),
                        flop=2.0 * bs * n * k * (24 + 256 + 32 * c))
        ctx.save_for_backward(xyz, centres, knn_indices, *params)
        ctx.k = k
        ctx.shapes = [t.shape for t in (w1, b1, w2, b2, w3, b3)]
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        xyz, centres, knn_indices, *params = ctx.saved_tensors
        k = ctx.k
        bs, _, m = xyz.shape
        n, c = centres.shape[2], params[4].shape[0]
        gout = gout.contiguous().float()
        sizes = [t.numel() for t in params]
        grads = list(torch.split(torch.empty(sum(sizes), dtype=torch.float32, device=xyz.device), sizes))
        ws_bytes = lib.camli_weightnet_bwd_workspace_bytes(c)
        workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=xyz.device)
        with _on_device(xyz):
            _lib.launch('camli_weightnet_bwd', lib.camli_weightnet_bwd, xyz.data_ptr(), centres.data_ptr(),
                        knn_indices.data_ptr(), knn_indices.stride(1), *[t.data_ptr() for t in params],
                        gout.data_ptr(), *[g.data_ptr() for g in grads], workspace.data_ptr(), ws_bytes,
                        bs, c, m, n, k, int(ctx.k_major), _stream_ptr(xyz),
                        work=(4.0 * bs * c * n * k + 8.0 * bs * n * k + 12.0 * bs * (m + n), 'B'),
                        flop=2.0 * bs * n * k * (3 * 32 * c + 2 * 32 * 32))
        grads = [g.view(shape) for g, shape in zip(grads, ctx.shapes)]
        return (None, None, None, None, *grads, None)


def weightnet_supported(mlp, c_out):
    """The fused kernel implements exactly MLP2d(3, [8, 32, C<=128], norm=None, act='relu')."""
    convs = getattr(mlp, 'convs', None)
    if convs is None or len(convs) != 3 or c_out > 128:
        return False
    dims = [(cv.conv_fn.in_channels, cv.conv_fn.out_channels) for cv in convs]
    plain = all(isinstance(cv.norm_fn, torch.nn.Identity) and isinstance(cv.act_fn, torch.nn.ReLU)
                and cv.conv_fn.bias is not None for cv in convs)
    return plain and dims == [(3, 8), (8, 32), (32, c_out)]


def weightnet_hidden8(xyz, centres, knn_indices, k, mlp):
    """weight_net = MLP2d(3, [8, 8, C], relu) of the PointPWC cost volume (camlipwc_l_core.py:45-46) through the same
    matrix-core kernel: the 8-wide second layer is zero-padded to the kernel's 32 hidden units (padded units are
    relu(0) = 0 and meet zero columns of the last layer, so values and gradients are unchanged; torch differentiates
    the padding).  C > 128 is evaluated in two channel halves."""
    _require_cuda('weightnet_hidden8', xyz, centres, knn_indices)
    convs = mlp.convs
    w1, b1 = convs[0].conv_fn.weight, convs[0].conv_fn.bias
    w2, b2 = convs[1].conv_fn.weight, convs[1].conv_fn.bias
    w3, b3 = convs[2].conv_fn.weight, convs[2].conv_fn.bias
    assert w1.shape[:2] == (8, 3) and w2.shape[:2] == (8, 8) and w3.shape[1] == 8
    pad = torch.nn.functional.pad
    w2p = pad(w2.reshape(8, 8), (0, 0, 0, 24))            # [32, 8]
    b2p = pad(b2, (0, 24))                                # [32]
    w3p = pad(w3.reshape(w3.shape[0], 8), (0, 24))        # [C, 32]
    args = (xyz.float().contiguous(), centres.float().contiguous(), knn_indices, k)
    c = w3.shape[0]
    if c <= 128:
        return _WeightNet.apply(*args, w1, b1, w2p, b2p, w3p, b3, False)
    half = c // 2
    return torch.cat([_WeightNet.apply(*args, w1, b1, w2p, b2p, w3p[:half], b3[:half], False),
                      _WeightNet.apply(*args, w1, b1, w2p, b2p, w3p[half:], b3[half:], False)], dim=1)


def weightnet_hidden8_supported(mlp):
    convs = getattr(mlp, 'convs', None)
    if convs is None or len(convs) != 3:
        return False
    dims = [(cv.conv_fn.in_channels, cv.conv_fn.out_channels) for cv in convs]
    plain = all(isinstance(cv.norm_fn, torch.nn.Identity) and isinstance(cv.act_fn, torch.nn.ReLU)
                and cv.conv_fn.bias is not None for cv in convs)
    return plain and dims[0] == (3, 8) and dims[1] == (8, 8) and dims[2][0] == 8 and dims[2][1] <= 256


def dw_k_major_ok(k, m, feat_like=None):
    """Shapes the k-major set-conv forward covers (camli_pointconv_dw_fwd, k_major = 1)."""
    return k in (4, 8, 16, 32) and m % 4 == 0 and m <= 8192


def weightnet(xyz, centres, knn_indices, k, mlp, k_major=False):
    """xyz [B,3,M], centres [B,3,N], knn_indices int64 [B,N,>=k] -> weight_net(xyz[knn] - centre) [B,C,N,k], or
    [B,C,k,N] with ``k_major`` (the layout the set-conv forward streams without an LDS transposition)."""
    _require_cuda('weightnet', xyz, centres, knn_indices)
    assert knn_indices.dtype == torch.int64 and knn_indices.stride(2) == 1 and knn_indices.shape[2] >= k
    assert knn_indices.stride(0) == knn_indices.shape[1] * knn_indices.stride(1)
    assert not (xyz.requires_grad or centres.requires_grad), 'coordinates are constants of this path'
    convs = mlp.convs
    return _WeightNet.apply(xyz.float().contiguous(), centres.float().contiguous(), knn_indices, k,
                            convs[0].conv_fn.weight, convs[0].conv_fn.bias, convs[1].conv_fn.weight,
                            convs[1].conv_fn.bias, convs[2].conv_fn.weight, convs[2].conv_fn.bias, k_major)



# ------------------------------------------------------------------------------------------------
# inverse index maps: the adjoint of a gather without atomics
# ------------------------------------------------------------------------------------------------
_inverse_maps = {}      # key -> (index tensor kept alive so its address cannot be recycled, order, offsets); small LRU


def inverse_map(idx_flat, m):
    """idx_flat int64 [B, I] with values in [0, m) -> (order int32 [B*I], offsets int32 [B*m + 1]): the flat
    positions b*I + i sorted (stably) by (b, idx) and the CSR bounds of every (b, source) segment.  This is what
    torch's index_put_(accumulate=True) recomputes with a device-wide sort on EVERY backward call; here it is
    built once per index tensor (neighbour tables are shared by the modules and iterations of a pass) and turns
    scatter-adds into coalesced segment sums."""
    key = (idx_flat.data_ptr(), tuple(idx_flat.shape), tuple(idx_flat.stride()), m, idx_flat._version)
    hit = _inverse_maps.pop(key, None)
    if hit is None:
        b, i = idx_flat.shape
        assert b * i < 2 ** 31 and b * m < 2 ** 31
        keys = (idx_flat + torch.arange(b, device=idx_flat.device).view(b, 1) * m).reshape(-1)
        sorted_keys, order = torch.sort(keys, stable=True)
        offsets = torch.searchsorted(sorted_keys, torch.arange(b * m + 1, device=idx_flat.device))
        hit = (idx_flat, order.to(torch.int32), offsets.to(torch.int32))
        while len(_inverse_maps) >= 16:
            _inverse_maps.pop(next(iter(_inverse_maps)))
    _inverse_maps[key] = hit          # (re)insert as most recent
    return hit[1], hit[2]


# ------------------------------------------------------------------------------------------------
# gather / interpolation / point cost-volume lookup (models/utils.py, models/camliraft_l_core.py)
# ------------------------------------------------------------------------------------------------
class _GatherCF(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, data, idx_flat):
        lib = _lib.load()
        b, c, m = data.shape
        i = idx_flat.shape[1]
        out = torch.empty((b, c, i), dtype=torch.float32, device=data.device)
        with _on_device(data):
            _lib.launch('camli_gather_cf_fwd', lib.camli_gather_cf_fwd, data.data_ptr(), idx_flat.data_ptr(),
                        out.data_ptr(), b, c, m, i, _stream_ptr(data),
                        work=(8.0 * b * c * i + 8.0 * b * i, 'B'))
        ctx.save_for_backward(idx_flat)
        ctx.m = m
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        (idx_flat,) = ctx.saved_tensors
        gout = gout.contiguous().float()
        b, c, i = gout.shape
        # gather-side adjoint through the inverse map: no atomics, every output written once
        order, offsets = inverse_map(idx_flat, ctx.m)
        gdata = torch.empty((b, c, ctx.m), dtype=torch.float32, device=gout.device)
        with _on_device(gout):
            _lib.launch('camli_gather_cf_bwd', lib.camli_gather_cf_bwd_sorted, gout.data_ptr(), order.data_ptr(),
                        offsets.data_ptr(), gdata.data_ptr(), b, c, ctx.m, i, _stream_ptr(gout),
                        work=(4.0 * b * c * (i + ctx.m) + 4.0 * b * (i + ctx.m), 'B'))
        return gdata, None


def gather_points(data, indices):
    """batch_indexing, channel-first: data [B,C,M], indices [B,I1..Im] -> [B,C,I1..Im] (utils.py:61-83)."""
    _require_cuda('gather_points', data, indices)
    b, c = data.shape[:2]
    flat = indices.reshape(b, -1).to(torch.int64).contiguous()
    out = _GatherCF.apply(data.float().contiguous(), flat)
    return out.view([b, c] + list(indices.shape[1:]))


class _GatherCL(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, data, idx_flat):
        lib = _lib.load()
        b, m, c = data.shape
        i = idx_flat.shape[1]
        out = torch.empty((b, i, c), dtype=torch.float32, device=data.device)
        with _on_device(data):
            _lib.launch('camli_gather_cl_fwd', lib.camli_gather_cl_fwd, data.data_ptr(), idx_flat.data_ptr(),
                        out.data_ptr(), b, c, m, i, _stream_ptr(data), work=(8.0 * b * c * i + 8.0 * b * i, 'B'))
        ctx.save_for_backward(idx_flat)
        ctx.m = m
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        (idx_flat,) = ctx.saved_tensors
        gout = gout.contiguous().float()
        b, i, c = gout.shape
        order, offsets = inverse_map(idx_flat, ctx.m)
        gdata = torch.empty((b, ctx.m, c), dtype=torch.float32, device=gout.device)
        with _on_device(gout):
            _lib.launch('camli_gather_cl_bwd', lib.camli_gather_cl_bwd_sorted, gout.data_ptr(), order.data_ptr(),
                        offsets.data_ptr(), gdata.data_ptr(), b, c, ctx.m, i, _stream_ptr(gout),
                        work=(4.0 * b * c * (i + ctx.m) + 4.0 * b * (i + ctx.m), 'B'))
        return gdata, None


def gather_rows(data, indices):
    """batch_indexing, channel-last: data [B,M,C] or [B,M], indices [B,I1..Im] -> [B,I1..Im,C] or [B,I1..Im]
    (utils.py:85-104)."""
    _require_cuda('gather_rows', data, indices)
    b = data.shape[0]
    flat = indices.reshape(b, -1).to(torch.int64).contiguous()
    rows = data.float().contiguous()
    if data.dim() == 2:
        return _GatherCL.apply(rows.unsqueeze(-1), flat).view([b] + list(indices.shape[1:]))
    return _GatherCL.apply(rows, flat).view([b] + list(indices.shape[1:]) + [data.shape[2]])


_INTERP_SORTED = os.environ.get('CAMLI_INTERP_SORTED', '1') != '0'      # 0: the float-atomic adjoint of the interpolation (A/B)
_interp_weights = {}        # key -> (tensors kept alive, knn_flat [B, Nq*k] int64, w [B, Nq, k]); small LRU like _inverse_maps


def _interp_geometry(in_xyz, q_xyz, knn, k):
    """(offsets, q_sorted, w_sorted) of an interpolation between two clouds, for the atomic-free adjoint: the CSR bounds of the
    neighbour table's inverse map and, in segment order, the query and the weight of every (query, slot) pair.  Computed once per
    (clouds, table) -- the GRU loops interpolate between the same clouds every iteration."""
    key = (in_xyz.data_ptr(), in_xyz._version, q_xyz.data_ptr(), q_xyz._version, knn.data_ptr(), knn._version,
           tuple(knn.shape), tuple(knn.stride()), k)
    hit = _interp_weights.pop(key, None)
    if hit is None:
        lib = _lib.load()
        b, _, m = in_xyz.shape
        nq = q_xyz.shape[2]
        w = torch.empty((b, nq, k), dtype=torch.float32, device=in_xyz.device)
        with _on_device(in_xyz):
            _lib.launch('camli_knn_interp_weights', lib.camli_knn_interp_weights, in_xyz.data_ptr(), q_xyz.data_ptr(),
                        knn.data_ptr(), knn.stride(1), w.data_ptr(), b, m, nq, k, _stream_ptr(in_xyz),
                        work=(8.0 * b * nq * k + 12.0 * b * nq * (1 + k) + 4.0 * b * nq * k, 'B'))
        order, offsets = inverse_map(knn[:, :, :k].reshape(b, nq * k).contiguous(), m)
        order = order.long()
        q_sorted = ((order % (nq * k)) // k).to(torch.int32)
        hit = ((in_xyz, q_xyz, knn), offsets, q_sorted, w.reshape(-1)[order].contiguous())
        while len(_interp_weights) >= 8:
            _interp_weights.pop(next(iter(_interp_weights)))
    _interp_weights[key] = hit
    return hit[1], hit[2], hit[3]


class _KnnInterp(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, in_xyz, feat, q_xyz, knn, k, invariant=False):
        lib = _lib.load()
        b, c, m = feat.shape
        nq = q_xyz.shape[2]
        ctx.invariant = invariant
        out = torch.empty((b, c, nq), dtype=torch.float32, device=feat.device)
        with _on_device(feat):
            _lib.launch('camli_knn_interp_fwd', lib.camli_knn_interp_fwd, in_xyz.data_ptr(), feat.data_ptr(),
                        q_xyz.data_ptr(), knn.data_ptr(), knn.stride(1), out.data_ptr(), b, c, m, nq, k,
                        _stream_ptr(feat),
                        work=(4.0 * b * c * nq * (1 + k) + 8.0 * b * nq * k + 12.0 * b * nq * (1 + k), 'B'))
        coords = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        ctx.save_for_backward(in_xyz, q_xyz, knn, *([feat] if coords else []))
        ctx.dims = (b, c, m, nq, k)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        in_xyz, q_xyz, knn = ctx.saved_tensors[:3]
        b, c, m, nq, k = ctx.dims
        gout = gout.contiguous().float()
        gfeat = g_in = g_q = None
        with _on_device(gout):
            if ctx.needs_input_grad[1] and _INTERP_SORTED and ctx.invariant and b * nq * k < 2 ** 31 \
                    and not (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]):
                # round 5: no atomics -- the weights once per pair of clouds, a segment sum per input point over the inverse map
                # of the neighbour table (both cached across the iterations of a pass), fixed summation order, no zero-fill.
                # Only where the caller declares the geometry invariant across calls (the GRU loops' up-sampling of the
                # 3-D predictions): with clouds that move every call the sort behind the inverse map would cost more than the
                # atomics do
                offsets, q_sorted, w_sorted = _interp_geometry(in_xyz, q_xyz, knn, k)
                gfeat = torch.empty((b, c, m), dtype=torch.float32, device=gout.device)
                _lib.launch('camli_knn_interp_bwd', lib.camli_knn_interp_bwd_sorted, gout.data_ptr(), w_sorted.data_ptr(),
                            q_sorted.data_ptr(), offsets.data_ptr(), gfeat.data_ptr(), b, c, m, nq, _stream_ptr(gout),
                            work=(4.0 * b * c * (nq * k + m) + 8.0 * b * nq * k + 4.0 * b * m, 'B'))
            elif ctx.needs_input_grad[1]:
                gfeat = torch.zeros((b, c, m), dtype=torch.float32, device=gout.device)
                _lib.launch('camli_knn_interp_bwd', lib.camli_knn_interp_bwd, in_xyz.data_ptr(), gout.data_ptr(),
                            q_xyz.data_ptr(), knn.data_ptr(), knn.stride(1), gfeat.data_ptr(), b, c, m, nq, k,
                            _stream_ptr(gout),
                            work=(4.0 * b * c * nq * (1 + 2 * k) + 8.0 * b * nq * k + 12.0 * b * nq * (1 + k), 'B'))
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
                # CamLiPWC back-warps with a live flow (camlipwc_core.py:172-179): the coordinates carry a gradient
                feat = ctx.saved_tensors[3]
                g_in = torch.zeros_like(in_xyz) if ctx.needs_input_grad[0] else None
                g_q = torch.empty_like(q_xyz) if ctx.needs_input_grad[2] else None
                _lib.launch('camli_knn_interp_bwd_xyz', lib.camli_knn_interp_bwd_xyz, in_xyz.data_ptr(), feat.data_ptr(),
                            gout.data_ptr(), q_xyz.data_ptr(), knn.data_ptr(), knn.stride(1),
                            g_in.data_ptr() if g_in is not None else None, g_q.data_ptr() if g_q is not None else None,
                            b, c, m, nq, k, _stream_ptr(gout),
                            work=(4.0 * b * c * nq * (1 + k) + 8.0 * b * nq * k + 12.0 * b * nq * (2 + 2 * k), 'B'))
        return g_in, gfeat, g_q, None, None, None


def knn_interpolate(input_xyz, input_features, query_xyz, knn_indices, k, invariant=False):
    """IDW interpolation given the k nearest inputs per query (utils.py:138-146); differentiable wrt the
    features and, when they require it, both coordinate sets.  ``invariant``: the caller passes the SAME clouds and neighbour
    table on every call of a pass (the feature adjoint then runs atomic-free on per-pass geometry)."""
    _require_cuda('knn_interpolate', input_xyz, input_features, query_xyz, knn_indices)
    return _KnnInterp.apply(input_xyz.float().contiguous(), input_features.float().contiguous(),
                            query_xyz.float().contiguous(), knn_indices, k, invariant)


class _Corr3DGather(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, cost, xyz1, xyz2, knn):
        lib = _lib.load()
        b, n, m = cost.shape
        k = knn.shape[2]
        out = torch.empty((b, 4, n, k), dtype=torch.float32, device=cost.device)
        with _on_device(cost):
            _lib.launch('camli_corr3d_gather_fwd', lib.camli_corr3d_gather_fwd, xyz1.data_ptr(), xyz2.data_ptr(),
                        cost.data_ptr(), knn.data_ptr(), out.data_ptr(), b, n, m, k, _stream_ptr(cost),
                        work=(b * n * k * (8.0 + 16.0 + 4.0 + 12.0), 'B'))
        ctx.save_for_backward(knn)
        ctx.dims = (b, n, m, k)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        (knn,) = ctx.saved_tensors
        b, n, m, k = ctx.dims
        gout = gout.contiguous().float()
        gcost = torch.zeros((b, n, m), dtype=torch.float32, device=gout.device)
        with _on_device(gout):
            _lib.launch('camli_corr3d_gather_bwd', lib.camli_corr3d_gather_bwd, gout.data_ptr(), knn.data_ptr(),
                        gcost.data_ptr(), b, n, m, k, _stream_ptr(gout),
                        work=(b * n * k * (8.0 + 4.0 + 8.0), 'B'))
        return gcost, None, None, None


def corr3d_lookup_input(cost_volume, xyz1, xyz2, knn_indices):
    """[B,N,M] cost volume + coordinates + cross KNN -> the [B,4,N,k] tensor (dxyz, cost entry) the
    cost MLP consumes (camliraft_l_core.py:62-76) in one launch; gradient to the cost volume only."""
    _require_cuda('corr3d_lookup_input', cost_volume, xyz1, xyz2, knn_indices)
    assert not xyz1.requires_grad and not xyz2.requires_grad
    return _Corr3DGather.apply(cost_volume.float().contiguous(), xyz1.float().contiguous(),
                               xyz2.float().contiguous(), knn_indices.contiguous())


class _PointVolumes(torch.autograd.Function):
    """V_l = f1^T . f2_l / C for every level of the point cost-volume pyramid, one launch per level forward and the
    grouped adjoint GEMMs backward (camli_allpairs_build_fwd / _bwd, the fp32 matrix-core kernel of the 2-D all-pairs
    volume with scale 1/C).  The reference builds level 0 with torch.bmm + a division over the volume and every coarser
    level by gathering and averaging VOLUME columns (camliraft_l_core.py:51-60); the KNN mean acts on the target point
    only and is linear, so level l is the same GEMM against the KNN-averaged target FEATURES (equal up to fp32 summation
    order), each volume element is written once, and the adjoint never scatters into volume-sized tensors."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, feat1, *feat2_levels):
        lib = _lib.load()
        feat1 = feat1.contiguous()
        feat2_levels = [t.contiguous() for t in feat2_levels]
        b, c, n = feat1.shape
        sizes = [t.shape[2] for t in feat2_levels]
        vols = [torch.empty((b, n, m), dtype=torch.float32, device=feat1.device) for m in sizes]
        total = sum(sizes)
        with _on_device(feat1):
            _lib.launch('camli_allpairs_build_fwd', lib.camli_allpairs_build_fwd, feat1.data_ptr(), _ptr_array(feat2_levels),
                        _ptr_array(vols), (ctypes.c_int * len(sizes))(*sizes), len(sizes), b, c, n, 1.0 / c, _stream_ptr(feat1),
                        work=(4.0 * b * n * total + 4.0 * b * c * (n + total), 'B'), flop=2.0 * b * n * total * c)
        ctx.save_for_backward(feat1, *feat2_levels)
        return tuple(vols)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, *gvols):
        lib = _lib.load()
        feat1, *feat2_levels = ctx.saved_tensors
        b, c, n = feat1.shape
        sizes = [t.shape[2] for t in feat2_levels]
        gvols = [torch.zeros((b, n, m), dtype=torch.float32, device=feat1.device) if g is None else g.contiguous().float()
                 for g, m in zip(gvols, sizes)]
        g1 = torch.empty_like(feat1)
        g2_levels = [torch.empty_like(t) for t in feat2_levels]
        total = sum(sizes)
        with _on_device(feat1):
            _lib.launch('camli_allpairs_build_bwd', lib.camli_allpairs_build_bwd, feat1.data_ptr(), _ptr_array(feat2_levels),
                        _ptr_array(gvols), (ctypes.c_int * len(sizes))(*sizes), len(sizes), g1.data_ptr(), _ptr_array(g2_levels),
                        b, c, n, 1.0 / c, _stream_ptr(feat1),
                        work=(4.0 * b * n * total + 4.0 * b * c * 2 * (n + total), 'B'), flop=4.0 * b * n * total * c)
        return (g1, *g2_levels)


def point_volume_pyramid(feat1, feat2, parents):
    """Correlation3D.build_cost_volume_pyramid (camliraft_l_core.py:51-60) on the matrix cores: feat1 [B,C,N],
    feat2 [B,C,M0], parents[l] int64 [B,M_{l+1},k] = the k nearest level-l targets of every level-(l+1) target.
    Returns the list of [B,N,M_l] volumes."""
    _require_cuda('point_volume_pyramid', feat1, feat2)
    levels = [feat2.float()]
    for idx in parents:
        levels.append(gather_points(levels[-1], idx).mean(dim=-1))          # [B,C,M_l]: tiny next to the volumes
    return list(_PointVolumes.apply(feat1.float(), *levels))


class Corr3DPyramid:
    """The point cost-volume pyramid of one pass (camliraft_l_core.py:51-60) for the multi-level lookup: `levels` are
    the [B,N,M_l] volumes (still differentiable functions of the features), `token` ties every lookup to the pass, and
    the lookups' backward ADD into one persistent gradient volume per level (allocated + zeroed by the first one).
    The token node's backward then hands those totals to autograd as the gradients of the level tensors.  Without
    it every lookup of every GRU iteration returns zero-filled volume-sized gradients (250 MB per iteration at batch
    8) that autograd has to sum."""

    def __init__(self, levels):
        self.levels = [lvl.float().contiguous() for lvl in levels]
        self.sizes = [lvl.shape[2] for lvl in self.levels]
        self.grads = None
        self.token = _Corr3DToken.apply(self, *self.levels)


class _Corr3DToken(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pyr, *levels):
        ctx.pyr = weakref.ref(pyr)          # pyr.token is this node's output: no strong back-reference (cycle)
        ctx.meta = [(tuple(lvl.shape), lvl.device) for lvl in levels]
        return levels[0].new_zeros(1)

    @staticmethod
    def backward(ctx, _gtoken):
        pyr = ctx.pyr()
        grads = None
        if pyr is not None:
            grads, pyr.grads = pyr.grads, None
        if grads is None:
            grads = [torch.zeros(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.meta]
        return (None, *grads)


class _Corr3DLookupLevels(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, token, xyz1, xyz2, pyr, *knn_levels):
        lib = _lib.load()
        b, _, n = xyz1.shape
        m0, k, nl = xyz2.shape[2], knn_levels[0].shape[2], len(knn_levels)
        out = torch.empty((b, 4, n, nl * k), dtype=torch.float32, device=xyz1.device)
        sizes = (ctypes.c_int * nl)(*pyr.sizes)
        with _on_device(xyz1):
            _lib.launch('camli_corr3d_gather_fwd', lib.camli_corr3d_gather_levels_fwd, xyz1.data_ptr(), xyz2.data_ptr(),
                        _ptr_array(pyr.levels), _ptr_array(knn_levels), sizes, nl, out.data_ptr(), b, n, m0, k,
                        _stream_ptr(xyz1), work=(b * n * nl * k * (8.0 + 16.0 + 4.0 + 12.0), 'B'))
        ctx.save_for_backward(*knn_levels)
        ctx.pyr, ctx.dims = pyr, (b, n, m0, k, nl)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        knn_levels = ctx.saved_tensors
        pyr = ctx.pyr
        b, n, m0, k, nl = ctx.dims
        gout = gout.contiguous().float()
        if pyr.grads is None:
            pyr.grads = [torch.zeros_like(lvl) for lvl in pyr.levels]
        sizes = (ctypes.c_int * nl)(*pyr.sizes)
        with _on_device(gout):
            _lib.launch('camli_corr3d_gather_bwd', lib.camli_corr3d_gather_levels_bwd, gout.data_ptr(), _ptr_array(knn_levels),
                        _ptr_array(pyr.grads), sizes, nl, b, n, m0, k, _stream_ptr(gout),
                        work=(b * n * nl * k * (8.0 + 4.0 + 8.0), 'B'))
        return (None, None, None, None) + (None,) * nl


def corr3d_lookup_levels(pyr, xyz1, xyz2, knn_levels):
    """Multi-level input of the cost MLP, [B,4,N,L*k], for NESTED target levels (xyz2 [B,3,M0] = the level-0 cloud,
    level l = its first pyr.sizes[l] points); gradient to the cost volumes only (through pyr.token)."""
    _require_cuda('corr3d_lookup_levels', xyz1, xyz2, *knn_levels)
    assert not xyz1.requires_grad and not xyz2.requires_grad and len(knn_levels) == len(pyr.levels) <= 4
    assert all(kn.is_contiguous() and kn.dtype == torch.int64 for kn in knn_levels) and min(pyr.sizes) >= knn_levels[0].shape[2]
    return _Corr3DLookupLevels.apply(pyr.token, xyz1.float().contiguous(), xyz2.float().contiguous(), pyr, *knn_levels)


class _Corr3DCostMLP(torch.autograd.Function):
    """relu(W2 relu(W1 x + b1) + b2) summed over the k neighbours of every level, see csrc/hip/corr3dmlp.hip."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, lookup, w1, b1, w2, b2, levels):
        lib = _lib.load()
        b, _, n, lk = lookup.shape
        hidden, k = w2.shape[0], lk // levels
        out = torch.empty((b, levels * hidden, n), dtype=torch.float32, device=lookup.device)
        cols = float(b) * n * lk
        with _on_device(lookup):
            _lib.launch('camli_corr3d_mlp_fwd', lib.camli_corr3d_mlp_fwd, lookup.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                        w2.data_ptr(), b2.data_ptr(), out.data_ptr(), b, n, levels, k, hidden, _stream_ptr(lookup),
                        work=(16.0 * cols + 4.0 * out.numel(), 'B'), flop=2.0 * cols * (4 * hidden + hidden * hidden))
        ctx.save_for_backward(lookup, w1, b1, w2, b2)
        ctx.params = [_runtime.deferral_target(t) for t in (w1, b1, w2, b2)]
        ctx.levels = levels
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        lookup, w1, b1, w2, b2 = ctx.saved_tensors
        b, _, n, lk = lookup.shape
        hidden, k = w2.shape[0], lk // ctx.levels
        gout = gout.contiguous().float()
        # the kernel writes channel 3 only (the cost-volume entry; the coordinates are not differentiable on this path):
        # channels 0-2 are zeros, not uninitialised memory (anomaly detection, any future consumer of the gradient)
        glookup = torch.empty_like(lookup)
        glookup[:, :3].zero_()
        grads, deferred = [], []
        for param, like in zip(ctx.params, (w1, b1, w2, b2)):
            if param is not None:
                grads.append(_runtime.PARAM_GRADS.slot(param, lambda like=like: torch.zeros_like(like), False))
            else:
                grads.append(torch.zeros_like(like))
            deferred.append(param is not None)
        ws = torch.empty(lib.camli_corr3d_mlp_bwd_workspace_bytes(b, n) // 4, dtype=torch.float32, device=lookup.device)
        cols = float(b) * n * lk
        with _on_device(lookup):
            _lib.launch('camli_corr3d_mlp_bwd', lib.camli_corr3d_mlp_bwd, lookup.data_ptr(), gout.data_ptr(), w1.data_ptr(),
                        b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), glookup.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                        grads[2].data_ptr(), grads[3].data_ptr(), ws.data_ptr(), b, n, ctx.levels, k, hidden, _stream_ptr(lookup),
                        work=(20.0 * cols + 4.0 * gout.numel(), 'B'), flop=2.0 * cols * (2 * 4 * hidden + 3 * hidden * hidden))
        return (glookup,) + tuple(None if d else g for g, d in zip(grads, deferred)) + (None,)


def corr3d_cost_mlp_supported(lookup, convs, levels):
    """The fused cost MLP covers the reference's configuration: two bias + ReLU layers 4 -> 32 -> 32, 4 levels x 16
    neighbours, fp32 parameters, a point count that is a multiple of 8."""
    if len(convs) != 2 or lookup.dim() != 4 or lookup.shape[1] != 4 or lookup.shape[3] % levels:
        return False
    c1, c2 = convs
    if c1.bias is None or c2.bias is None or tuple(c1.weight.shape[:2]) != (c2.weight.shape[1], 4):
        return False
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (c1.weight, c1.bias, c2.weight, c2.bias)):
        return False
    return bool(_lib.load().camli_corr3d_mlp_supported(levels, lookup.shape[3] // levels, c2.weight.shape[0], lookup.shape[2])) \
        and c2.weight.shape[0] == c2.weight.shape[1]


def corr3d_cost_mlp(lookup, conv1, conv2, levels):
    """lookup [B,4,N,levels*k] -> [B, levels*hidden, N] (level-major channels): ``cost_mlp(lookup).sum(-1)`` of every level,
    camliraft_l_core.py:96-100, without the two [B,hidden,N,levels*k] activations."""
    _require_cuda('corr3d_cost_mlp', lookup)
    return _Corr3DCostMLP.apply(lookup.float().contiguous(), conv1.weight, conv1.bias, conv2.weight, conv2.bias, levels)


# ------------------------------------------------------------------------------------------------
# PointPWC learnable cost volume (models/camlipwc_l_core.py:53-106), see csrc/hip/pwc3d.hip
# ------------------------------------------------------------------------------------------------
def _gather_adjoint_sorted(lib, grad_nk, idx_flat, m):
    """grad_nk [B,C,I] -> sum into [B,C,m] through the inverse map of idx_flat [B,I] (no atomics)."""
    b, c, i = grad_nk.shape
    order, offsets = inverse_map(idx_flat, m)
    out = torch.empty((b, c, m), dtype=torch.float32, device=grad_nk.device)
    _lib.launch('camli_gather_cf_bwd', lib.camli_gather_cf_bwd_sorted, grad_nk.data_ptr(), order.data_ptr(),
                offsets.data_ptr(), out.data_ptr(), b, c, m, i, _stream_ptr(grad_nk),
                work=(4.0 * b * c * (i + m) + 4.0 * b * (i + m), 'B'))
    return out


class _Pwc3dPair(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, a, bm, e, idx, slope):
        lib = _lib.load()
        b, c, n = a.shape
        m, k = bm.shape[2], idx.shape[2]
        h1 = torch.empty((b, c, n, k), dtype=torch.float32, device=a.device)
        with _on_device(a):
            _lib.launch('camli_pwc3d_pair_fwd', lib.camli_pwc3d_pair_fwd, a.data_ptr(), bm.data_ptr(), e.data_ptr(),
                        idx.data_ptr(), h1.data_ptr(), b, c, m, n, k, slope, _stream_ptr(a),
                        work=(4.0 * b * c * n * (3 * k + 1) + 8.0 * b * n * k, 'B'))
        ctx.save_for_backward(h1, idx)
        ctx.slope, ctx.m = slope, m
        return h1

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gh1):
        lib = _lib.load()
        h1, idx = ctx.saved_tensors
        b, c, n, k = h1.shape
        gh1 = gh1.contiguous().float()
        gpre = torch.empty_like(h1)
        ga = torch.empty((b, c, n), dtype=torch.float32, device=h1.device)
        with _on_device(h1):
            _lib.launch('camli_pwc3d_pair_bwd', lib.camli_pwc3d_pair_bwd, gh1.data_ptr(), h1.data_ptr(), gpre.data_ptr(),
                        ga.data_ptr(), b, c, n, k, ctx.slope, _stream_ptr(h1), work=(4.0 * b * c * n * (3 * k + 1), 'B'))
            gbm = _gather_adjoint_sorted(lib, gpre.view(b, c, n * k), idx.view(b, n * k), ctx.m) if ctx.needs_input_grad[1] else None
        return ga, gbm, gpre, None, None


def pwc3d_pair(a, bm, e, idx, slope=0.1):
    """leaky_relu(a[:, :, n] + bm[:, :, idx[n, j]] + e[:, :, n, j]): the first cost-MLP layer of the PointPWC cost
    volume with the concatenation [f1 | f2_knn | dxyz] split by input block (a = W1a.f1, bm = W1b.f2, e = W1c.d + b1)."""
    _require_cuda('pwc3d_pair', a, bm, e, idx)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.shape[:2] == (a.shape[0], a.shape[2])
    assert e.shape == (a.shape[0], a.shape[1], a.shape[2], idx.shape[2]) and bm.shape[:2] == a.shape[:2]
    return _Pwc3dPair.apply(a.float().contiguous(), bm.float().contiguous(), e.float().contiguous(), idx, float(slope))


class _KSum(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, w, h):
        lib = _lib.load()
        b, c, n, k = w.shape
        out = torch.empty((b, c, n), dtype=torch.float32, device=w.device)
        with _on_device(w):
            _lib.launch('camli_ksum_fwd', lib.camli_ksum_fwd, w.data_ptr(), h.data_ptr(), out.data_ptr(), b, c, n, k,
                        _stream_ptr(w), work=(4.0 * b * c * n * (2 * k + 1), 'B'))
        ctx.save_for_backward(w, h)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        lib = _lib.load()
        w, h = ctx.saved_tensors
        b, c, n, k = w.shape
        g = g.contiguous().float()
        gw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        gh = torch.empty_like(h) if ctx.needs_input_grad[1] else None
        with _on_device(w):
            _lib.launch('camli_ksum_bwd', lib.camli_ksum_bwd, g.data_ptr(), w.data_ptr(), h.data_ptr(),
                        gw.data_ptr() if gw is not None else None, gh.data_ptr() if gh is not None else None,
                        b, c, n, k, _stream_ptr(w), work=(4.0 * b * c * n * (4 * k + 1), 'B'))
        return gw, gh


def ksum(w, h):
    """sum over the neighbour axis of w * h: [B,C,N,k] x [B,C,N,k] -> [B,C,N] (one pass each way)."""
    _require_cuda('ksum', w, h)
    assert w.shape == h.shape and w.dim() == 4
    return _KSum.apply(w.float().contiguous(), h.float().contiguous())


class _GatherWSum(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, w, feat, idx):
        lib = _lib.load()
        b, c, n, k = w.shape
        m = feat.shape[2]
        out = torch.empty((b, c, n), dtype=torch.float32, device=w.device)
        with _on_device(w):
            _lib.launch('camli_gather_wsum_fwd', lib.camli_gather_wsum_fwd, w.data_ptr(), feat.data_ptr(), idx.data_ptr(),
                        out.data_ptr(), b, c, m, n, k, _stream_ptr(w), work=(4.0 * b * c * n * (2 * k + 1) + 8.0 * b * n * k, 'B'))
        ctx.save_for_backward(w, feat, idx)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        lib = _lib.load()
        w, feat, idx = ctx.saved_tensors
        b, c, n, k = w.shape
        m = feat.shape[2]
        g = g.contiguous().float()
        gw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        t = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        gfeat = None
        with _on_device(w):
            _lib.launch('camli_gather_wsum_bwd', lib.camli_gather_wsum_bwd, g.data_ptr(), w.data_ptr(), feat.data_ptr(),
                        idx.data_ptr(), gw.data_ptr() if gw is not None else None, t.data_ptr() if t is not None else None,
                        b, c, m, n, k, _stream_ptr(w), work=(4.0 * b * c * n * (4 * k + 1) + 8.0 * b * n * k, 'B'))
            if t is not None:
                gfeat = _gather_adjoint_sorted(lib, t.view(b, c, n * k), idx.view(b, n * k), m)
        return gw, gfeat, None


def gather_wsum(w, feat, idx):
    """out[b,c,n] = sum_j w[b,c,n,j] * feat[b,c,idx[b,n,j]] (patch-to-patch aggregation, camlipwc_l_core.py:97-101)."""
    _require_cuda('gather_wsum', w, feat, idx)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.shape == (w.shape[0], w.shape[2], w.shape[3])
    return _GatherWSum.apply(w.float().contiguous(), feat.float().contiguous(), idx)


# ------------------------------------------------------------------------------------------------
# PointConv neighbourhood mixing (models/point_conv.py:60-66)
# ------------------------------------------------------------------------------------------------
class _PointConvMix(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, feat_cl, wgt, knn_indices, k):
        lib = _lib.load()
        b, m, ch = feat_cl.shape
        wn, n = wgt.shape[1], wgt.shape[2]
        out = torch.empty((b, n, wn, ch), dtype=torch.float32, device=feat_cl.device)
        with _on_device(feat_cl):
            _lib.launch('camli_pointconv_mix_fwd', lib.camli_pointconv_mix_fwd, feat_cl.data_ptr(), wgt.data_ptr(),
                        knn_indices.data_ptr(), knn_indices.stride(1), out.data_ptr(), b, m, n, ch, wn, k,
                        _stream_ptr(feat_cl), work=(4.0 * b * n * (k * ch + wn * k + wn * ch) + 8.0 * b * n * k, 'B'))
        ctx.save_for_backward(feat_cl, wgt, knn_indices)
        ctx.k = k
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        feat_cl, wgt, knn_indices = ctx.saved_tensors
        b, m, ch = feat_cl.shape
        wn, n = wgt.shape[1], wgt.shape[2]
        k = ctx.k
        gout = gout.contiguous().float()
        gwgt = torch.empty_like(wgt) if ctx.needs_input_grad[1] else None
        if k == 16 and wn == 16:
            # atomic-free: per-point row gradients into a scratch tensor, then a segment sum through the inverse
            # neighbour map (built once per neighbour table)
            gfeat = scratch = order = offsets = None
            if ctx.needs_input_grad[0]:
                flat = knn_indices[:, :, :k].reshape(b, n * k) if knn_indices.shape[2] != k else knn_indices.view(b, n * k)
                order, offsets = inverse_map(flat, m)
                gfeat = torch.empty_like(feat_cl)
                scratch = torch.empty(lib.camli_pointconv_mix_bwd_scratch_bytes(b, n, ch, k) // 4, dtype=torch.float32,
                                      device=gout.device)
            with _on_device(feat_cl):
                _lib.launch('camli_pointconv_mix_bwd', lib.camli_pointconv_mix_bwd_sorted, gout.data_ptr(), feat_cl.data_ptr(),
                            wgt.data_ptr(), knn_indices.data_ptr(), knn_indices.stride(1),
                            order.data_ptr() if order is not None else None,
                            offsets.data_ptr() if offsets is not None else None,
                            scratch.data_ptr() if scratch is not None else None,
                            gfeat.data_ptr() if gfeat is not None else None, gwgt.data_ptr() if gwgt is not None else None,
                            b, m, n, ch, wn, k, _stream_ptr(feat_cl),
                            work=(4.0 * b * n * (wn * ch + 2 * k * ch + 2 * wn * k) + 8.0 * b * n * k, 'B'))
            return gfeat, gwgt, None, None
        gfeat = torch.zeros_like(feat_cl) if ctx.needs_input_grad[0] else None
        with _on_device(feat_cl):
            _lib.launch('camli_pointconv_mix_bwd', lib.camli_pointconv_mix_bwd, gout.data_ptr(), feat_cl.data_ptr(),
                        wgt.data_ptr(), knn_indices.data_ptr(), knn_indices.stride(1),
                        gfeat.data_ptr() if gfeat is not None else None, gwgt.data_ptr() if gwgt is not None else None,
                        b, m, n, ch, wn, k, _stream_ptr(feat_cl),
                        work=(4.0 * b * n * (wn * ch + 2 * k * ch + 2 * wn * k) + 8.0 * b * n * k, 'B'))
        return gfeat, gwgt, None, None


def pointconv_mix(feat_cl, wgt, knn_indices, k):
    """feat_cl [B,M,CH] channel-last, wgt [B,Wn,N,k] (as weight_net emits it), knn_indices int64
    [B,N,>=k] -> [B,N,Wn,CH] = per-point (Wn x k) @ (k x CH) over the gathered neighbour rows."""
    _require_cuda('pointconv_mix', feat_cl, wgt, knn_indices)
    assert knn_indices.dtype == torch.int64 and knn_indices.stride(2) == 1 and knn_indices.shape[2] >= k
    assert knn_indices.stride(0) == knn_indices.shape[1] * knn_indices.stride(1)
    assert wgt.shape[3] == k and wgt.shape[1] <= 16
    return _PointConvMix.apply(feat_cl.float().contiguous(), wgt.float().contiguous(), knn_indices, k)


# ------------------------------------------------------------------------------------------------
# convex flow up-sampling (models/utils.py:191-204, models/raft_core.py:183-197)
# ------------------------------------------------------------------------------------------------
class _ConvexUpsample(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, flow, mask, mask_bias, scale, mask_scale, out_rows=None):
        lib = _lib.load()
        b, _, h, w = flow.shape
        rows = h * scale if out_rows is None else int(out_rows)
        assert 1 <= rows <= h * scale
        ctx.rows = rows
        out = torch.empty((b, 2, rows, w * scale), dtype=torch.float32, device=flow.device)
        with _on_device(flow):
            _lib.launch('camli_convex_upsample_fwd', lib.camli_convex_upsample_rows_fwd, flow.data_ptr(), mask.data_ptr(),
                        mask_bias.data_ptr() if mask_bias is not None else None,
                        out.data_ptr(), b, h, w, scale, rows, float(mask_scale), _stream_ptr(flow),
                        work=(4.0 * b * h * w * (9 * scale * scale + 2 * scale * scale + 2), 'B'))
        if mask_bias is None:
            ctx.save_for_backward(flow, mask)
        else:
            ctx.save_for_backward(flow, mask, mask_bias)
            ctx.bias_param = _runtime.deferral_target(mask_bias)
        ctx.scale, ctx.mask_scale = scale, mask_scale
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        flow, mask = ctx.saved_tensors[:2]
        mask_bias = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        b, _, h, w = flow.shape
        gout = gout.contiguous().float()
        gflow = torch.zeros_like(flow)
        gmask = torch.empty_like(mask)
        with _on_device(flow):
            _lib.launch('camli_convex_upsample_bwd', lib.camli_convex_upsample_rows_bwd, gout.data_ptr(), flow.data_ptr(),
                        mask.data_ptr(), mask_bias.data_ptr() if mask_bias is not None else None, gflow.data_ptr(),
                        gmask.data_ptr(), b, h, w, ctx.scale, ctx.rows, float(ctx.mask_scale), _stream_ptr(flow),
                        work=(4.0 * b * h * w * (2 * 9 * ctx.scale ** 2 + 2 * ctx.scale ** 2 + 4), 'B'))
        gbias = None
        if mask_bias is not None and (ctx.bias_param is not None or ctx.needs_input_grad[2]):
            # d/d bias = the per-channel sum of the mask gradient: the identity form of the bias epilogue's adjoint
            c = mask.shape[1]
            deferred = ctx.bias_param is not None
            acc = _runtime.PARAM_GRADS.slot(ctx.bias_param, lambda: _zero_slice(c, gmask), False) if deferred else _zero_slice(c, gmask)
            with _on_device(flow):
                _lib.launch('camli_bias_act_bwd', lib.camli_bias_act_bwd, gmask.data_ptr(), None, None, None, acc.data_ptr(),
                            b, c, h * w, 0, _stream_ptr(flow), work=(4.0 * b * c * h * w, 'B'))
            gbias = None if deferred else acc
        return gflow, gmask, gbias, None, None, None


def convex_upsample(flow, mask, scale_factor=8, mask_scale=1.0, mask_bias=None, out_rows=None):
    """flow [B,2,h,w], raw mask [B,9*S*S,h,w] -> [B,2,h*S,w*S]; ``mask_scale`` multiplies the mask inside
    the kernel (RAFT passes 0.25, raft_core.py:195); ``mask_bias`` [9*S*S]: added to the mask first (the bias of the mask
    head's last convolution, folded in); ``out_rows``: keep the first out_rows fine rows only (the un-padding of a
    bottom-padded image, done by the kernel)."""
    _require_cuda('convex_upsample', flow, mask)
    assert flow.shape[1] == 2 and mask.shape[1] == 9 * scale_factor * scale_factor
    if mask_bias is not None:
        assert mask_bias.shape == (mask.shape[1],) and mask_bias.is_contiguous() and mask_bias.dtype == torch.float32
    return _ConvexUpsample.apply(flow.float().contiguous(), mask.float().contiguous(), mask_bias, scale_factor, mask_scale, out_rows)


# ------------------------------------------------------------------------------------------------
# convolutional GRU, elementwise halves (models/raft_core.py:123-139)
# ------------------------------------------------------------------------------------------------
class _GruGates(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, pre_zr, ctx_zr, h):
        lib = _lib.load()
        pre_zr, ctx_zr, h = pre_zr.contiguous(), ctx_zr.contiguous(), h.contiguous()
        b, c = h.shape[0], h.shape[1]
        p = h[0, 0].numel()
        z, r, rh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        with _on_device(h):
            _lib.launch('camli_gru_gates_fwd', lib.camli_gru_gates_fwd, pre_zr.data_ptr(), ctx_zr.data_ptr(), h.data_ptr(),
                        z.data_ptr(), r.data_ptr(), rh.data_ptr(), b, c, p, _stream_ptr(h), work=(32.0 * b * c * p, 'B'))
        ctx.save_for_backward(z, r, h)
        ctx.mark_non_differentiable(r)
        # no zero tensors for outputs without a gradient (r never has one: that was a 33 MB fill per call; backward takes None)
        ctx.set_materialize_grads(False)
        return z, rh, r

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gz, grh, _gr):
        lib = _lib.load()
        z, r, h = ctx.saved_tensors
        b, c = h.shape[0], h.shape[1]
        p = h[0, 0].numel()
        gz = gz.float() if gz is not None else torch.zeros_like(h)
        grh = grh.float() if grh is not None else torch.zeros_like(h)
        # grh is a channel slice of the gradient of cat([r*h, x]): read in place (camli_gru_gates_bwd_strided)
        gz_bs, grh_bs = _batch_strided(gz), _batch_strided(grh)
        if gz_bs is None:
            gz, gz_bs = gz.contiguous(), c * p
        if grh_bs is None:
            grh, grh_bs = grh.contiguous(), c * p
        gpre = torch.empty((b, 2 * c) + tuple(h.shape[2:]), dtype=torch.float32, device=h.device)
        gh = torch.empty_like(h)
        with _on_device(h):
            _lib.launch('camli_gru_gates_bwd', lib.camli_gru_gates_bwd_strided, gz.data_ptr(), gz_bs, grh.data_ptr(), grh_bs,
                        z.data_ptr(), r.data_ptr(), h.data_ptr(), gpre.data_ptr(), gh.data_ptr(), b, c, p, _stream_ptr(h),
                        work=(32.0 * b * c * p, 'B'))
        return gpre, gpre, gh


class _GruBlend(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, pre_q, ctx_q, z, h, sanitize=False):
        lib = _lib.load()
        ctx.sanitize = bool(sanitize)
        pre_q, ctx_q, z, h = pre_q.contiguous(), ctx_q.contiguous(), z.contiguous(), h.contiguous()
        b, c = h.shape[0], h.shape[1]
        p = h[0, 0].numel()
        q, h_new = torch.empty_like(h), torch.empty_like(h)
        with _on_device(h):
            _lib.launch('camli_gru_blend_fwd', lib.camli_gru_blend_fwd, pre_q.data_ptr(), ctx_q.data_ptr(), z.data_ptr(),
                        h.data_ptr(), q.data_ptr(), h_new.data_ptr(), b, c, p, int(ctx.sanitize), _stream_ptr(h),
                        work=(24.0 * b * c * p, 'B'))
        ctx.save_for_backward(z, h, q)
        return h_new

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        lib = _lib.load()
        z, h, q = ctx.saved_tensors
        b, c = h.shape[0], h.shape[1]
        p = h[0, 0].numel()
        g = g.contiguous().float()
        gpre, gz, gh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        with _on_device(h):
            _lib.launch('camli_gru_blend_bwd', lib.camli_gru_blend_bwd, g.data_ptr(), z.data_ptr(), h.data_ptr(),
                        q.data_ptr(), gpre.data_ptr(), gz.data_ptr(), gh.data_ptr(), b, c, p, int(ctx.sanitize), _stream_ptr(h),
                        work=(28.0 * b * c * p, 'B'))
        return gpre, gpre, gz, gh, None


def gru_gates(pre_zr, ctx_zr, h):
    """(z, r*h) of one GRU half-step: z|r = sigmoid(pre_zr + ctx_zr) ([B,2C,...]), h [B,C,...]."""
    _require_cuda('gru_gates', pre_zr, ctx_zr, h)
    z, rh, _ = _GruGates.apply(pre_zr, ctx_zr, h)
    return z, rh


def gru_blend(pre_q, ctx_q, z, h, nan_to_num=False):
    """h' = (1 - z) * h + z * tanh(pre_q + ctx_q); ``nan_to_num``: followed by torch.nan_to_num (raft_core.py:138) in the
    same pass, its adjoint folded into the blend's."""
    _require_cuda('gru_blend', pre_q, ctx_q, z, h)
    return _GruBlend.apply(pre_q, ctx_q, z, h, bool(nan_to_num))


# ------------------------------------------------------------------------------------------------
# bias + activation epilogue (models/mlp.py:41-128, models/raft_core.py:155-197)
# ------------------------------------------------------------------------------------------------
ACT_CODES = {None: 0, 'none': 0, 'relu': 1, 'leaky_relu': 2, 'sigmoid': 3, 'tanh': 4,
             'relu_nan_to_num': 5}      # relu, then torch.nan_to_num (raft_core.py:163-164); needs a plane of 4k elements


_STRIDED_GRADS = os.environ.get('CAMLI_STRIDED_GRADS', '1') != '0'      # 0: copy sliced gradients out first (A/B)


def _batch_strided(t):
    """t [B,C,...] fp32 whose batch entries are dense [C,...] blocks a fixed stride apart -- a contiguous tensor or a channel
    slice of one (the adjoint of a cat hands those over) -> its batch stride in floats, usable by the *_strided entry points
    (16-byte aligned, stride a multiple of 4); else None."""
    if t.dtype != torch.float32 or t.dim() < 2 or not _STRIDED_GRADS:
        return None
    inner = 1
    for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if size != 1 and stride != inner:
            return None
        inner *= size
    bs = t.stride(0) if t.shape[0] > 1 else inner
    if bs < inner or bs % 4 or (t.data_ptr() % 16):
        return None
    return bs


# ------------------------------------------------------------------------------------------------
# layout passes around the channels-last convolutions of the update block (cores/blocks.py:_CatConvCL)
# ------------------------------------------------------------------------------------------------
def _transpose_planes(src, src_off, src_bs, src_rs, dst, dst_off, dst_bs, dst_rs, b, rows, cols):
    lib = _lib.load()
    with _on_device(src):
        _lib.launch('camli_transpose_planes', lib.camli_transpose_planes, src.data_ptr() + 4 * src_off, src_bs, src_rs,
                    dst.data_ptr() + 4 * dst_off, dst_bs, dst_rs, b, rows, cols, _stream_ptr(src), work=(8.0 * b * rows * cols, 'B'))


def nchw_into_channels_last(part, x_cl, c0):
    """part [B,C,H,W] (contiguous, or a channel slice of a contiguous map) -> channels c0 .. c0+C of x_cl, a [B,Ct,H,W] tensor
    in channels_last memory format (memory [B,H,W,Ct])."""
    _require_cuda('nchw_into_channels_last', part, x_cl)
    b, c, hh, ww = part.shape
    ct, p = x_cl.shape[1], hh * ww
    bs = _batch_strided(part)
    if bs is None:
        part = part.float().contiguous()
        bs = c * p
    _transpose_planes(part, 0, bs, p, x_cl, c0, p * ct, ct, b, c, p)


def channels_last_to_nchw(x_cl, c0, c):
    """channels c0 .. c0+c of x_cl ([B,Ct,H,W] in channels_last memory format) -> a contiguous [B,c,H,W] tensor."""
    _require_cuda('channels_last_to_nchw', x_cl)
    b, ct, hh, ww = x_cl.shape
    if not (x_cl.stride(1) == 1 and x_cl.stride(3) == ct and x_cl.stride(2) == ww * ct and (b == 1 or x_cl.stride(0) == hh * ww * ct)):
        # a library build that answered a channels-last convolution with an NCHW tensor: reading it as
        # [B,H,W,C] memory would scramble it silently
        x_cl = x_cl.contiguous(memory_format=torch.channels_last)
        if x_cl.stride(1) != 1:         # (size-1 dimensions make the format ambiguous: force the physical layout)
            x_cl = x_cl.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    p = hh * ww
    out = torch.empty((b, c, hh, ww), dtype=torch.float32, device=x_cl.device)
    _transpose_planes(x_cl, c0, p * ct, ct, out, 0, c * p, p, b, p, c)
    return out


# ------------------------------------------------------------------------------------------------
# channels-last tap convolution on the fp32 matrix cores (csrc/hip/convcl.hip, round 5): forward, data gradient (the same
# kernel on the negated taps and the transposed packing) and weight gradient of a stride-1 convolution whose operands are
# NHWC tensors.  GRU2D's 1x5 / 5x1 convolutions (models/raft_core.py:110-140) run on it in training (cores/blocks._CatConvCL).
# ------------------------------------------------------------------------------------------------
def convcl_taps(kh, kw, ph, pw, negate=False):
    """(T, dy bytes, dx bytes) of a kh x kw kernel with padding (ph, pw), taps in row-major order (the order of the weight
    tensor's last two dimensions); ``negate``: the taps of the data gradient."""
    import struct
    sign = -1 if negate else 1
    dy = [sign * (ky - ph) for ky in range(kh) for _ in range(kw)]
    dx = [sign * (kx - pw) for _ in range(kh) for kx in range(kw)]
    return kh * kw, struct.pack('%db' % len(dy), *dy), struct.pack('%db' % len(dx), *dx)


def convcl_pack(w):
    """[Cout, Cin, kh, kw] -> [Cout][T][Cin] (forward) and [Cin][T][Cout] (data gradient); cached on the tensor for as long
    as its version stands (GRU2D's weight blocks are cut once per pass)."""
    cached = getattr(w, '_camli_convcl_packs', None)
    if cached is not None and cached[0] == w._version:
        return cached[1], cached[2]
    with torch.no_grad():
        wd = w.detach().float()
        co, ci, kh, kw = wd.shape
        fwd = wd.permute(0, 2, 3, 1).reshape(co, kh * kw, ci).contiguous()
        bwd = wd.permute(1, 2, 3, 0).reshape(ci, kh * kw, co).contiguous()
    try:
        w._camli_convcl_packs = (w._version, fwd, bwd)
    except Exception:       # noqa: BLE001 -- a tensor subclass that refuses attributes: pack again next time
        pass
    return fwd, bwd


def _nhwc_ld(t):
    """floats per pixel of a [B,H,W,C] tensor whose pixels are rows of one dense [P, ld] matrix (channel slices allowed);
    None when it is not one.  (The strides of size-1 dimensions carry no information and are not looked at.)"""
    b, hh, ww, c = t.shape
    ld = t.stride(2) if ww > 1 else (t.stride(1) if hh > 1 else (t.stride(0) if b > 1 else c))
    ok = ((c == 1 or t.stride(3) == 1) and ld >= c and (hh == 1 or t.stride(1) == ww * ld)
          and (b == 1 or t.stride(0) == hh * ww * ld))
    return ld if ok and ld % 4 == 0 and t.data_ptr() % 16 == 0 else None


def convcl_supported(cin, cout, parts=1):
    """forward + both adjoints: channels in multiples of 16 in / 128 out for the convolution itself; the weight gradient wants
    256 | Cin and 128 | Cout, or (one input tensor) 128 | Cin and 256 | Cout (camli_convcl_wrw)."""
    fwd = cin % 16 == 0 and cout % 128 == 0 and cout >= 128 and parts <= 2
    dgrad = cout % 16 == 0 and cin % 128 == 0
    wrw = (cin % 256 == 0 and cout % 128 == 0) or (parts == 1 and cin % 128 == 0 and cout % 256 == 0)
    return fwd and dgrad and wrw


def convcl(xs, wp, taps, split=None, out=None, accumulate=(False, False)):
    """y[p][n] = sum_t sum_c cat(xs)[p + tap_t][c] * wp[n][t][c].  xs: one or two fp32 NHWC tensors [B,H,W,Ci] (channel slices of
    wider NHWC tensors allowed); wp [Cout][T][Cin]; taps from convcl_taps.  Returns [B,H,W,Cout], or with ``split`` = N0 the pair
    ([B,H,W,N0], [B,H,W,Cout-N0]).  ``out``: existing output tensor(s); ``accumulate[i]``: add into out[i] instead of writing it."""
    _require_cuda('convcl', wp, *xs)
    lib = _lib.load()
    t, dy, dx = taps
    x0 = xs[0]
    x1 = xs[1] if len(xs) > 1 else None
    b, hh, ww, c0 = x0.shape
    c1 = x1.shape[3] if x1 is not None else 0
    cout = wp.shape[0]
    assert wp.shape == (cout, t, c0 + c1) and wp.is_contiguous() and wp.dtype == torch.float32
    ld0, ld1 = _nhwc_ld(x0), (_nhwc_ld(x1) if x1 is not None else 0)
    if ld0 is None or ld1 is None:
        raise _lib.CamliHipError('convcl: inputs must be dense fp32 NHWC tensors (16-byte aligned, pixel stride a multiple of 4)')
    n0 = cout if split is None else int(split)
    if out is None:
        y0 = torch.empty((b, hh, ww, n0), dtype=torch.float32, device=x0.device)
        y1 = torch.empty((b, hh, ww, cout - n0), dtype=torch.float32, device=x0.device) if n0 < cout else None
        assert not any(accumulate)
    else:
        y0, y1 = (out, None) if torch.is_tensor(out) else out
        assert y0.shape == (b, hh, ww, n0) and (y1 is None) == (n0 == cout) and (y1 is None or y1.shape == (b, hh, ww, cout - n0))
    ldy0, ldy1 = _nhwc_ld(y0), (_nhwc_ld(y1) if y1 is not None else 0)
    if ldy0 is None or ldy1 is None:
        raise _lib.CamliHipError('convcl: outputs must be dense fp32 NHWC tensors')
    with _on_device(x0):
        _lib.launch('camli_convcl_fwd', lib.camli_convcl_fwd, x0.data_ptr(), ld0, c0, x1.data_ptr() if x1 is not None else 0, ld1, c1,
                    wp.data_ptr(), y0.data_ptr(), ldy0, n0, y1.data_ptr() if y1 is not None else 0, ldy1, b, hh, ww, cout, t, dy, dx,
                    int(bool(accumulate[0])), int(bool(accumulate[1])), _stream_ptr(x0),
                    work=(4.0 * b * hh * ww * (c0 + c1 + cout), 'B'), flop=2.0 * b * hh * ww * cout * (c0 + c1) * t)
    return y0 if y1 is None else (y0, y1)


_convcl_workspaces = {}


def convcl_wrw(xs, gy, taps, kernel_hw, out=None):
    """Weight gradient [Cout, Cin, kh, kw] of ``convcl(xs, pack(w), taps)`` for the output gradient gy [B,H,W,Cout] (NHWC);
    ``out``: add into this tensor instead of creating one."""
    _require_cuda('convcl_wrw', gy, *xs)
    lib = _lib.load()
    t, dy, dx = taps
    x0 = xs[0]
    x1 = xs[1] if len(xs) > 1 else None
    b, hh, ww, c0 = x0.shape
    c1 = x1.shape[3] if x1 is not None else 0
    cout = gy.shape[3]
    ld0, ld1, ldg = _nhwc_ld(x0), (_nhwc_ld(x1) if x1 is not None else 0), _nhwc_ld(gy)
    if ld0 is None or ld1 is None or ldg is None:
        raise _lib.CamliHipError('convcl_wrw: operands must be dense fp32 NHWC tensors (16-byte aligned, pixel stride a multiple of 4)')
    need = lib.camli_convcl_wrw_workspace_bytes(b, hh, ww, c0 + c1, cout, t)
    if need <= 0:
        raise _lib.CamliHipError('convcl_wrw: unsupported shape Cin=%d Cout=%d' % (c0 + c1, cout))
    # one workspace per (device, stream): the parts of a launch are consumed by the reduction of the same launch
    key = (gy.device, torch.cuda.current_stream(gy.device).cuda_stream)
    ws = _convcl_workspaces.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _convcl_workspaces[key] = torch.empty(need // 4, dtype=torch.float32, device=gy.device)
    accumulate = out is not None
    gw = out if accumulate else torch.empty((cout, c0 + c1) + tuple(kernel_hw), dtype=torch.float32, device=gy.device)
    assert gw.is_contiguous() and gw.shape == (cout, c0 + c1) + tuple(kernel_hw) and kernel_hw[0] * kernel_hw[1] == t
    with _on_device(gy):
        _lib.launch('camli_convcl_wrw', lib.camli_convcl_wrw, x0.data_ptr(), ld0, c0, x1.data_ptr() if x1 is not None else 0, ld1, c1,
                    gy.data_ptr(), ldg, ws.data_ptr(), ws.numel() * 4, gw.data_ptr(), int(accumulate), b, hh, ww, cout, t, dy, dx,
                    _stream_ptr(gy), work=(4.0 * b * hh * ww * (c0 + c1 + cout), 'B'), flop=2.0 * b * hh * ww * cout * (c0 + c1) * t)
    return gw


# ------------------------------------------------------------------------------------------------
# GRU2D, one whole update (both half-steps) as ONE autograd node on channels-last tensors (round 5).
# models/raft_core.py:122-139 with the context term hoisted (cores/raft2d.GRU2D.prepare):
#     half-step s:   z | r = sigmoid(conv_zr_s(cat[h, m]) + ctx_zr_s);   q = tanh(conv_q_s(cat[r h, m]) + ctx_q_s);   h <- (1 - z) h + z q
# Forward: two layout passes in (h, m -> NHWC), four convolutions with the gate arithmetic in their epilogues
# (camli_convcl_gru_gates / _blend), one layout pass out: 7 launches where the per-convolution nodes took 24 (six transposes
# per convolution node, gate / blend kernels, the motion features transposed four times).  Backward: blend / gate adjoints on the
# NHWC tensors (the stand-alone kernels of gru.hip read them as [P][128][1]), the data gradients as convolutions on the negated
# taps that ADD into the hidden-state / motion gradients they complete, weight gradients by camli_convcl_wrw.
# ------------------------------------------------------------------------------------------------
def _to_nhwc(x):
    """[B,C,H,W] fp32 -> a new dense [B,H,W,C] tensor (camli_transpose_planes)."""
    b, c, hh, ww = x.shape
    out = torch.empty((b, hh, ww, c), dtype=torch.float32, device=x.device)
    nchw_into_channels_last(x, out.permute(0, 3, 1, 2), 0)
    return out


def _to_nchw(x_n):
    return channels_last_to_nchw(x_n.permute(0, 3, 1, 2), 0, x_n.shape[3])


# CAMLI_GRU_WINO=1: GRU2D's half-step convolutions and their data gradients as 1-D Winograd F(4,5) (csrc/hip/wino1d.hip:
# 8 multiplications per 4 outputs where the 5-tap form spends 20); 0: the tap convolutions of convcl.hip
_GRU_WINO = os.environ.get('CAMLI_GRU_WINO', '1') != '0'
# CAMLI_GRU_WINO_WRW=0: their weight gradients stay on the tap form (camli_convcl_wrw) while the rest runs the Winograd form
_GRU_WINO_WRW = os.environ.get('CAMLI_GRU_WINO_WRW', '1') != '0'
# CAMLI_GRU_KEEP_V=0: the weight gradients transform their inputs again instead of contracting the transformed input [8][tiles][C] the
# forward kept (134 MB per convolution at batch 8, 6.4 GB over the 12 updates of a pass)
_GRU_KEEP_V = os.environ.get('CAMLI_GRU_KEEP_V', '1') != '0'


def wino1d_weights(wp, flip):
    """U [8][N][C] of packed weights wp [N][5][C] (convcl_pack), for the forward (flip False) or -- from the transposed packing
    -- the data gradient (flip True)."""
    lib = _lib.load()
    n, t, c = wp.shape
    assert t == 5 and wp.is_contiguous() and wp.dtype == torch.float32
    u = torch.empty((8, n, c), dtype=torch.float32, device=wp.device)
    with _on_device(wp):
        _lib.launch('camli_wino1d_weights', lib.camli_wino1d_weights, wp.data_ptr(), u.data_ptr(), n, c, int(flip), _stream_ptr(wp),
                    work=(4.0 * 13 * n * c, 'B'))
    return u


def _wino1d_workspace(b, hh, ww, cin, cout, axis, device):
    need = _lib.load().camli_wino1d_workspace_bytes(b, hh, ww, cin, cout, axis)
    return torch.empty(need // 4, dtype=torch.float32, device=device), need


def wino1d_conv(xs, u, axis, split=None, out=None, accumulate=(False, False)):
    """convcl(xs, wp, taps of a 1x5 (axis 0) / 5x1 (axis 1) kernel) on the Winograd form; u = wino1d_weights(wp, ...)."""
    _require_cuda('wino1d_conv', u, *xs)
    lib = _lib.load()
    x0 = xs[0]
    x1 = xs[1] if len(xs) > 1 else None
    b, hh, ww, c0 = x0.shape
    c1 = x1.shape[3] if x1 is not None else 0
    cout = u.shape[1]
    assert u.shape == (8, cout, c0 + c1)
    ld0, ld1 = _nhwc_ld(x0), (_nhwc_ld(x1) if x1 is not None else 0)
    n0 = cout if split is None else int(split)
    if out is None:
        y0 = torch.empty((b, hh, ww, n0), dtype=torch.float32, device=x0.device)
        y1 = torch.empty((b, hh, ww, cout - n0), dtype=torch.float32, device=x0.device) if n0 < cout else None
        assert not any(accumulate)
    else:
        y0, y1 = (out, None) if torch.is_tensor(out) else out
    ldy0, ldy1 = _nhwc_ld(y0), (_nhwc_ld(y1) if y1 is not None else 0)
    if None in (ld0, ld1, ldy0, ldy1):
        raise _lib.CamliHipError('wino1d_conv: operands must be dense fp32 NHWC tensors (16-byte aligned, pixel stride a multiple of 4)')
    ws, need = _wino1d_workspace(b, hh, ww, c0 + c1, cout, axis, x0.device)
    tiles = need // (32 * (c0 + c1 + cout))
    with _on_device(x0):
        _lib.launch('camli_wino1d_conv', lib.camli_wino1d_conv, x0.data_ptr(), ld0, c0, x1.data_ptr() if x1 is not None else 0, ld1, c1,
                    u.data_ptr(), y0.data_ptr(), ldy0, n0, y1.data_ptr() if y1 is not None else 0, ldy1, ws.data_ptr(), need, b, hh, ww,
                    cout, axis, int(bool(accumulate[0])), int(bool(accumulate[1])), _stream_ptr(x0),
                    work=(4.0 * b * hh * ww * (c0 + c1 + cout) + 2.0 * need, 'B'), flop=2.0 * 8 * tiles * cout * (c0 + c1))
    return y0 if y1 is None else (y0, y1)


def wino1d_wrw(xs, gy, axis, out=None, v=None):
    """convcl_wrw(xs, gy, taps of a 1x5 (axis 0) / 5x1 (axis 1) kernel) contracted in the Winograd domain: the weight gradient
    [Cout, Cin, 1, 5] | [Cout, Cin, 5, 1]; ``out``: add into this tensor; ``v``: the transformed input the forward kept
    (camli_wino1d_gru_gates / _blend's v_keep), contracted as it lies instead of transforming xs again."""
    _require_cuda('wino1d_wrw', gy, *xs)
    lib = _lib.load()
    x0 = xs[0]
    x1 = xs[1] if len(xs) > 1 else None
    b, hh, ww, c0 = x0.shape
    c1 = x1.shape[3] if x1 is not None else 0
    cout = gy.shape[3]
    ld0, ld1, ldg = _nhwc_ld(x0), (_nhwc_ld(x1) if x1 is not None else 0), _nhwc_ld(gy)
    need = lib.camli_wino1d_wrw_workspace_bytes(b, hh, ww, c0 + c1, cout, axis)
    if None in (ld0, ld1, ldg) or need <= 0:
        raise _lib.CamliHipError('wino1d_wrw: unsupported operands (dense fp32 NHWC, Cin a multiple of 256, Cout of 128)')
    ws = torch.empty(need // 4, dtype=torch.float32, device=gy.device)
    shape = (cout, c0 + c1, 1, 5) if axis == 0 else (cout, c0 + c1, 5, 1)
    gw = out if out is not None else torch.empty(shape, dtype=torch.float32, device=gy.device)
    assert gw.shape == shape and gw.is_contiguous()
    tiles = b * (hh * ((ww + 3) // 4) if axis == 0 else ww * ((hh + 3) // 4))
    with _on_device(gy):
        _lib.launch('camli_wino1d_wrw', lib.camli_wino1d_wrw, x0.data_ptr(), ld0, c0, x1.data_ptr() if x1 is not None else 0, ld1, c1,
                    gy.data_ptr(), ldg, gw.data_ptr(), v.data_ptr() if v is not None else None, ws.data_ptr(), need, b, hh, ww, cout, axis,
                    int(out is not None), _stream_ptr(gy),
                    work=(4.0 * b * hh * ww * ((c0 + c1) * (2.0 if v is not None else 3.0) + cout * 3.0), 'B'), flop=2.0 * 8 * tiles * cout * (c0 + c1))
    return gw


class GRU2DPass:
    """What the GRU2D updates of one pass share: the four weight blocks and the four hoisted context terms (NHWC), and the
    running totals of their gradients.  Every update's adjoint ADDS into the totals (the weight gradients through
    camli_convcl_wrw's accumulate mode, the context gradients inside the gate / blend adjoint kernels); ``token`` makes the
    hub node's backward run after the last update's, and it hands the totals to autograd once.  Without the hub each of the 12
    updates returns 8 gradient tensors that autograd sums: 88 additions per step, 44 of them over [B,H,W,256] / [B,H,W,128]
    maps."""

    def __init__(self, weights, contexts):
        self.weights = tuple(weights)
        self.contexts = tuple(contexts)
        self.gw = [None] * 4
        self.gc = [None] * 4
        self.side = None            # runtime._WgradSide once a weight gradient has been issued beside the main chain
        self.done = None            # event behind the last update adjoint's accumulations (the hub node waits on it)
        self.two_lane = _runtime.lanes_live()       # this pass runs two-lane: its backward may use the weight-gradient side stream
        self.wino = {}              # (id of a packed weight tensor, flip) -> its Winograd transform, once per pass
        self.token = _GRU2DHub.apply(self, *self.weights, *self.contexts)

    def wino_u(self, wp, flip):
        key = (id(wp), flip)
        if key not in self.wino:
            self.wino[key] = (wp, wino1d_weights(wp, flip))         # (the packed tensor is kept: its id stays unique)
        return self.wino[key][1]


class _GRU2DHub(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hub, *tensors):
        ctx.hub = weakref.ref(hub)          # hub.token is this node's output: no strong back-reference (cycle)
        ctx.meta = [(tuple(t.shape), t.device) for t in tensors]
        return tensors[0].new_zeros(1)

    @staticmethod
    def backward(ctx, _gtoken):
        hub = ctx.hub()
        totals = [None] * 8
        if hub is None and any(ctx.needs_input_grad[1:]):
            # the pass object owns the running totals: without it the gradients of the GRU's weights / context terms would
            # silently come out as zeros
            raise RuntimeError('GRU2DPass was released before its backward ran: keep the object returned by GRU2D.prepare '
                               'alive until the pass has been differentiated')
        if hub is not None:
            if hub.done is not None:
                # the update adjoints return no gradient for the token, so the engine does not order this node's stream behind
                # theirs: wait for the last accumulation explicitly (they may run in a lane / Branch of their own one day)
                torch.cuda.current_stream(hub.done[1]).wait_event(hub.done[0])
                hub.done = None
            if hub.side is not None:        # the weight gradients were accumulated on the side stream
                hub.side.join()
                for t in hub.gw:
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream(t.device))
                hub.side = None
            totals = hub.gw + hub.gc
            hub.gw, hub.gc = [None] * 4, [None] * 4
        need = ctx.needs_input_grad[1:]
        grads = [(t if t is not None else torch.zeros(shape, dtype=torch.float32, device=dev)) if n else None
                 for t, (shape, dev), n in zip(totals, ctx.meta, need)]
        return (None, *grads)


class _GRU2DStepCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, m, token, hub):
        lib = _lib.load()
        b, hd, hh, ww = h.shape
        cx = m.shape[1]
        w_zr1, w_q1, w_zr2, w_q2 = hub.weights
        c_zr1, c_q1, c_zr2, c_q2 = hub.contexts
        h0, mn = _to_nhwc(h.float()), _to_nhwc(m.float())
        ctx.geom = []
        ctx.hub = hub
        kept = ctx.kept = []        # per half-step: the transformed inputs (z|r convolution, q convolution) or (None, None)
        saved = [h0, mn]
        hcur = h0
        with _on_device(h):
            for half, (w_zr, w_q, c_zr, c_q) in enumerate(((w_zr1, w_q1, c_zr1, c_q1), (w_zr2, w_q2, c_zr2, c_q2))):
                kh, kw = w_zr.shape[2:]
                geom = (kh, kw, kh // 2, kw // 2)
                t, dy, dx = convcl_taps(*geom)
                wp_zr, wpt_zr = convcl_pack(w_zr)
                wp_q, wpt_q = convcl_pack(w_q)
                z, rh, r, q, hn = (torch.empty_like(h0) for _ in range(5))
                flop = 2.0 * b * hh * ww * (hd + cx) * t
                wino = _GRU_WINO and t == 5 and (kh, kw) in ((1, 5), (5, 1)) and hd == 128
                if wino:
                    # r6: the two convolutions as 1-D Winograd F(4,5) with the same epilogues (csrc/hip/wino1d.hip)
                    axis = 0 if kh == 1 else 1
                    ws, need = _wino1d_workspace(b, hh, ww, hd + cx, 2 * hd, axis, h.device)
                    tiles = need // (32 * (3 * hd + cx))
                    # the transformed inputs stay for the weight gradients when those can contract them as they lie
                    keep = (_GRU_KEEP_V and _GRU_WINO_WRW and any(ctx.needs_input_grad[:2]) and any(w.requires_grad for w in (w_zr, w_q))
                            and lib.camli_wino1d_wrw_reuse(b, hh, ww, hd + cx, 2 * hd, axis) and lib.camli_wino1d_wrw_reuse(b, hh, ww, hd + cx, hd, axis))
                    v_zr, v_q = ((torch.empty(8 * tiles * (hd + cx), dtype=torch.float32, device=h.device) for _ in range(2)) if keep
                                 else (None, None))
                    kept.append((v_zr, v_q))
                    _lib.launch('camli_wino1d_gru_gates', lib.camli_wino1d_gru_gates, hcur.data_ptr(), mn.data_ptr(), cx,
                                hub.wino_u(wp_zr, False).data_ptr(), c_zr.data_ptr(), z.data_ptr(), rh.data_ptr(), r.data_ptr(),
                                v_zr.data_ptr() if keep else None, ws.data_ptr(),
                                need, b, hh, ww, axis, _stream_ptr(h), work=(4.0 * b * hh * ww * (6 * hd + cx) + 2.0 * need, 'B'),
                                flop=2.0 * 8 * tiles * (hd + cx) * 2 * hd)
                    _lib.launch('camli_wino1d_gru_blend', lib.camli_wino1d_gru_blend, rh.data_ptr(), mn.data_ptr(), cx,
                                hub.wino_u(wp_q, False).data_ptr(), c_q.data_ptr(), z.data_ptr(), hcur.data_ptr(), hn.data_ptr(), q.data_ptr(),
                                int(half == 1), v_q.data_ptr() if keep else None, ws.data_ptr(), need, b, hh, ww, axis, _stream_ptr(h),
                                work=(4.0 * b * hh * ww * (6 * hd + cx) + 1.5 * need, 'B'), flop=2.0 * 8 * tiles * (hd + cx) * hd)
                else:
                    kept.append((None, None))
                    _lib.launch('camli_convcl_gru_gates', lib.camli_convcl_gru_gates, hcur.data_ptr(), mn.data_ptr(), cx, wp_zr.data_ptr(),
                                c_zr.data_ptr(), z.data_ptr(), rh.data_ptr(), r.data_ptr(), b, hh, ww, t, dy, dx, _stream_ptr(h),
                                work=(4.0 * b * hh * ww * (6 * hd + cx), 'B'), flop=flop * 2 * hd)
                    _lib.launch('camli_convcl_gru_blend', lib.camli_convcl_gru_blend, rh.data_ptr(), mn.data_ptr(), cx, wp_q.data_ptr(),
                                c_q.data_ptr(), z.data_ptr(), hcur.data_ptr(), hn.data_ptr(), q.data_ptr(), int(half == 1), b, hh, ww, t,
                                dy, dx, _stream_ptr(h), work=(4.0 * b * hh * ww * (6 * hd + cx), 'B'), flop=flop * hd)
                ctx.geom.append((geom, wpt_zr, wpt_q, wino))
                saved += [z, r, rh, q]
                if half == 0:
                    saved.append(hn)
                hcur = hn
        ctx.save_for_backward(*saved)
        return _to_nchw(hcur)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        h0, mn, z1, r1, rh1, q1, h1, z2, r2, rh2, q2 = ctx.saved_tensors
        hub = ctx.hub
        b, hh, ww, hd = h0.shape
        npix = b * hh * ww
        need_w = [w.requires_grad for w in hub.weights]
        need_c = [c.requires_grad for c in hub.contexts]
        gcur = _to_nhwc(g.float())             # gradient of the half-step's output
        gm = torch.empty_like(mn)
        first_m = True

        def total(slot, shape, want):
            """(running total to add into or NULL, tensor that becomes the total when there is none yet)"""
            if not want:
                return 0, None
            return (hub.gc[slot].data_ptr(), None) if hub.gc[slot] is not None else (0, slot)

        side = _runtime.wgrad_side(g.device) if hub.two_lane else None
        if side is not None:
            hub.side = side

        def wgrad(slot, xs, gpre, taps, khw, wino=False, v=None):
            """the weight gradient of one convolution, added into the pass total -- beside the data-gradient chain when a side
            stream is available (runtime.wgrad_side): nothing downstream needs it before the hub hands the totals over"""
            def run():
                if wino and _GRU_WINO_WRW:
                    return wino1d_wrw(xs, gpre, 0 if khw[0] == 1 else 1, out=hub.gw[slot], v=v)
                return convcl_wrw(xs, gpre, taps, khw, out=hub.gw[slot])
            if side is None:
                hub.gw[slot] = run()
                return
            side.fork(gpre, *xs, *([v] if v is not None else []))
            with side.stream():
                hub.gw[slot] = run()

        with _on_device(g):
            for half, (hin, z, r, rh, q) in ((1, (h1, z2, r2, rh2, q2)), (0, (h0, z1, r1, rh1, q1))):
                geom, wpt_zr, wpt_q, wino = ctx.geom[half]
                taps, ntaps = convcl_taps(*geom), convcl_taps(*geom, negate=True)
                axis = 0 if geom[0] == 1 else 1

                def data_gradient(gpre, wpt, out, accumulate):
                    """the convolution of the pre-activation gradient with the transposed, tap-reversed weights, split into
                    (hidden-side | motion) gradients"""
                    if wino:
                        wino1d_conv([gpre], hub.wino_u(wpt, True), axis, split=hd, out=out, accumulate=accumulate)
                    else:
                        convcl([gpre], wpt, ntaps, split=hd, out=out, accumulate=accumulate)
                izr, iq = 2 * half, 2 * half + 1
                gpre_q, gz, gh = torch.empty_like(h0), torch.empty_like(h0), torch.empty_like(h0)
                acc_ptr, becomes = total(iq, None, need_c[iq])
                _lib.launch('camli_gru_blend_bwd', lib.camli_gru_blend_bwd_acc, gcur.data_ptr(), z.data_ptr(), hin.data_ptr(), q.data_ptr(),
                            gpre_q.data_ptr(), gz.data_ptr(), gh.data_ptr(), acc_ptr, npix, hd, 1, int(half == 1), _stream_ptr(g),
                            work=(28.0 * npix * hd, 'B'))
                # q convolution: input gradient = (gradient of r h | + gradient of m), weight gradient
                grh = torch.empty_like(h0)
                data_gradient(gpre_q, wpt_q, (grh, gm), (False, not first_m))
                first_m = False
                if need_w[iq]:
                    wgrad(iq, [rh, mn], gpre_q, taps, geom[:2], wino, ctx.kept[half][1])
                if becomes is not None:      # the first contribution starts the total (a copy: gpre_q itself may still be read
                    hub.gc[iq] = gpre_q.clone()      # by the weight gradient on the side stream when later updates add into it)
                gpre_zr = torch.empty((b, hh, ww, 2 * hd), dtype=torch.float32, device=g.device)
                acc_ptr, becomes = total(izr, None, need_c[izr])
                _lib.launch('camli_gru_gates_bwd', lib.camli_gru_gates_bwd_into, gz.data_ptr(), hd, grh.data_ptr(), hd, z.data_ptr(),
                            r.data_ptr(), hin.data_ptr(), gpre_zr.data_ptr(), gh.data_ptr(), acc_ptr, npix, hd, 1, _stream_ptr(g),
                            work=(36.0 * npix * hd, 'B'))
                # z | r convolution: its input gradient completes the gradient of this half-step's hidden input and of m
                data_gradient(gpre_zr, wpt_zr, (gh, gm), (True, True))
                if need_w[izr]:
                    wgrad(izr, [hin, mn], gpre_zr, taps, geom[:2], wino, ctx.kept[half][0])
                if becomes is not None:
                    hub.gc[izr] = gpre_zr.clone()
                gcur = gh
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(g.device))
            hub.done = (ev, g.device)
        need = ctx.needs_input_grad
        return (_to_nchw(gcur) if need[0] else None, _to_nchw(gm) if need[1] else None, None, None)


def gru2d_step_supported(h, m, w_zr):
    return (h.is_cuda and h.dtype == torch.float32 and m.dtype == torch.float32 and h.shape[1] == 128 and m.shape[1] % 16 == 0
            and (h.shape[1] + m.shape[1]) % 256 == 0 and w_zr.dtype == torch.float32 and w_zr.shape[0] == 256
            and h.shape[0] * h.shape[2] * h.shape[3] * 256 * 4 < 0x7FF00000)


def gru2d_step_cl(h, m, hub):
    """One GRU2D update.  h [B,128,H,W], m [B,CX,H,W] (NCHW); hub = the pass's GRU2DPass (weights = the [h | m] blocks of the
    1x5 / 5x1 gates, contexts = the hoisted context terms as dense NHWC tensors)."""
    _require_cuda('gru2d_step_cl', h, m)
    return _GRU2DStepCL.apply(h, m, hub.token, hub)


class _BiasAct(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, bias, act):
        lib = _lib.load()
        if not x.is_contiguous():
            x = x.contiguous()
        b, c = x.shape[0], x.shape[1]
        p = x.numel() // (b * c)
        # relu / leaky_relu: keep one sign bit per element for the backward instead of re-reading y
        mask = None
        assert act != 5 or p % 4 == 0, 'relu_nan_to_num exists in the sign-mask form only'
        if act == 5 or (act in (1, 2) and p % 4 == 0 and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])):
            mask = torch.empty(lib.camli_bias_act_mask_bytes(b, c, p) // 8, dtype=torch.int64, device=x.device)
        with _on_device(x):
            _lib.launch('camli_bias_act_fwd', lib.camli_bias_act_fwd, x.data_ptr(), bias.data_ptr(),
                        mask.data_ptr() if mask is not None else None, b, c, p, act,
                        _stream_ptr(x), work=(8.0 * b * c * p + (b * c * p / 8.0 if mask is not None else 0.0), 'B'))
        ctx.mark_dirty(x)
        if mask is not None:
            ctx.save_for_backward(mask)
        elif act != 0:
            ctx.save_for_backward(x)
        ctx.act, ctx.masked, ctx.dims = act, mask is not None, (b, c, p)
        ctx.bias_param = _runtime.deferral_target(bias)
        return x

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        lib = _lib.load()
        saved = ctx.saved_tensors[0] if ctx.saved_tensors else None
        b, c, p = ctx.dims
        identity = ctx.act == 0
        gy = gy.float()
        gy_bs = None if identity else _batch_strided(gy)     # a channel slice of a wider gradient is read where it lies
        if gy_bs is None:
            gy = gy.contiguous()
            gy_bs = c * p
        gx = gy if identity else torch.empty(gy.shape, dtype=torch.float32, device=gy.device)   # identity: the input gradient IS gy
        deferred = ctx.bias_param is not None
        if identity and not (deferred or ctx.needs_input_grad[1]):
            return gx, None, None
        if deferred:      # the kernel's atomics accumulate straight into the parameter's per-pass buffer
            gbias = _runtime.PARAM_GRADS.slot(ctx.bias_param, lambda: _zero_slice(c, gy), False)
        else:
            gbias = _zero_slice(c, gy)
        with _on_device(gy):
            _lib.launch('camli_bias_act_bwd', lib.camli_bias_act_bwd_strided, gy.data_ptr(), gy_bs,
                        None if (ctx.masked or identity) else saved.data_ptr(), saved.data_ptr() if ctx.masked else None,
                        None if identity else gx.data_ptr(), gbias.data_ptr(), b, c, p, ctx.act, _stream_ptr(gy),
                        work=((4.0 if identity else (8.125 if ctx.masked else 12.0)) * b * c * p, 'B'))
        return gx, (None if deferred else gbias), None


class _BiasActCat(torch.autograd.Function):
    """cat([act_i(x_i + bias_i) ...] (+ [tail])) along the channels, every part written by its epilogue kernel straight into its
    slice of the result (camli_bias_act_into_fwd): the concatenation a following convolution reads exists without a cat pass.
    args = (x_0, bias_0, x_1, bias_1, ..., [tail]); acts = activation code per part.  The adjoint reads the slices of the
    incoming gradient in place (camli_bias_act_bwd_strided); the tail's gradient is its slice, as a view."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, acts, has_tail, *args):
        lib = _lib.load()
        n = len(acts)
        xs = [args[2 * i].contiguous() for i in range(n)]
        biases = [args[2 * i + 1] for i in range(n)]
        tail = args[2 * n] if has_tail else None
        b = xs[0].shape[0]
        spatial = tuple(xs[0].shape[2:])
        p = xs[0][0, 0].numel()
        chans = [x.shape[1] for x in xs]
        total = sum(chans) + (tail.shape[1] if tail is not None else 0)
        out = torch.empty((b, total) + spatial, dtype=torch.float32, device=xs[0].device)
        masks, c0 = [], 0
        with _on_device(out):
            for x, bias, act, c in zip(xs, biases, acts, chans):
                mask = None
                if act == 5 or (act in (1, 2) and p % 4 == 0):
                    mask = torch.empty(lib.camli_bias_act_mask_bytes(b, c, p) // 8, dtype=torch.int64, device=x.device)
                assert act in (0, 1, 2, 5) and (mask is not None or act == 0), 'bias_act_cat: relu-type parts on planes of 4k elements'
                _lib.launch('camli_bias_act_fwd', lib.camli_bias_act_into_fwd, x.data_ptr(), bias.data_ptr(),
                            mask.data_ptr() if mask is not None else None, out.data_ptr() + 4 * c0 * p, total * p, b, c, p, act,
                            _stream_ptr(x), work=(8.0 * b * c * p + (b * c * p / 8.0 if mask is not None else 0.0), 'B'))
                masks.append(mask)
                c0 += c
            if tail is not None:
                out[:, c0:].copy_(tail)
        ctx.save_for_backward(*[m for m in masks if m is not None])
        ctx.has_mask = [m is not None for m in masks]
        ctx.acts, ctx.chans, ctx.dims, ctx.has_tail = list(acts), chans, (b, p, total), has_tail
        ctx.bias_params = [_runtime.deferral_target(bias) for bias in biases]
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        b, p, total = ctx.dims
        gout = gout.float()
        if _batch_strided(gout) != total * p:
            gout = gout.contiguous()
        saved = list(ctx.saved_tensors)
        grads, c0 = [], 0
        spatial = tuple(gout.shape[2:])
        with _on_device(gout):
            for i, (act, c) in enumerate(zip(ctx.acts, ctx.chans)):
                mask = saved.pop(0) if ctx.has_mask[i] else None
                gslice = gout[:, c0:c0 + c]
                identity = act == 0
                deferred = ctx.bias_params[i] is not None
                need_bias = deferred or ctx.needs_input_grad[2 + 2 * i + 1]
                gx = gslice if identity else torch.empty((b, c) + spatial, dtype=torch.float32, device=gout.device)
                gbias = None
                if not identity or need_bias:
                    gbias = (_runtime.PARAM_GRADS.slot(ctx.bias_params[i], lambda c=c: _zero_slice(c, gout), False) if deferred
                             else _zero_slice(c, gout))
                    _lib.launch('camli_bias_act_bwd', lib.camli_bias_act_bwd_strided, gout.data_ptr() + 4 * c0 * p, total * p, None,
                                mask.data_ptr() if mask is not None else None, None if identity else gx.data_ptr(), gbias.data_ptr(),
                                b, c, p, act, _stream_ptr(gout), work=((4.0 if identity else 8.125) * b * c * p, 'B'))
                grads += [gx, None if (deferred or not need_bias) else gbias]
                c0 += c
        if ctx.has_tail:
            grads.append(gout[:, c0:])
        return (None, None, *grads)


def bias_act_cat(parts, tail=None):
    """parts: list of (x [B,C_i,...] fresh convolution output, bias [C_i], act name); tail: optional [B,C_t,...] appended
    unchanged.  -> cat([act(x_i + bias_i)..., tail], dim=1) without the cat pass (and without touching the x_i)."""
    _require_cuda('bias_act_cat', *[x for x, _, _ in parts])
    acts = tuple(ACT_CODES[a] for _, _, a in parts)
    flat = []
    for x, bias, _ in parts:
        flat += [x.float(), bias.float()]
    if tail is not None:
        flat.append(tail.float())
    return _BiasActCat.apply(acts, tail is not None, *flat)


def bias_act_cat_ok(parts, tail=None):
    """the planes are multiples of 4 elements (16-byte stores into the slices, sign masks) and the activations are the
    mask-backed ones"""
    p = parts[0][0][0, 0].numel()
    return p % 4 == 0 and all(ACT_CODES[a] in (0, 1, 2, 5) and x[0, 0].numel() == p and bias is not None for x, bias, a in parts)


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


class _BiasActNHWC(torch.autograd.Function):
    """act(x + bias[c] (+ res)) in place on a CHANNELS-LAST convolution output [B,C,H,W] (memory [B,H,W,C]); act 0 / 1.
    The ResNet trunk runs channels-last (MIOpen's implicit-GEMM kernels are NHWC kernels), see biasact.hip."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, bias, res, act):
        lib = _lib.load()
        b, c, h, w = x.shape
        n_pix = b * h * w
        if res is not None and not _is_nhwc(res):
            res = res.contiguous(memory_format=torch.channels_last)
        mask = None
        if act == 1 and any(ctx.needs_input_grad[:3]):
            mask = torch.empty(lib.camli_bias_act_nhwc_mask_bytes(n_pix, c) // 8, dtype=torch.int64, device=x.device)
        with _on_device(x):
            _lib.launch('camli_bias_act_fwd', lib.camli_bias_act_nhwc_fwd, x.data_ptr(), bias.data_ptr(),
                        res.data_ptr() if res is not None else None, mask.data_ptr() if mask is not None else None, n_pix, c, act,
                        _stream_ptr(x), work=((12.0 if res is not None else 8.0) * n_pix * c, 'B'))
        ctx.mark_dirty(x)
        if mask is not None:
            ctx.save_for_backward(mask)
        ctx.act, ctx.dims, ctx.has_res = act, (n_pix, c), res is not None
        ctx.bias_param = _runtime.deferral_target(bias)
        return x

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        lib = _lib.load()
        n_pix, c = ctx.dims
        gy = gy.float()
        if not _is_nhwc(gy):
            gy = gy.contiguous(memory_format=torch.channels_last)
        relu = ctx.act == 1
        if relu and not ctx.saved_tensors:      # forward ran without a mask (no gradient was expected)
            raise RuntimeError('bias_act (channels-last): backward without a saved sign mask')
        gx = torch.empty_like(gy) if relu else gy
        deferred = ctx.bias_param is not None
        gbias = None
        if deferred or ctx.needs_input_grad[1] or relu:
            gbias = (_runtime.PARAM_GRADS.slot(ctx.bias_param, lambda: _zero_slice(c, gy), False) if deferred
                     else _zero_slice(c, gy))
            ws = torch.empty(lib.camli_bias_act_nhwc_bwd_workspace_bytes(n_pix, c) // 4, dtype=torch.float32, device=gy.device)
            with _on_device(gy):
                _lib.launch('camli_bias_act_bwd', lib.camli_bias_act_nhwc_bwd, gy.data_ptr(),
                            ctx.saved_tensors[0].data_ptr() if relu else None, gx.data_ptr() if relu else None, gbias.data_ptr(),
                            ws.data_ptr(), n_pix, c, ctx.act, _stream_ptr(gy), work=((8.125 if relu else 4.0) * n_pix * c, 'B'))
        return gx, (None if deferred else gbias), (gx if ctx.has_res else None), None


def _nhwc_epilogue_ok(x, act):
    c = x.shape[1]
    return _is_nhwc(x) and act in (0, 1) and 4 <= c <= 1024 and (c & (c - 1)) == 0 and x.dtype == torch.float32


class _MaxPool3x3S2(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        b, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((b, c, ho, wo), dtype=torch.float32, device=x.device)
        arg = torch.empty((b, c, ho, wo), dtype=torch.uint8, device=x.device)
        with _on_device(x):
            _lib.launch('camli_maxpool3x3s2_fwd', lib.camli_maxpool3x3s2_fwd, x.data_ptr(), y.data_ptr(), arg.data_ptr(),
                        b * c, h, w, ho, wo, _stream_ptr(x), work=(b * c * (4.0 * h * w + 5.0 * ho * wo), 'B'))
        ctx.save_for_backward(arg)
        ctx.in_hw = (h, w)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        b, c, ho, wo = arg.shape
        h, w = ctx.in_hw
        gy = gy.contiguous().float()
        gx = torch.empty((b, c, h, w), dtype=torch.float32, device=gy.device)
        with _on_device(gy):
            _lib.launch('camli_maxpool3x3s2_bwd', lib.camli_maxpool3x3s2_bwd, gy.data_ptr(), arg.data_ptr(), gx.data_ptr(),
                        b * c, h, w, ho, wo, _stream_ptr(gy), work=(b * c * (4.0 * h * w + 5.0 * ho * wo), 'B'))
        return gx


def maxpool3x3s2(x):
    """nn.MaxPool2d(3, stride=2, padding=1) on [B,C,H,W] (the ResNet stem): one byte of arg-max per output, gather adjoint."""
    _require_cuda('maxpool3x3s2', x)
    assert x.dim() == 4
    return _MaxPool3x3S2.apply(x.float())


class _BiasActRes(torch.autograd.Function):
    """y = act(x + bias[c] + res) in place on x: the closing statement of a residual block in one pass.  The adjoint is the
    plain bias/activation adjoint; ``res`` receives the same gradient tensor as ``x``."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, bias, res, act):
        lib = _lib.load()
        if not x.is_contiguous():
            x = x.contiguous()
        res = res.contiguous()
        b, c = x.shape[0], x.shape[1]
        p = x.numel() // (b * c)
        mask = None
        if act == 1 and p % 4 == 0 and any(ctx.needs_input_grad[:3]):
            mask = torch.empty(lib.camli_bias_act_mask_bytes(b, c, p) // 8, dtype=torch.int64, device=x.device)
        with _on_device(x):
            _lib.launch('camli_bias_act_fwd', lib.camli_bias_act_res_fwd, x.data_ptr(), bias.data_ptr(), res.data_ptr(),
                        mask.data_ptr() if mask is not None else None, b, c, p, act, _stream_ptr(x),
                        work=(12.0 * b * c * p + (b * c * p / 8.0 if mask is not None else 0.0), 'B'))
        ctx.mark_dirty(x)
        if mask is not None:
            ctx.save_for_backward(mask)
        elif act != 0:
            ctx.save_for_backward(x)
        ctx.act, ctx.masked, ctx.dims = act, mask is not None, (b, c, p)
        ctx.bias_param = _runtime.deferral_target(bias)
        return x

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        gx, gbias, _ = _BiasAct.backward(ctx, gy)
        return gx, gbias, gx, None


def _fresh_fp32(x, keep_layout=False):
    """The tensor the in-place epilogues overwrite must be the very object handed to ``Function.apply`` (ctx.mark_dirty on
    anything else -- a dtype cast made by custom_fwd under autocast, a ``.contiguous()`` copy made inside forward -- is what
    torch warns it will stop tolerating): make those copies HERE, as ordinary differentiable operations."""
    if x.dtype != torch.float32:
        x = x.float()
    if not keep_layout and not x.is_contiguous():
        x = x.contiguous()
    return x


def bias_act_res(x, bias, res, act):
    """act(x + bias[c] + res) in place on the fresh convolution output x; act None or 'relu'; res shaped like x."""
    _require_cuda('bias_act_res', x, bias, res)
    assert act in (None, 'relu') and res.shape == x.shape
    if _nhwc_epilogue_ok(x, ACT_CODES[act]):
        return _BiasActNHWC.apply(_fresh_fp32(x, keep_layout=True), bias.float().contiguous(), res.float(), ACT_CODES[act])
    return _BiasActRes.apply(_fresh_fp32(x), bias.float(), res.float(), ACT_CODES[act])


def bias_act(x, bias, act):
    """act(x + bias[c]) in place on the (fresh) convolution output x [B,C,...]; ``act`` as in ACT_CODES."""
    _require_cuda('bias_act', x, bias)
    if _nhwc_epilogue_ok(x, ACT_CODES[act]):
        return _BiasActNHWC.apply(_fresh_fp32(x, keep_layout=True), bias.float().contiguous(), None, ACT_CODES[act])
    return _BiasAct.apply(_fresh_fp32(x), bias.float(), ACT_CODES[act])


# ------------------------------------------------------------------------------------------------
# two-channel 3x3 convolution heads (models/raft_core.py:169-181 FlowHead2D.conv2, PWC's conv_last)
# ------------------------------------------------------------------------------------------------
class _Conv3x3Co2(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        b, cin, h, w = x.shape
        y = torch.empty((b, 2, h, w), dtype=torch.float32, device=x.device)
        with _on_device(x):
            _lib.launch('camli_conv3x3_co2_fwd', lib.camli_conv3x3_co2_fwd, x.data_ptr(), weight.data_ptr(),
                        bias.data_ptr() if bias is not None else None, y.data_ptr(), b, cin, h, w, _stream_ptr(x),
                        work=(4.0 * b * h * w * (cin + 2), 'B'))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.w_param = _runtime.deferral_target(weight)
        ctx.b_param = _runtime.deferral_target(bias) if bias is not None else None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        b, cin, h, w = x.shape
        gy = gy.contiguous().float()
        gx = gw = gb = None
        with _on_device(x):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _lib.launch('camli_conv3x3_co2_bwd_data', lib.camli_conv3x3_co2_bwd_data, gy.data_ptr(), weight.data_ptr(),
                            gx.data_ptr(), b, cin, h, w, _stream_ptr(x), work=(4.0 * b * h * w * (cin + 2), 'B'))
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1] or want_b:
                ws = torch.empty(lib.camli_conv3x3_co2_bwd_weight_workspace_bytes(b, cin, w) // 4, dtype=torch.float32, device=x.device)
                deferred = ctx.w_param is not None and (not want_b or ctx.b_param is not None)
                if deferred:      # iteration-shared parameters: the kernel adds into the per-pass accumulators
                    gw_buf = _runtime.PARAM_GRADS.slot(ctx.w_param, lambda: torch.zeros_like(weight), False)
                    gb_buf = _runtime.PARAM_GRADS.slot(ctx.b_param, lambda: _zero_slice(2, gy), False) if want_b else None
                else:
                    gw_buf = torch.empty_like(weight)
                    gb_buf = torch.empty(2, dtype=torch.float32, device=x.device) if want_b else None
                _lib.launch('camli_conv3x3_co2_bwd_weight', lib.camli_conv3x3_co2_bwd_weight, gy.data_ptr(), x.data_ptr(),
                            ws.data_ptr(), gw_buf.data_ptr(), gb_buf.data_ptr() if gb_buf is not None else None,
                            1 if deferred else 0, b, cin, h, w, _stream_ptr(x), work=(4.0 * b * h * w * (cin + 2), 'B'))
                if not deferred:
                    gw, gb = gw_buf, gb_buf
        return gx, gw, gb


def conv3x3_co2_supported(conv, x):
    """A plain 3x3 / stride 1 / padding 1 convolution to exactly two channels on a contiguous fp32 NCHW map."""
    return (isinstance(conv, torch.nn.Conv2d) and conv.out_channels == 2 and tuple(conv.kernel_size) == (3, 3)
            and tuple(conv.stride) == (1, 1) and conv.padding == (1, 1) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and conv.padding_mode == 'zeros' and conv.in_channels <= 640 and x.dim() == 4 and x.is_cuda
            and x.shape[2] <= 65535 and x.shape[0] <= 65535 and x.shape[1] * x.shape[2] * x.shape[3] < 2 ** 31)


def conv3x3_co2(x, weight, bias):
    """y = conv2d(x, weight, bias, padding=1) for weight [2,Cin,3,3]: three HBM-bound kernels instead of an implicit
    GEMM with N = 2 (csrc/hip/smallconv.hip)."""
    _require_cuda('conv3x3_co2', x, weight)
    assert weight.shape[0] == 2 and weight.shape[2:] == (3, 3) and weight.shape[1] == x.shape[1]
    return _Conv3x3Co2.apply(x.float().contiguous(), weight.float().contiguous(), bias.float() if bias is not None else None)


# ------------------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) on the fp32 matrix cores (csrc/hip/winograd.hip, round 6): the 3x3 / stride 1 / padding 1 convolutions of
# the RAFT update block -- MotionEncoder2D.conv_c2 / conv (models/raft_core.py:148,151), FlowHead2D.conv1 (:173), the mask
# head's first convolution (:188) -- forward, data gradient (the same three launches on the output gradient with the
# transposed, tap-reversed weights) and weight gradient.  CAMLI_WINO=0 leaves them with the library (A/B).
# ------------------------------------------------------------------------------------------------
_WINO = os.environ.get('CAMLI_WINO', '1') != '0'
# output tile of the product's Winograd convolutions: F(4x4,3x3) (36 multiplications per 16 outputs: 4 x fewer than the direct
# form, transform-domain tensors 2.25 x the image-domain ones) or F(2x2,3x3) (16 per 4: 2.25 x fewer, 4 x the bytes, a tenth
# of the rounding error).  CAMLI_WINO_TILE=2 | 4.
_WINO_TILE = int(os.environ.get('CAMLI_WINO_TILE', '4'))
assert _WINO_TILE in (2, 4), 'CAMLI_WINO_TILE must be 2 or 4'
_wino_weights = {}          # id(weight) -> [weakref(weight), version, {(flip, tile): U}]


def wino_supported(conv, x):
    """A plain 3x3 / stride 1 / padding 1 convolution on an fp32 NCHW map whose channel counts fill the matrix-core tiles."""
    return (_WINO and isinstance(conv, torch.nn.Conv2d) and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
            and conv.padding == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and wino_shape_supported(conv.in_channels, conv.out_channels) and x.dim() == 4 and x.is_cuda
            and x.dtype == torch.float32 and conv.weight.dtype == torch.float32 and _runtime.own_kernels_allowed())


def wino_shape_supported(cin, cout):
    """Both directions run on the plane GEMMs (contraction over Cin forward, over Cout in the data gradient, each padded to
    a multiple of 16).  Fewer than 96 rows on a GEMM's M side leave its 128-row tile mostly padding (MotionEncoder2D.conv_f2,
    128 -> 64: measured 123 us against the library's 95): those stay with the library."""
    return cin >= 96 and cout >= 96


def wino_transformed_weights(w, flip, tile=None):
    """U [P][Kp][Mp] of w [Cout][Cin][3][3] (camli_wino_weights), computed once per value of the weight tensor: the 12 GRU
    iterations of a pass (and its backward) share it; an optimiser step bumps the tensor's version."""
    tile = tile or _WINO_TILE
    key = id(w)
    entry = _wino_weights.get(key)
    if entry is None or entry[0]() is not w or entry[1] != w._version:
        if len(_wino_weights) > 64:
            for k in [k for k, e in _wino_weights.items() if e[0]() is None]:
                del _wino_weights[k]
        entry = _wino_weights[key] = [weakref.ref(w), w._version, {}]
    u = entry[2].get((flip, tile))
    if u is None:
        lib = _lib.load()
        cout, cin = w.shape[0], w.shape[1]
        k, m = (cout, cin) if flip else (cin, cout)
        u = torch.empty(lib.camli_wino_weight_floats(k, m, tile), dtype=torch.float32, device=w.device)
        wd = w.detach()
        wd = wd if wd.is_contiguous() else wd.contiguous()
        with _on_device(w):
            _lib.launch('camli_wino_weights', lib.camli_wino_weights, wd.data_ptr(), u.data_ptr(), cout, cin, int(flip), tile, _stream_ptr(w),
                        work=(4.0 * (9 + (tile + 2) ** 2) * cout * cin, 'B'))
        u._camli_stream = torch.cuda.current_stream(w.device)
        u._camli_seen = {u._camli_stream.cuda_stream}
        u._camli_tile = tile
        entry[2][(flip, tile)] = u
    else:
        # produced on another stream (a Branch, the priming pass's single lane): order this stream behind the producer ONCE --
        # the tensor never changes afterwards, and waiting on every use would serialise an auxiliary stream behind the main
        # one for good where the weights stay put (inference)
        cur = torch.cuda.current_stream(w.device)
        if cur.cuda_stream not in u._camli_seen:
            cur.wait_stream(u._camli_stream)
            u._camli_seen.add(cur.cuda_stream)
    return u


def _image_stride(t):
    """t [B,C,H,W] fp32 whose images are dense [C,H,W] blocks a fixed stride apart (a contiguous tensor or a channel slice of
    one) -> that stride in floats, else None.  No alignment demanded: the Winograd transforms fall back to 4-byte accesses."""
    if t.dtype != torch.float32 or t.dim() != 4:
        return None
    inner = 1
    for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if size != 1 and stride != inner:
            return None
        inner *= size
    bs = t.stride(0) if t.shape[0] > 1 else inner
    return bs if bs >= inner else None


def wino_mask_bits(b, c, hh, ww, device):
    """An (uninitialised) activation-bit tensor for a [b,c,hh,ww] map: [b, c, hh, ceil(ww / 8)] bytes, bit j of byte s = pixel
    8 s + j (camli_wino_mask_bytes)."""
    return torch.empty((b, c, hh, (ww + 7) // 8), dtype=torch.uint8, device=device)


def wino_pack_bits(mask):
    """bool / 0-1 tensor [B,C,H,W] -> activation bits in the kernels' format (tests; the product's bits come from the output
    transform)."""
    b, c, hh, ww = mask.shape
    pad = (-ww) % 8
    m = torch.nn.functional.pad(mask.to(torch.uint8), (0, pad)).view(b, c, hh, -1, 8)
    weights = (2 ** torch.arange(8, device=mask.device)).to(torch.uint8)
    return (m * weights).sum(-1).to(torch.uint8).contiguous()


def _wino_tiles(b, hh, ww, tile):
    """transform-domain row length NT of a [b, ., hh, ww] map (winograd.h make_geometry)"""
    tiles = b * ((hh + tile - 1) // tile) * (((ww + 7) // 8) * (8 // tile))
    return (tiles + 15) // 16 * 16


def wino_conv3x3(x, u, n_out, bias=None, act=None, out=None, accumulate=False, bits=None, bits_out=None):
    """act(conv3x3(x) + bias) for pre-transformed weights u (wino_transformed_weights).  x [B,C,H,W] fp32, dense or a channel
    slice of a dense NCHW tensor; ``bits``: activation bits of x (wino_mask_bits geometry), x reads as zero where its bit is
    clear; ``out``: an existing [B,n_out,H,W] tensor (or channel slice), ``accumulate``: add into it; ``bits_out``: receives the
    activation bits of the output (act 'relu' / 'relu_nan_to_num')."""
    _require_cuda('wino_conv3x3', x, u)
    lib = _lib.load()
    b, c, hh, ww = x.shape
    xbs = _image_stride(x)
    if xbs is None:
        x = x.contiguous()
        xbs = c * hh * ww
    w8 = (ww + 7) // 8
    assert bits is None or (bits.dtype == torch.uint8 and bits.shape == (b, c, hh, w8) and bits.is_contiguous())
    assert bits_out is None or (act is not None and bits_out.dtype == torch.uint8 and bits_out.shape == (b, n_out, hh, w8)
                                and bits_out.is_contiguous())
    if out is None:
        assert not accumulate
        out = torch.empty((b, n_out, hh, ww), dtype=torch.float32, device=x.device)
    ybs = _image_stride(out)
    if ybs is None or out.shape != (b, n_out, hh, ww):
        raise _lib.CamliHipError('wino_conv3x3: the output must be a dense fp32 [B,%d,H,W] tensor or a channel slice of one' % n_out)
    tile = getattr(u, '_camli_tile', _WINO_TILE)           # the tile the weights were transformed for
    need = lib.camli_wino_workspace_bytes(b, c, n_out, hh, ww, tile)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    planes, tiles = (tile + 2) ** 2, _wino_tiles(b, hh, ww, tile)
    with _on_device(x):
        _lib.launch('camli_wino_conv3x3', lib.camli_wino_conv3x3, x.data_ptr(), xbs, bits.data_ptr() if bits is not None else None,
                    u.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ybs,
                    bits_out.data_ptr() if bits_out is not None else None, ws.data_ptr(), need, b, c,
                    n_out, hh, ww, {None: 0, 'relu': 1, 'relu_nan_to_num': 2}[act], int(bool(accumulate)), tile, _stream_ptr(x),
                    work=(4.0 * b * hh * ww * (c + n_out) + 2 * need, 'B'), flop=2.0 * planes * tiles * c * n_out)
    return out


def wino_wrw(x, gy, bits=None, out=None, gbias=None, gbias_accumulate=False, tile=None):
    """Weight gradient [N,C,3,3] of conv3x3(x) for the output gradient gy [B,N,H,W], contracted in the Winograd domain
    (camli_wino_wrw); ``bits``: the forward's activation bits, gy reads as zero where its bit is clear; ``out``: add into this
    tensor instead of creating one; ``gbias`` [N]: also write (or, ``gbias_accumulate``, add) the bias gradient = the
    per-channel sum of the masked gy."""
    _require_cuda('wino_wrw', x, gy)
    lib = _lib.load()
    b, c, hh, ww = x.shape
    n = gy.shape[1]
    assert gy.shape == (b, n, hh, ww)
    xbs, gbs = _image_stride(x), _image_stride(gy)
    if xbs is None:
        x, xbs = x.contiguous(), c * hh * ww
    if gbs is None:
        gy, gbs = gy.contiguous(), n * hh * ww
    assert bits is None or (bits.dtype == torch.uint8 and bits.shape == (b, n, hh, (ww + 7) // 8) and bits.is_contiguous())
    tile = tile or _WINO_TILE
    need = lib.camli_wino_wrw_workspace_bytes(b, c, n, hh, ww, tile)
    if need <= 0:
        raise _lib.CamliHipError('wino_wrw: unsupported shape B=%d C=%d N=%d %dx%d' % (b, c, n, hh, ww))
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = out if out is not None else torch.empty((n, c, 3, 3), dtype=torch.float32, device=x.device)
    assert gw.shape == (n, c, 3, 3) and gw.is_contiguous() and gw.dtype == torch.float32
    planes, tiles = (tile + 2) ** 2, _wino_tiles(b, hh, ww, tile)
    with _on_device(x):
        _lib.launch('camli_wino_wrw', lib.camli_wino_wrw, x.data_ptr(), xbs, gy.data_ptr(), gbs, bits.data_ptr() if bits is not None else None,
                    gw.data_ptr(), gbias.data_ptr() if gbias is not None else None, ws.data_ptr(), need, b, c, n, hh, ww,
                    int(out is not None), int(bool(gbias_accumulate)), tile, _stream_ptr(x),
                    work=(4.0 * b * hh * ww * (c + n) + 2.0 * 4 * planes * tiles * (c + n), 'B'), flop=2.0 * planes * tiles * c * n)
    return gw


# CAMLI_WINO_WRW=lib: the weight gradients of the Winograd convolutions on the library (A/B)
_WINO_WRW = os.environ.get('CAMLI_WINO_WRW', 'hip') != 'lib'


class _Conv3x3Wino(torch.autograd.Function):
    """conv2d(x, w, padding=1) without bias (the epilogue kernels add it)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return wino_conv3x3(x, wino_transformed_weights(w, False), w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.float()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = wino_conv3x3(gy, wino_transformed_weights(w, True), w.shape[1])
        if ctx.needs_input_grad[1]:
            if _WINO_WRW:
                gw = wino_wrw(x, gy)
            else:
                gw = torch.ops.aten.convolution_backward(gy.contiguous(), x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return gx, gw


def conv3x3_wino(x, weight):
    return _Conv3x3Wino.apply(x, weight)


class _WinoConvCat(torch.autograd.Function):
    """cat([act(conv3x3(x, w) + bias), act_i(raw_i + bias_i) ..., (tail)], dim=1) as ONE node: the Winograd output transform
    adds the bias, applies the activation and writes its channels straight into the concatenation; the other parts (fresh
    convolution outputs of other kernels) are written by their epilogue kernel as in _BiasActCat.  In the adjoint the ReLU mask
    (the saved output > 0) rides on the loads of the two transforms of the output gradient -- no masked copy of it exists --
    and the bias gradient comes out of the transform-domain plane that holds the tile sums.  act: None | 'relu' |
    'relu_nan_to_num'.  args = (x, w, bias, raw_0, bias_0, ..., [tail])."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, act, other_acts, has_tail, x, w, bias, *rest):
        lib = _lib.load()
        k = len(other_acts)
        raws = [rest[2 * i].contiguous() for i in range(k)]
        biases = [rest[2 * i + 1] for i in range(k)]
        tail = rest[2 * k] if has_tail else None
        b, _, hh, ww = x.shape
        p = hh * ww
        n = w.shape[0]
        chans = [r.shape[1] for r in raws]
        total = n + sum(chans) + (tail.shape[1] if has_tail else 0)
        out = torch.empty((b, total, hh, ww), dtype=torch.float32, device=x.device)
        need_bits = act is not None and any(ctx.needs_input_grad[3:6])
        bits = wino_mask_bits(b, n, hh, ww, x.device) if need_bits else None
        wino_conv3x3(x, wino_transformed_weights(w, False), n, bias=bias, act=act, out=out[:, :n], bits_out=bits)
        masks, c0 = [], n
        with _on_device(out):
            for raw, ob, oact, c in zip(raws, biases, other_acts, chans):
                mask = None
                if oact != 0:
                    assert oact in (1, 2, 5) and p % 4 == 0, 'wino_conv_cat: relu-type parts on planes of 4k elements'
                    mask = torch.empty(lib.camli_bias_act_mask_bytes(b, c, p) // 8, dtype=torch.int64, device=x.device)
                _lib.launch('camli_bias_act_fwd', lib.camli_bias_act_into_fwd, raw.data_ptr(), ob.data_ptr(),
                            mask.data_ptr() if mask is not None else None, out.data_ptr() + 4 * c0 * p, total * p, b, c, p, oact,
                            _stream_ptr(x), work=(8.0 * b * c * p + (b * c * p / 8.0 if mask is not None else 0.0), 'B'))
                masks.append(mask)
                c0 += c
            if tail is not None:
                out[:, c0:].copy_(tail)
        ctx.save_for_backward(x, w, bits, *[m for m in masks if m is not None])
        ctx.dims = (b, total, hh, ww)
        ctx.has_mask = [m is not None for m in masks]
        ctx.act, ctx.other_acts, ctx.chans, ctx.has_tail = act, list(other_acts), chans, has_tail
        ctx.w_param, ctx.b_param = _runtime.deferral_target(w), _runtime.deferral_target(bias)
        ctx.bias_params = [_runtime.deferral_target(ob) for ob in biases]
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gout):
        lib = _lib.load()
        x, w, bits = ctx.saved_tensors[:3]
        saved = list(ctx.saved_tensors[3:])
        b, total, hh, ww = ctx.dims
        p = hh * ww
        n, cin = w.shape[0], w.shape[1]
        gout = gout.float()
        if _batch_strided(gout) != total * p:
            gout = gout.contiguous()
        gs = gout[:, :n]
        gx = gw = gb = None
        if ctx.needs_input_grad[3]:
            gx = wino_conv3x3(gs, wino_transformed_weights(w, True), cin, bits=bits)
        # iteration-shared parameters accumulate in their per-pass buffers (runtime.PARAM_GRADS) and reach .grad once
        acc_w = _runtime.PARAM_GRADS.slot(ctx.w_param, lambda: torch.zeros_like(w), False) if ctx.w_param is not None else None
        acc_b = _runtime.PARAM_GRADS.slot(ctx.b_param, lambda: _zero_slice(n, gout), False) if ctx.b_param is not None else None
        need_w = ctx.needs_input_grad[4] or acc_w is not None
        need_b = ctx.needs_input_grad[5] or acc_b is not None
        if need_w or need_b:
            if need_b and acc_b is None:
                gb = torch.empty(n, dtype=torch.float32, device=gout.device)
            if _WINO_WRW:
                gw = wino_wrw(x, gs, bits=bits, out=acc_w, gbias=(acc_b if acc_b is not None else gb) if need_b else None,
                              gbias_accumulate=acc_b is not None)
                if acc_w is not None:
                    gw = None
            else:
                gm = (gs if bits is None else gs * _wino_unpack_bits(bits, ww)).contiguous()
                gw = torch.ops.aten.convolution_backward(gm, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
                if acc_w is not None:
                    acc_w.add_(gw)
                    gw = None
                if need_b:
                    if acc_b is not None:
                        acc_b.add_(gm.sum((0, 2, 3)))
                    else:
                        gb = gm.sum((0, 2, 3))
        grads, c0 = [], n
        with _on_device(gout):
            for i, (oact, c) in enumerate(zip(ctx.other_acts, ctx.chans)):
                mask = saved.pop(0) if ctx.has_mask[i] else None
                identity = oact == 0
                deferred = ctx.bias_params[i] is not None
                need_bias = deferred or ctx.needs_input_grad[6 + 2 * i + 1]
                g_raw = gout[:, c0:c0 + c] if identity else torch.empty((b, c, hh, ww), dtype=torch.float32, device=gout.device)
                gbias = None
                if not identity or need_bias:
                    gbias = (_runtime.PARAM_GRADS.slot(ctx.bias_params[i], lambda c=c: _zero_slice(c, gout), False) if deferred
                             else _zero_slice(c, gout))
                    _lib.launch('camli_bias_act_bwd', lib.camli_bias_act_bwd_strided, gout.data_ptr() + 4 * c0 * p, total * p, None,
                                mask.data_ptr() if mask is not None else None, None if identity else g_raw.data_ptr(), gbias.data_ptr(),
                                b, c, p, oact, _stream_ptr(gout), work=((4.0 if identity else 8.125) * b * c * p, 'B'))
                grads += [g_raw, None if (deferred or not need_bias) else gbias]
                c0 += c
        if ctx.has_tail:
            grads.append(gout[:, c0:])
        return (None, None, None, gx, gw, gb, *grads)


def _wino_unpack_bits(bits, ww):
    """activation bits -> a 0 / 1 float tensor [B,C,H,W] (the library fall-back of the adjoint, CAMLI_WINO_WRW=lib)"""
    shifts = torch.arange(8, device=bits.device, dtype=torch.uint8)
    m = ((bits.unsqueeze(-1) >> shifts) & 1).flatten(3)[..., :ww]
    return m.to(torch.float32)


def wino_conv_cat(x, conv, act, others=(), tail=None):
    """cat([act(conv(x)), act_i(raw_i + bias_i) for (raw_i, bias_i, act_i) in others, tail], dim=1) for a convolution that
    ``wino_supported`` accepts: bias, activation and the concatenation ride on the Winograd output transform (no pass over the
    convolution output), the ReLU adjoint on the input transforms of the backward."""
    _require_cuda('wino_conv_cat', x)
    assert act in (None, 'relu', 'relu_nan_to_num')
    flat = []
    for raw, ob, _ in others:
        flat += [raw.float(), ob.float()]
    if tail is not None:
        flat.append(tail.float())
    bias = conv.bias if conv.bias is not None else torch.zeros(conv.out_channels, dtype=torch.float32, device=x.device)
    return _WinoConvCat.apply(act, tuple(ACT_CODES[a] for _, _, a in others), tail is not None, x, conv.weight, bias, *flat)


# CAMLI_WINO_EPILOGUE=0: bias / activation / concatenation of the Winograd convolutions as separate epilogue passes (A/B)
_WINO_EPILOGUE = os.environ.get('CAMLI_WINO_EPILOGUE', '1') != '0'


def wino_epilogue_ok(conv, x, act):
    return _WINO_EPILOGUE and act in (None, 'relu', 'relu_nan_to_num') and wino_supported(conv, x)


# ------------------------------------------------------------------------------------------------
# input side: models/ids.py:4-33, models/camliraft.py:38-46
# ------------------------------------------------------------------------------------------------
def persp2paral_pair(pcs, intrinsics, persp, paral):
    """pcs [B,6,N] (both clouds), intrinsics [B,3] = (f, cx, cy) -> (pc1, pc2) in the parallel camera, one launch
    (no autograd: the clouds are inputs)."""
    _require_cuda('persp2paral_pair', pcs, intrinsics)
    lib = _lib.load()
    pcs, intrinsics = pcs.float().contiguous(), intrinsics.float().contiguous()
    b, six, n = pcs.shape
    assert six == 6 and intrinsics.shape == (b, 3)
    ratio_w = (paral['sensor_w'] - 1) / (persp['sensor_w'] - 1)
    ratio_h = (paral['sensor_h'] - 1) / (persp['sensor_h'] - 1)
    out1 = torch.empty((b, 3, n), dtype=torch.float32, device=pcs.device)
    out2 = torch.empty_like(out1)
    with _on_device(pcs):
        _lib.launch('camli_persp2paral', lib.camli_persp2paral, pcs.data_ptr(), intrinsics.data_ptr(), out1.data_ptr(),
                    out2.data_ptr(), b, n, float(ratio_w), float(ratio_h), float(min(ratio_w, ratio_h)),
                    float((paral['sensor_w'] - 1) / 2), float((paral['sensor_h'] - 1) / 2), _stream_ptr(pcs),
                    work=(48.0 * b * n, 'B'))
    return out1, out2


def project_pc2image(pc, camera_info, scale=(1.0, 1.0)):
    """pc [B,3,N] -> uv [B,2,N]: models/utils.py:234-259 times ``scale`` = (grid - 1) / (sensor - 1) per axis, one
    launch (no autograd: the clouds are inputs / detached)."""
    _require_cuda('project_pc2image', pc)
    lib = _lib.load()
    pc = pc.float().contiguous()
    b, three, n = pc.shape
    assert three == 3
    uv = torch.empty((b, 2, n), dtype=torch.float32, device=pc.device)
    mode = camera_info['projection_mode']
    if mode == 'perspective':
        intr = torch.stack([camera_info['f'], camera_info['cx'], camera_info['cy']], dim=1).float().contiguous()
        args = (intr.data_ptr(), uv.data_ptr(), b, n, 1, 0.0, 0.0)
    elif mode == 'parallel':
        args = (None, uv.data_ptr(), b, n, 0, float(camera_info['cx']), float(camera_info['cy']))
    else:
        raise NotImplementedError(mode)
    with _on_device(pc):
        _lib.launch('camli_project_pc2image', lib.camli_project_pc2image, pc.data_ptr(), *args, float(scale[0]), float(scale[1]),
                    _stream_ptr(pc), work=((12.0 if mode == 'perspective' else 8.0) * b * n + 8.0 * b * n, 'B'))
    return uv


def pad_normalize(images, pad, mean, std):
    """images [B,6,H,W] -> (image1, image2) [B,3,Hp,Wp]: replicate padding ``pad`` = [left, right, top(0), bottom] as
    InputPadder builds it, then (x - mean[c]) / std[c]; one pass over both frames."""
    _require_cuda('pad_normalize', images)
    lib = _lib.load()
    images = images.float().contiguous()
    b, six, h, w = images.shape
    left, right, top, bottom = pad
    assert six == 6 and top == 0
    hp, wp = h + bottom, w + left + right
    out1 = torch.empty((b, 3, hp, wp), dtype=torch.float32, device=images.device)
    out2 = torch.empty_like(out1)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    with _on_device(images):
        _lib.launch('camli_pad_normalize', lib.camli_pad_normalize, images.data_ptr(), out1.data_ptr(), out2.data_ptr(),
                    b, h, w, hp, wp, left, m3, s3, _stream_ptr(images), work=(4.0 * b * 6 * (h * w + hp * wp), 'B'))
    return out1, out2

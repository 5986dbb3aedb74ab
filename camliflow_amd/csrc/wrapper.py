"""The operator boundary of the hot path.

Mirror of the reference's ``models/csrc/wrapper.py`` (same function names, argument meaning, layout
sniffing, asserts and return types) with the native symbols replaced by the C-ABI entry points of
``libcamli_hip.so``.  Differences by design:

* no silent degradation: the reference turns all four natives off when one import fails
  (wrapper.py:4-15) and falls back to Python; here a missing library or a non-CUDA tensor raises.
* kernels run on torch's CURRENT stream (the reference launches on the legacy default stream).
* the correlation backward writes NHWC gradients directly (reference: NCHW + permute + copy,
  wrapper.py:34-35).
* ``cpp_impl=False`` keeps its reference meaning -- "compose the op from torch primitives"
  (wrapper.py:41-50,83-96,115-117) -- and runs on whatever device the tensors live on.  It is the
  reference's alternative formulation, not an oracle: KNN through ``topk`` differs from the native
  semantics on near-ties exactly as it does in the reference.
"""
import torch
import torch.nn.functional

from . import _lib


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_ptr(t):
    """hipStream_t of torch's current stream on t's device (the raw getter skips the Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on_device(t):
    """Device guard for a launch: a no-op when t already lives on the current device (one process per
    GPU: always, after the first torch.cuda.set_device), torch.cuda.device(t.device) otherwise."""
    if t.device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(t.device)


_zero_arena = {}            # device -> [chunk, used]
_ARENA_FLOATS = 1 << 20


def _zero_slice(n, like):
    """n zero floats for a kernel that ACCUMULATES into its (small) output: carved from a pre-zeroed
    4 MB chunk instead of one fill launch per call; every slice is handed out once, a fresh chunk is
    zeroed (and synchronised, so any stream may use it) when the current one is used up."""
    if n > 4096 or torch.cuda.is_current_stream_capturing():
        return torch.zeros(n, dtype=torch.float32, device=like.device)
    state = _zero_arena.get(like.device)
    if state is None or state[1] + n > _ARENA_FLOATS:
        state = _zero_arena[like.device] = [torch.zeros(_ARENA_FLOATS, dtype=torch.float32, device=like.device), 0]
        torch.cuda.current_stream(like.device).synchronize()
    start = state[1]
    state[1] = start + ((n + 31) // 32) * 32          # 128-byte aligned slices
    return state[0][start:start + n]


def _require_cuda(name, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise _lib.CamliHipError(
                "%s: expected CUDA (ROCm) tensors, got device '%s'.  camliflow_amd has no CPU path; "
                "pass cpp_impl=False for the reference's torch-composed formulation." % (name, t.device))


class CorrelationFunction(torch.autograd.Function):
    """Counterpart of wrapper.py:18-37.  Inputs are NHWC, the cost volume is NCHW."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input1, input2, max_displacement):
        lib = _lib.load()
        assert input1.is_contiguous() and input2.is_contiguous(), 'inputs must be contiguous (correlation.cpp:12-13)'
        assert input1.shape == input2.shape and input1.dtype == torch.float32 and input2.dtype == torch.float32
        ctx.save_for_backward(input1, input2)
        ctx.max_displacement = max_displacement
        b, h, w, c = input1.shape
        d = 2 * max_displacement + 1
        output = torch.empty((b, d * d, h, w), dtype=torch.float32, device=input1.device)
        with _on_device(input1):
            _lib.launch('camli_corr2d_fwd', lib.camli_corr2d_fwd, input1.data_ptr(), input2.data_ptr(), output.data_ptr(),
                                            b, c, h, w, max_displacement, _stream_ptr(input1),
                        work=(4.0 * b * h * w * (2 * c + d * d), 'B'))
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        lib = _lib.load()
        input1, input2 = ctx.saved_tensors
        b, h, w, c = input1.shape
        grad_output = grad_output.contiguous().float()
        grad_input1 = torch.empty_like(input1)
        grad_input2 = torch.empty_like(input2)
        with _on_device(input1):
            _lib.launch('camli_corr2d_bwd', lib.camli_corr2d_bwd, grad_output.data_ptr(), input1.data_ptr(), input2.data_ptr(),
                                            grad_input1.data_ptr(), grad_input2.data_ptr(),
                                            b, c, h, w, ctx.max_displacement, _stream_ptr(input1),
                        work=(4.0 * b * h * w * ((2 * ctx.max_displacement + 1) ** 2 + 4 * c), 'B'))
        return grad_input1, grad_input2, None


def correlation2d(input1: torch.Tensor, input2: torch.Tensor, max_displacement: int, cpp_impl=True):
    """Local cost volume, [B,C,H,W] x [B,C,H,W] -> [B,(2md+1)^2,H,W] (wrapper.py:40-57)."""
    def _correlation_composed(_input1, _input2, _md):
        height, width = _input1.shape[2:]
        padded = torch.nn.functional.pad(_input2, [_md] * 4)
        planes = [torch.mean(_input1 * padded[:, :, i:i + height, j:j + width], 1, keepdim=True)
                  for i in range(2 * _md + 1) for j in range(2 * _md + 1)]
        return torch.cat(planes, 1)

    if not cpp_impl:
        return _correlation_composed(input1, input2, max_displacement)
    _require_cuda('correlation2d', input1, input2)
    input1 = input1.permute(0, 2, 3, 1).contiguous().float()
    input2 = input2.permute(0, 2, 3, 1).contiguous().float()
    return CorrelationFunction.apply(input1, input2, max_displacement)


def squared_distance(xyz1: torch.Tensor, xyz2: torch.Tensor):
    """Pairwise squared distances [B,n1,n2] via |a|^2 + |b|^2 - 2ab (wrapper.py:60-72)."""
    assert xyz1.shape[-1] == xyz2.shape[-1] and xyz1.shape[-1] <= 3  # channel-last
    batch_size, n_points1, n_points2 = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    dist = -2 * torch.matmul(xyz1, xyz2.permute(0, 2, 1))
    dist += torch.sum(xyz1 ** 2, -1).view(batch_size, n_points1, 1)
    dist += torch.sum(xyz2 ** 2, -1).view(batch_size, 1, n_points2)
    return dist


def furthest_point_sampling(xyz: torch.Tensor, n_samples: int, cpp_impl=True):
    """FPS from seed index 0: [B,N,3] -> int64 [B,n_samples] (wrapper.py:75-103)."""
    def _fps_composed(_xyz, _n):
        batch_size, n_points, _ = _xyz.shape
        picks = torch.zeros(batch_size, _n, dtype=torch.int64, device=_xyz.device)
        dists = torch.full((batch_size, n_points), 1e10, device=_xyz.device)
        rows = torch.arange(batch_size, dtype=torch.int64, device=_xyz.device)
        cur = torch.zeros(batch_size, dtype=torch.int64, device=_xyz.device)
        for i in range(_n):
            picks[:, i] = cur
            centre = _xyz[rows, cur, :].view(batch_size, 1, 3)
            dists = torch.minimum(dists, torch.sum((_xyz - centre) ** 2, -1))
            cur = torch.max(dists, -1)[1]
        return picks

    assert xyz.shape[2] == 3 and xyz.shape[1] > n_samples
    if not cpp_impl:
        return _fps_composed(xyz, n_samples).to(torch.int64)
    _require_cuda('furthest_point_sampling', xyz)
    lib = _lib.load()
    xyz = xyz.contiguous().float()
    b, n, _ = xyz.shape
    out = torch.empty((b, n_samples), dtype=torch.int64, device=xyz.device)
    with _on_device(xyz):
        _lib.launch('camli_fps', lib.camli_fps, xyz.data_ptr(), out.data_ptr(), b, n, n_samples, _stream_ptr(xyz),
                        work=(float(b) * n * n_samples, 'point-updates'),
                        flop=float(n_samples))     # not flop: the DEPENDENT selection steps of the launch (latency roofline, bench.py)
    return out


def k_nearest_neighbor(input_xyz: torch.Tensor, query_xyz: torch.Tensor, k: int, cpp_impl=True):
    """k nearest inputs per query, ascending: -> int64 [B,n_queries,k] (wrapper.py:106-127).

    Accepts [B,N,D] or [B,D,N]; the layout is sniffed exactly like the reference (shape[1] <= 3
    means channel-first, wrapper.py:119-122).
    """
    if input_xyz.shape[1] <= 3:  # channel_first to channel_last
        assert query_xyz.shape[1] == input_xyz.shape[1]
        input_xyz = input_xyz.transpose(1, 2).contiguous()
        query_xyz = query_xyz.transpose(1, 2).contiguous()

    if not cpp_impl:
        dists = squared_distance(query_xyz, input_xyz)
        return dists.topk(k, dim=2, largest=False).indices.to(torch.long)

    _require_cuda('k_nearest_neighbor', input_xyz, query_xyz)
    lib = _lib.load()
    input_xyz = input_xyz.contiguous().float()
    query_xyz = query_xyz.contiguous().float()
    b, m, d = input_xyz.shape
    nq = query_xyz.shape[1]
    assert query_xyz.shape[0] == b and query_xyz.shape[2] == d
    out = torch.empty((b, nq, k), dtype=torch.int64, device=query_xyz.device)
    with _on_device(input_xyz):
        _lib.launch('camli_knn', lib.camli_knn, input_xyz.data_ptr(), query_xyz.data_ptr(), out.data_ptr(),
                    b, m, nq, d, k, _stream_ptr(input_xyz), work=(float(b) * m * nq, 'pairs'))
    return out


def k_nearest_neighbor_prefixes(input_xyz: torch.Tensor, query_xyz: torch.Tensor, sizes, k: int, prior=None):
    """[k_nearest_neighbor(input_xyz[:, :m], query_xyz, k) for m in sizes] in one scan (sizes strictly descending,
    sizes[0] = all inputs): the levels of the FPS pyramid are nested prefixes (models/utils.py:121-125), and the
    sequential insertion semantics of the search make the k-list after the first m candidates the answer for that
    prefix.  Channel-last [B,M,D] / [B,Nq,D] only (internal helper of the cores, not part of the reference boundary).
    ``prior``: the list this function returned for the same clouds a GRU iteration earlier (camli_knn_prefixes_prior: a bound
    on every k-th distance, same results, a fraction of the sorted insertions)."""
    import ctypes
    _require_cuda('k_nearest_neighbor_prefixes', input_xyz, query_xyz)
    lib = _lib.load()
    input_xyz = input_xyz.contiguous().float()
    query_xyz = query_xyz.contiguous().float()
    b, m, d = input_xyz.shape
    nq = query_xyz.shape[1]
    sizes = [int(v) for v in sizes]
    assert sizes[0] == m and all(a > c for a, c in zip(sizes[:-1], sizes[1:])) and len(sizes) <= 4
    outs = [torch.empty((b, nq, k), dtype=torch.int64, device=query_xyz.device) for _ in sizes]
    ptrs = (ctypes.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    csizes = (ctypes.c_int * len(sizes))(*sizes)
    if prior is not None:
        assert len(prior) == len(sizes) and all(p.shape == (b, nq, k) and p.dtype == torch.int64 and p.is_contiguous() and
                                                p.device == query_xyz.device for p in prior)
        pptrs = (ctypes.c_void_p * len(prior))(*[p.data_ptr() for p in prior])
        with _on_device(input_xyz):
            _lib.launch('camli_knn', lib.camli_knn_prefixes_prior, input_xyz.data_ptr(), query_xyz.data_ptr(), ptrs, pptrs, csizes,
                        len(sizes), b, m, nq, d, k, _stream_ptr(input_xyz), work=(float(b) * m * nq, 'pairs'))
        return outs
    with _on_device(input_xyz):
        _lib.launch('camli_knn', lib.camli_knn_prefixes, input_xyz.data_ptr(), query_xyz.data_ptr(), ptrs, csizes, len(sizes),
                    b, m, nq, d, k, _stream_ptr(input_xyz), work=(float(b) * m * nq, 'pairs'))
    return outs

"""Which implementation the composite operators of the cores use.

``'hip'``       (default) every composite op with a fused gfx950 kernel calls it through the C-ABI;
                tensors must live on the GPU and the library must be present -- otherwise the call
                raises.  This is the product path.
``'composed'``  the same math composed from torch primitives exactly as the reference writes it
                (models/utils.py, models/point_conv.py, models/raft_core.py ...).  Device-agnostic;
                it is the fp32 torch reference the fused kernels are tested against, and what the
                CPU tests / the cpu_baseline leg of bench.py select explicitly.

There is no automatic switching: a CPU tensor under 'hip' is an error, not a fallback.
"""
import contextlib

_BACKEND = 'hip'


def backend():
    return _BACKEND


def set_backend(name):
    global _BACKEND
    if name not in ('hip', 'composed'):
        raise ValueError("backend must be 'hip' or 'composed', got %r" % (name,))
    _BACKEND = name


@contextlib.contextmanager
def use_backend(name):
    prev = backend()
    set_backend(name)
    try:
        yield
    finally:
        set_backend(prev)


def fused():
    return _BACKEND == 'hip'


# ------------------------------------------------------------------------------------------------
# deferred parameter gradients: the 12 GRU iterations share their parameters, so autograd adds a
# fresh weight / bias gradient into .grad after every call (~1,200 small add + sum launches per
# step).  With this switch on, the fused 1x1-convolution and bias/activation nodes accumulate into
# one buffer per parameter inside their own kernels (GEMM with beta = 1, float atomics) and a
# callback at the end of backward() moves the totals into .grad.  Opt-in because it bypasses the
# functional API: torch.autograd.grad(...) would not see these gradients (training loops call
# .backward()).
# ------------------------------------------------------------------------------------------------
_DEFER_PARAM_GRADS = False


def set_deferred_param_grads(enabled):
    global _DEFER_PARAM_GRADS
    _DEFER_PARAM_GRADS = bool(enabled)


def deferred_param_grads():
    return _DEFER_PARAM_GRADS


class _ParamGradSink:
    """Per-parameter accumulators of one backward pass (see set_deferred_param_grads)."""

    def __init__(self):
        self.entries = {}
        self.armed = False

    def slot(self, param, make, reduce_batch):
        import torch
        entry = self.entries.get(id(param))
        if entry is None:
            if not self.armed:
                torch.autograd.variable.Variable._execution_engine.queue_callback(self.flush)
                self.armed = True
            entry = self.entries[id(param)] = [param, make(), reduce_batch, None]
        entry[3] = torch.cuda.current_stream(param.device)
        return entry[1]

    def flush(self):
        import torch
        entries, self.entries, self.armed = self.entries, {}, False
        with torch.no_grad():
            for param, acc, reduce_batch, stream in entries.values():
                current = torch.cuda.current_stream(param.device)
                if stream is not None and stream != current:
                    current.wait_stream(stream)          # the accumulator was last written on another lane
                    acc.record_stream(current)
                grad = (acc.sum(0) if reduce_batch else acc).view_as(param)
                param.grad = grad if param.grad is None else param.grad + grad


PARAM_GRADS = _ParamGradSink()


# ------------------------------------------------------------------------------------------------
# two-lane execution: point branch on a side HIP stream next to the image branch
# ------------------------------------------------------------------------------------------------
_OVERLAP = False
_side_streams = {}


def set_overlap(enabled):
    """Run the 3-D (point) branch of the fused model on a second HIP stream.  The point kernels are
    small (B*2048 points) and leave most CUs idle; the image branch's convolutions do not depend
    on them between fusion points, so the two lanes overlap.  Results are unchanged."""
    global _OVERLAP
    _OVERLAP = bool(enabled)


def overlap():
    return _OVERLAP


def _flatten(items):
    for it in items:
        if isinstance(it, (list, tuple)):
            yield from _flatten(it)
        elif it is not None:
            yield it


class Lanes:
    """Fork/join helper.  ``side()`` is a context that issues work on the side stream;
    ``to_side(...)`` / ``to_main(...)`` order the two streams at a hand-over and tell the caching
    allocator that the handed-over tensors are in use on the other stream."""

    def __init__(self, device):
        import torch
        self.enabled = _OVERLAP and _BACKEND == 'hip' and device.type == 'cuda'
        if self.enabled:
            self._torch = torch
            self.main = torch.cuda.current_stream(device)
            key = (device.index, self.main.cuda_stream)
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device)
            self.side_stream = _side_streams[key]

    def side(self):
        if not self.enabled:
            return contextlib.nullcontext()
        return self._torch.cuda.stream(self.side_stream)

    def to_side(self, *tensors):
        if self.enabled:
            self.side_stream.wait_stream(self.main)
            for t in _flatten(tensors):
                t.record_stream(self.side_stream)

    def to_main(self, *tensors):
        if self.enabled:
            self.main.wait_stream(self.side_stream)
            for t in _flatten(tensors):
                t.record_stream(self.main)

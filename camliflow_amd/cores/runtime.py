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

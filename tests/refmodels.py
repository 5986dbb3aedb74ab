"""Load the REFERENCE's model code with this repo's operator semantics substituted for
``models.csrc`` (the "load unchanged" arrangement of SURVEY 8b).  Build container only."""
import sys

import refshim


def install(native_semantics=True):
    """After this, ``import models.camliraft_core`` etc. resolve to /root/reference with
    ``models.csrc.{k_nearest_neighbor,furthest_point_sampling,correlation2d}`` replaced by the
    oracle-backed operators (= the native kernels' index semantics) when native_semantics=True."""
    refshim.install()
    import models.csrc as ref_csrc  # noqa: F401  (prints the reference's "failed to load CUDA ext" notice)
    if native_semantics:
        from oracle import torch_ops
        for name in ('k_nearest_neighbor', 'furthest_point_sampling', 'correlation2d'):
            setattr(ref_csrc, name, getattr(torch_ops, name))
    for mod in [m for m in sys.modules if m.startswith('models.') and m not in ('models.csrc', 'models.csrc.wrapper')]:
        del sys.modules[mod]  # re-bind `from .csrc import ...` in modules imported earlier

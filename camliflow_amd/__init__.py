"""camliflow_amd -- MI355X (gfx950) native hot path of CamLiFlow / CamLiRAFT.

Layout
  csrc/    hand-written HIP kernels, the C-ABI shared library (include/camli_hip.h) and the
           Python operator boundary that mirrors the reference's ``models/csrc`` package
  cores/   host-side mirror of the reference's model cores (orchestration of the operators)

The HIP library is mandatory: importing the operators works everywhere (so that the package can be
built and inspected on a machine without a GPU), but calling one without ``libcamli_hip.so`` or with
non-CUDA tensors raises -- there is no CPU fallback in this package.
"""

import os as _os

# hipBLASLt's stream-K GEMM kernels (the `SK3` Tensile solutions it picks for most fp32 shapes on gfx950) make their
# workgroups wait for each other's partial tiles through flags in ONE synchroniser buffer per library handle, and torch
# uses one handle per host thread for ALL streams.  Two such GEMMs running at the same time on two streams then wait on
# flags the other one resets: the GPU spins at 100 % busy for good (round 3: reproduced 6 of 6 with the CLFM's two
# directions on two streams, cured 2 of 2 by this switch; in hindsight also the "two-lane start-up dead-lock" that round 2
# saw in 5 of 9 runs).  Data-parallel mode keeps the kernels but gives every workgroup whole tiles -- no cross-workgroup
# waits -- and costs nothing measurable on this workload (248.3 vs 249.1 ms per step).  Must be in the environment before
# the first GEMM creates the handle; an explicit user setting wins.
import sys as _sys


def _cuda_live():
    """torch was imported AND has created its CUDA context (a GEMM, hence a BLAS handle, may already exist)."""
    torch = _sys.modules.get('torch')
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:      # noqa: BLE001 -- a torch without CUDA support
        return False


# what set_overlap(True) consults (cores/runtime.streamk_safe): was the switch already in the environment, or did this
# import put it there BEFORE the process touched the GPU?  If torch had a live CUDA context and the variable was unset,
# a handle in stream-K mode may exist and the variable lands too late.
STREAMK_PRESET = 'TENSILE_STREAMK_DATA_PARALLEL' in _os.environ
STREAMK_SET_BEFORE_CUDA = STREAMK_PRESET or not _cuda_live()
_os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')

__version__ = "0.1.0"

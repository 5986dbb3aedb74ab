"""camliflow_amd -- MI355X (gfx950) native hot path of CamLiFlow / CamLiRAFT.

Layout
  csrc/    hand-written HIP kernels, the C-ABI shared library (include/camli_hip.h) and the
           Python operator boundary that mirrors the reference's ``models/csrc`` package
  cores/   host-side mirror of the reference's model cores (orchestration of the operators)

The HIP library is mandatory: importing the operators works everywhere (so that the package can be
built and inspected on a machine without a GPU), but calling one without ``libcamli_hip.so`` or with
non-CUDA tensors raises -- there is no CPU fallback in this package.
"""

__version__ = "0.1.0"

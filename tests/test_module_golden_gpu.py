"""The modules of SURVEY 8c on the HIP path against the goldens recorded from the reference's own modules
(tests/golden/make_module_golden.py): outputs, input gradients, parameter-gradient norms.  Strict mode: no op with a
fused kernel may drop to the composed formulation."""
import pytest

from test_module_golden import MODULE_RUNS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(MODULE_RUNS))
def test_hip_module_matches_reference_golden(name, golden):
    from camliflow_amd.cores import runtime
    with runtime.use_backend('hip'):
        runtime.set_strict(True)
        try:
            MODULE_RUNS[name](golden(name), 'cuda')
        finally:
            runtime.set_strict(False)

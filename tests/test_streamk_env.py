"""Multi-stream safety switch (camliflow_amd/__init__.py): importing the package puts hipBLASLt's stream-K kernels into
data-parallel mode unless the user chose otherwise -- two stream-K GEMMs on two streams of one handle dead-lock."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_value):
    env = {k: v for k, v in os.environ.items() if k != 'TENSILE_STREAMK_DATA_PARALLEL'}
    if env_value is not None:
        env['TENSILE_STREAMK_DATA_PARALLEL'] = env_value
    out = subprocess.run([sys.executable, '-c', 'import os, camliflow_amd; print(os.environ["TENSILE_STREAMK_DATA_PARALLEL"])'],
                         cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    return out.stdout.strip()


def test_package_import_sets_data_parallel_streamk():
    assert _run(None) == '1'


def test_user_setting_wins():
    assert _run('0') == '0'


def _probe(code, env_value, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('TENSILE_STREAMK_DATA_PARALLEL', 'CAMLI_OVERLAP_FORCE')}
    if env_value is not None:
        env['TENSILE_STREAMK_DATA_PARALLEL'] = env_value
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, '-W', 'always', '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    return out.stdout.strip(), out.stderr


_OVERLAP_PROBE = ('import camliflow_amd\n'
                  'from camliflow_amd.cores import runtime\n'
                  'runtime.set_overlap(True)\n'
                  'print(runtime.streamk_safe(), runtime.overlap())')


def test_overlap_is_granted_when_the_switch_was_in_place_before_cuda():
    out, err = _probe(_OVERLAP_PROBE, None)
    assert out == 'True True' and 'refused' not in err


def test_overlap_is_refused_when_the_user_switched_data_parallel_off():
    """set_overlap(True) must not enable concurrent GEMM streams when stream-K cannot be guaranteed off."""
    out, err = _probe(_OVERLAP_PROBE, '0')
    assert out == 'False False' and 'multi-stream execution refused' in err
    out, err = _probe(_OVERLAP_PROBE, '0', {'CAMLI_OVERLAP_FORCE': '1'})
    assert out == 'False True'


def test_overlap_is_refused_when_cuda_was_live_before_the_package_import():
    """A host program that ran GPU work first: the variable set by the import lands after the BLAS handle may exist.
    Simulated on a CPU-only box by making torch.cuda.is_initialized() report a live context before the import."""
    code = ('import torch\n'
            'torch.cuda.is_initialized = lambda: True\n' + _OVERLAP_PROBE)
    out, err = _probe(code, None)
    assert out == 'False False' and 'multi-stream execution refused' in err
    out, err = _probe(code, '1')          # preset by the launcher: fine whatever ran before
    assert out == 'True True'


def test_tuned_gemm_table_ships_and_is_not_installed_without_a_gpu():
    """runtime.use_tuned_gemms: the table (TunableOp results of the bench configurations on the MI355X image) is in the package,
    carries the validators TunableOp checks, and nothing is touched on a machine without a GPU or when the caller opted out."""
    import os
    import torch
    from camliflow_amd.cores import runtime
    assert os.path.exists(runtime.GEMM_TUNING_FILE)
    lines = open(runtime.GEMM_TUNING_FILE).read().splitlines()
    validators = [ln for ln in lines if ln.startswith('Validator,')]
    assert {ln.split(',')[1] for ln in validators} >= {'PT_VERSION', 'HIPBLASLT_VERSION', 'ROCBLAS_VERSION', 'GCN_ARCH_NAME'}
    assert any('gfx950' in ln for ln in validators)
    rows = [ln for ln in lines if ln and not ln.startswith('Validator,')]
    assert len(rows) > 50 and all(len(ln.split(',')) >= 3 for ln in rows)
    if not torch.cuda.is_available():
        assert runtime.use_tuned_gemms() is False
    os.environ['CAMLI_TUNED_GEMMS'] = '0'
    try:
        assert runtime.use_tuned_gemms() is False
    finally:
        os.environ.pop('CAMLI_TUNED_GEMMS')

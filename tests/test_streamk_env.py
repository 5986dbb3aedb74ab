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

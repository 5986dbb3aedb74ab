import os
import sys

import pytest

os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')     # before torch creates a BLAS handle (camliflow_amd/__init__.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'needs_reference: imports /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    import refshim
    if refshim.reference_available():
        return
    skip = pytest.mark.skip(reason='/root/reference not present on this machine')
    for item in items:
        if 'needs_reference' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope='session')
def oracle_dense():
    """oracle/dense.py: numpy restatement of the dense round-3 kernels' reference code (pinned by tests/test_dense_oracle.py)."""
    from oracle import dense
    return dense

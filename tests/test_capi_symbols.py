"""The C-ABI library builds, loads and exports exactly what include/camli_hip.h declares (CPU; no
compute calls), and the operator boundary fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'camli_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(camli_\w+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from camliflow_amd.csrc import build, _lib
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert 'camli_knn' in names and 'camli_fps' in names and 'camli_corr2d_fwd' in names
    for name in names:
        assert hasattr(lib, name), 'missing export: ' + name
    # and the python binding table covers the header, no more no less
    assert sorted(_lib.PROTOTYPES) == names
    assert _lib.load().camli_version() >= 100


def test_header_cites_reference_for_every_entry_point():
    text = open(os.path.join(ROOT, 'include', 'camli_hip.h')).read()
    for needle in ['k_nearest_neighbor.cpp', 'furthest_point_sampling.cpp', 'correlation.cpp', 'wrapper.py']:
        assert needle in text


def test_boundary_has_reference_signatures():
    import inspect
    from camliflow_amd import csrc
    assert list(inspect.signature(csrc.correlation2d).parameters) == ['input1', 'input2', 'max_displacement', 'cpp_impl']
    assert list(inspect.signature(csrc.furthest_point_sampling).parameters) == ['xyz', 'n_samples', 'cpp_impl']
    assert list(inspect.signature(csrc.k_nearest_neighbor).parameters) == ['input_xyz', 'query_xyz', 'k', 'cpp_impl']
    assert list(inspect.signature(csrc.squared_distance).parameters) == ['xyz1', 'xyz2']


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_ops_refuse_cpu_tensors():
    from camliflow_amd import csrc
    from camliflow_amd.csrc._lib import CamliHipError
    x = torch.rand(1, 100, 3)
    with pytest.raises(CamliHipError):
        csrc.k_nearest_neighbor(x, x, 3)
    with pytest.raises(CamliHipError):
        csrc.furthest_point_sampling(x, 10)
    with pytest.raises(CamliHipError):
        csrc.correlation2d(torch.rand(1, 4, 5, 5), torch.rand(1, 4, 5, 5), 1)
    with pytest.raises(AssertionError):
        csrc.furthest_point_sampling(x, 100)  # wrapper.py:98: n_points must exceed n_samples


def test_composed_formulation_runs_anywhere():
    from camliflow_amd import csrc
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 64, 3, generator=g)
    idx = csrc.k_nearest_neighbor(x, x, 4, cpp_impl=False)
    assert idx.shape == (2, 64, 4) and torch.equal(idx[:, :, 0], torch.arange(64)[None].expand(2, 64))
    assert csrc.furthest_point_sampling(x, 8, cpp_impl=False).shape == (2, 8)
    out = csrc.correlation2d(torch.rand(1, 4, 5, 6, generator=g), torch.rand(1, 4, 5, 6, generator=g), 2, cpp_impl=False)
    assert out.shape == (1, 25, 5, 6)


def test_python_boundary_keeps_every_operator_of_the_reference_wrapper():
    """models/csrc/wrapper.py:18-127 exports correlation2d, furthest_point_sampling, k_nearest_neighbor and (utils) squared_distance;
    the cores also rely on the nested-prefix search.  A lost definition must fail here, on the CPU."""
    from camliflow_amd.csrc import wrapper
    for name in ('correlation2d', 'furthest_point_sampling', 'k_nearest_neighbor', 'k_nearest_neighbor_prefixes', 'squared_distance',
                 'CorrelationFunction'):
        assert callable(getattr(wrapper, name, None)), name
    import camliflow_amd.csrc as boundary
    for name in ('correlation2d', 'furthest_point_sampling', 'k_nearest_neighbor'):
        assert callable(getattr(boundary, name, None)), name

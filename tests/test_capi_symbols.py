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


def test_splitk_workspace_size_is_host_arithmetic():
    """camli_allpairs_build_bwd_workspace_bytes launches nothing: level l of the pyramid adjoint is cut into min(2^l, 8) ranges of
    K steps of at least 16 steps (32 source pixels each) and every range of a split level holds a [B,C,P_l] partial."""
    from camliflow_amd.csrc import _lib
    lib = _lib.load()

    def want(p_levels, b, c, p):
        steps = (p + 31) // 32
        total = 0
        for lvl, pl in enumerate(p_levels):
            parts = 8 if lvl >= 3 else 1 << lvl
            while parts > 1 and steps // parts < 16:
                parts //= 2
            if parts > 1:
                total += parts * b * c * pl
        return 4 * total

    for p_levels, b, c, p in (((8160, 2040, 510, 120), 8, 256, 8160), ((2160, 540, 135, 28), 1, 256, 2160), ((480, 120, 30, 6), 2, 64, 480),
                              ((100,), 3, 8, 100)):
        arr = (ctypes.c_int * len(p_levels))(*p_levels)
        assert int(lib.camli_allpairs_build_bwd_workspace_bytes(arr, len(p_levels), b, c, p)) == want(p_levels, b, c, p)
    assert int(lib.camli_allpairs_build_bwd_workspace_bytes(None, 4, 8, 256, 8160)) == 0


def test_cat_conv_channels_last_node_on_cpu():
    """blocks._CatConvCL without a GPU (torch copies instead of camli_transpose_planes): conv2d(cat(parts)) and its three
    gradients, for the GRU's two filter shapes and a part list of one."""
    from camliflow_amd.cores.blocks import cat_conv_cl
    g = torch.Generator().manual_seed(0)
    for ksize, padding, widths in (((1, 5), (0, 2), (8, 6)), ((5, 1), (2, 0), (8, 6)), ((3, 3), (1, 1), (7,))):
        parts = [torch.randn(2, c, 9, 11, generator=g, requires_grad=True) for c in widths]
        w = torch.randn(5, sum(widths), *ksize, generator=g, requires_grad=True)
        gy = torch.randn(2, 5, 9, 11, generator=g)
        want = torch.nn.functional.conv2d(torch.cat(parts, 1), w, None, padding=padding)
        want_g = torch.autograd.grad(want, parts + [w], gy)
        got = cat_conv_cl(parts, w, padding)
        got_g = torch.autograd.grad(got, parts + [w], gy)
        assert got.is_contiguous() and torch.allclose(got, want, atol=1e-5)
        for a, b in zip(got_g, want_g):
            assert a.shape == b.shape and torch.allclose(a, b, atol=1e-4)

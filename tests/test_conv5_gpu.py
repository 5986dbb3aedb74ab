"""GRU2D's separable convolutions as implicit GEMMs on the fp32 matrix cores (csrc/hip/conv5.hip, camli_conv5_fwd, round 4):
the plain form and the two GRU epilogues against oracle/dense.py + oracle/glue.py (numpy, pinned on the reference's GRU2D:
tests/test_dense_oracle.py), the inference path of this repo's GRU2D against the reference module's recorded output
(tests/golden/dense_gru2d.npz, models/raft_core.py:110-140), and against its own training-path formulation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, want, tol=2e-5, what=''):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape and np.abs(got - want).max() <= tol * scale, (what, float(np.abs(got - want).max()), scale)


CASES = [  # (B, C0, C1, Cout, H, W): tile edges (W > 128, W % 4 != 0, H % 4 != 0), channel counts off the 8 / 128 grids
    (2, 24, 8, 40, 9, 21), (1, 128, 128, 256, 17, 30), (2, 16, 24, 32, 11, 13), (1, 5, 3, 7, 3, 5), (1, 64, 70, 130, 6, 150),
    (1, 8, 0, 16, 1, 1), (2, 128, 128, 128, 20, 36),
]


@pytest.mark.parametrize('case', CASES, ids=str)
@pytest.mark.parametrize('vertical', [False, True])
def test_conv5_plain_vs_oracle(case, vertical, oracle_dense):
    from camliflow_amd.csrc import fused
    b, c0, c1, cout, h, w = case
    rng = np.random.default_rng(sum(case) + vertical)
    x0, x1 = rng.standard_normal((b, c0, h, w), dtype=np.float32), rng.standard_normal((b, c1, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, c0 + c1) + ((5, 1) if vertical else (1, 5))) * (5 * (c0 + c1)) ** -0.5).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    with torch.no_grad():
        got = fused.conv5(dev(x0), dev(x1), fused.pack_conv5_weight(dev(wt)), vertical, dev(bias))
    _close(got, oracle_dense.conv5_fwd(np.concatenate([x0, x1], axis=1), wt, bias), what='conv5')


@pytest.mark.parametrize('case', [(2, 16, 24, 11, 13), (1, 128, 128, 17, 30), (2, 8, 8, 5, 140)], ids=str)
@pytest.mark.parametrize('vertical', [False, True])
def test_conv5_gru_epilogues_vs_oracle(case, vertical, oracle_dense):
    """gates: z, r*h = split(sigmoid(conv(cat[h, x]) + ctx)); blend: h' = (1 - z) h + z tanh(conv(cat[r*h, x]) + ctx)."""
    from camliflow_amd.csrc import fused
    from oracle import glue
    b, hd, md, h, w = case
    rng = np.random.default_rng(sum(case) + vertical)
    shape = (5, 1) if vertical else (1, 5)
    hh = np.tanh(rng.standard_normal((b, hd, h, w), dtype=np.float32))
    x = rng.standard_normal((b, md, h, w), dtype=np.float32)
    w_zr = (rng.standard_normal((2 * hd, hd + md) + shape) * (5 * (hd + md)) ** -0.5).astype(np.float32)
    w_q = (rng.standard_normal((hd, hd + md) + shape) * (5 * (hd + md)) ** -0.5).astype(np.float32)
    ctx_zr = rng.standard_normal((b, 2 * hd, h, w), dtype=np.float32)
    ctx_q = rng.standard_normal((b, hd, h, w), dtype=np.float32)
    with torch.no_grad():
        z, rh = fused.conv5_gru_gates(dev(hh), dev(x), fused.pack_conv5_weight(dev(w_zr)), dev(ctx_zr), vertical)
    want_z, want_rh, _ = glue.gru_gates_fwd(oracle_dense.conv5_fwd(np.concatenate([hh, x], 1), w_zr), ctx_zr, hh)
    _close(z, want_z, what='z')
    _close(rh, want_rh, what='r*h')
    for nan_to_num in (False, True):
        with torch.no_grad():
            hn = fused.conv5_gru_blend(dev(want_rh), dev(x), fused.pack_conv5_weight(dev(w_q)), dev(ctx_q), dev(want_z), dev(hh),
                                       vertical, nan_to_num=nan_to_num)
        want, _ = glue.gru_blend_fwd(oracle_dense.conv5_fwd(np.concatenate([want_rh, x], 1), w_q), ctx_q, want_z, hh, nan_to_num)
        _close(hn, want, what='h')


def _gru_from_golden(g):
    from camliflow_amd.cores.raft2d import GRU2D
    hd, cd = int(g['hidden']), int(g['context'])
    gru = GRU2D(hidden_dim=hd, input_dim=g['x'].shape[1]).cuda()
    with torch.no_grad():
        for name in ('convz1', 'convr1', 'convq1', 'convz2', 'convr2', 'convq2'):
            getattr(gru, name).weight.copy_(dev(g[name + '_w']))
            getattr(gru, name).bias.copy_(dev(g[name + '_b']))
    return gru, hd, cd


def test_gru2d_inference_path_vs_reference_module_golden(golden, monkeypatch):
    """prepare() + step() of this repo's GRU2D under no_grad on the 'hip' backend = the conv5 kernels; expected: the
    REFERENCE GRU2D's recorded new hidden state."""
    from camliflow_amd.cores import runtime
    monkeypatch.setenv('CAMLI_CONV5', '1')          # opt-in: the step is faster on the library convolutions (raft2d.GRU2D.prepare)
    g = golden('dense_gru2d')
    gru, hd, cd = _gru_from_golden(g)
    h0, x = dev(g['h0']), dev(g['x'])
    with torch.no_grad(), runtime.use_backend('hip'):
        runtime.set_census(True)
        runtime.reset_census()
        state = gru.prepare(x[:, :cd].contiguous())
        out = gru.step(h0, x[:, cd:].contiguous(), state)
        census = runtime.census()
        runtime.set_census(False)
    assert census['fused'].get('camli_conv5_fwd', 0) == 4, census['fused']
    _close(out, g['out'], what='h1')


def test_gru2d_inference_path_equals_training_path_formulation(monkeypatch):
    """Same weights, same inputs at the step's real channel counts: conv5 kernels (no_grad) vs library convolutions + gate
    kernels (grad enabled), 1/8-resolution size with a ragged edge."""
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.raft2d import GRU2D
    monkeypatch.setenv('CAMLI_CONV5', '1')
    torch.manual_seed(0)
    gru = GRU2D(hidden_dim=128, input_dim=128 + 128).cuda()
    ctx = torch.randn(2, 128, 34, 60, device='cuda')
    h0 = torch.tanh(torch.randn(2, 128, 34, 60, device='cuda'))
    motion = torch.randn(2, 128, 34, 60, device='cuda')
    with runtime.use_backend('hip'):
        with torch.no_grad():
            fast = gru.step(h0, motion, gru.prepare(ctx))
        slow = gru.step(h0, motion, gru.prepare(ctx)).detach()
    assert (fast - slow).abs().max().item() <= 2e-5

"""Module-level goldens recorded from the REFERENCE's own modules (tests/golden/make_module_golden.py):
(1) the oracle functions that no op-level fixture reaches are pinned on tensors captured INSIDE those modules;
(2) this repo's host-side mirror of every module reproduces the reference's output and gradients -- here on the CPU
    with the oracle-backed operators (composed formulation); tests/test_module_golden_gpu.py repeats (2) on the HIP path.
"""
import numpy as np
import pytest
import torch

from modelutils import oracle_boundary


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.requires_grad_(True) if grad else t


def load_params(mod, g):
    state = {k[2:].replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith('p_')}
    mod.load_state_dict(state, strict=True)
    return mod


def check_grad_norms(mod, g, rtol=2e-3):
    got = {n: p.grad.double().norm().item() for n, p in mod.named_parameters() if p.grad is not None}
    for name, want in zip(g['gn_names'], g['gn_values']):
        assert abs(got[str(name)] - want) <= rtol * want + 1e-6, (name, got[str(name)], want)


def close(got, want, rtol=1e-4):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    return np.abs(got - want).max() <= rtol * np.abs(want).max() + 1e-6


def gclose(got, want, rtol=1e-3):
    """gradients: relative L2 error (a max / ReLU routing decision that flips on a last-ulp difference moves single
    elements by O(1) -- the same convention as the model-level gradient checks)"""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    return np.linalg.norm((got - want).ravel()) <= rtol * np.linalg.norm(want.ravel()) + 1e-6


# ---- (1) oracle pins -------------------------------------------------------------------------------------------
def test_oracle_pointconv_mix_pinned_on_the_reference_linear_input(golden, oracle_lib):
    g = golden('module_pointconv')
    feat_cl = np.concatenate([g['xyz'], g['feat']], axis=1).transpose(0, 2, 1)
    mixed = oracle_lib.pointconv_mix_fwd(feat_cl, g['wgt'], g['knn'], 16)
    assert close(mixed.reshape(g['mixed'].shape), g['mixed'], 1e-5)


def test_oracle_corr3d_gather_pinned_on_the_reference_cost_mlp_input(golden, oracle_lib):
    g = golden('module_corr3d_raft')
    cost = np.einsum('bcn,bcm->bnm', g['f1'], g['f2']).astype(np.float32) / np.float32(g['f1'].shape[1])
    got = oracle_lib.corr3d_gather_fwd(g['xyz1'], g['xyz2'], cost, g['cross0'])
    assert np.array_equal(got[:, :3], g['lookup0'][:, :3])           # offsets: exact
    assert close(got[:, 3], g['lookup0'][:, 3], 1e-5)                # cost entries: up to the bmm's summation order


def test_oracle_pointconv_dw_bwd_pinned_on_the_reference_gradients(golden, oracle_lib):
    g = golden('module_pointconv_dw_bwd')
    k = int(g['k'])
    out, arg = oracle_lib.pointconv_dw_fwd(g['mlp_out'], g['weight'], g['knn'], k)
    assert close(out, g['out'], 1e-6)
    gfeat, gweight = oracle_lib.pointconv_dw_bwd(g['grad_out'], g['mlp_out'], g['weight'], g['knn'], arg, k)
    assert close(gfeat, g['gmlp_out'], 1e-5) and close(gweight, g['gweight'], 1e-5)


def test_oracle_knn_interp_adjoints_pinned_on_the_reference_autograd(golden, oracle_lib):
    g = golden('knn_interpolation_grad')
    assert close(oracle_lib.knn_interp_fwd(g['in_xyz'], g['feat'], g['q_xyz'], g['knn']), g['out'], 1e-5)
    assert close(oracle_lib.knn_interp_bwd(g['in_xyz'], g['grad_out'], g['q_xyz'], g['knn'], g['feat'].shape[2]), g['gfeat'], 1e-5)
    g_in, g_q = oracle_lib.knn_interp_bwd_xyz(g['in_xyz'], g['feat'], g['grad_out'], g['q_xyz'], g['knn'])
    assert close(g_in, g['g_in_xyz'], 2e-4) and close(g_q, g['g_q_xyz'], 2e-4)


def test_oracle_convex_upsample_pinned_on_the_reference(golden, oracle_lib):
    g = golden('convex_upsample')
    assert close(oracle_lib.convex_upsample_fwd(g['flow'], g['mask'], 8), g['out8'], 1e-5)
    assert close(oracle_lib.convex_upsample_fwd(g['flow'], g['mask4'], 4), g['out4'], 1e-5)


def test_oracle_input_side_pinned_on_the_reference_preprocessing(golden, oracle_lib):
    g = golden('input_side')
    i1, i2 = oracle_lib.pad_normalize(g['images'], [int(v) for v in g['pad']], [123.675, 116.280, 103.530], [58.395, 57.120, 57.375])
    assert np.array_equal(i1, g['image1']) and np.array_equal(i2, g['image2'])
    p1, p2 = oracle_lib.persp2paral(g['pcs'], g['intrinsics'], g['persp_hw'], g['paral_hw'])
    assert np.allclose(p1, g['pc1'], rtol=1e-6, atol=1e-6) and np.allclose(p2, g['pc2'], rtol=1e-6, atol=1e-6)


def test_oracle_projection_pinned_on_the_reference(golden, oracle_lib):
    """oracle_project_pc2image == the reference's project_pc2image (both cameras) + the in-place grid rescale"""
    g = golden('project_pc2image')
    gh, gw = [int(v) for v in g['grid_hw']]
    for name, persp, (sh, sw) in (('persp', 1, g['persp_hw']), ('paral', 0, g['paral_hw'])):
        cx, cy = [float(v) for v in g['paral_c']]
        plain = oracle_lib.project_pc2image(g['pc_' + name], g['intrinsics'], persp, cx, cy, 1.0, 1.0)
        assert np.array_equal(plain, g['uv_' + name])
        grid = oracle_lib.project_pc2image(g['pc_' + name], g['intrinsics'], persp, cx, cy, (gw - 1) / (int(sw) - 1), (gh - 1) / (int(sh) - 1))
        assert np.array_equal(grid, g['uv_grid_' + name])


def test_oracle_pwc3d_pieces_reproduce_the_reference_cost_volume(golden, oracle_lib):
    """pair / ksum / gather_wsum (oracle) + the module's own small MLPs (numpy) == the reference Correlation3D output"""
    g = golden('module_corr3d_pwc')
    p = {k[2:].replace('__', '.'): g[k] for k in g.files if k.startswith('p_')}
    xyz1, xyz2, f1, f2, own = g['xyz1'], g['xyz2'], g['f1'], g['f2'], g['own']
    b, c, n = f1.shape
    cross = oracle_lib.knn(xyz2.transpose(0, 2, 1), xyz1.transpose(0, 2, 1), 16)

    def gather(x, idx):
        return np.stack([x[i][:, idx[i]] for i in range(x.shape[0])])

    def mlp(x, prefix, acts):
        for i, act in enumerate(acts):
            w = p['%s.convs.%d.conv_fn.weight' % (prefix, i)][:, :, 0, 0]
            x = np.einsum('oc,bcnk->bonk', w, x) + p['%s.convs.%d.conv_fn.bias' % (prefix, i)][None, :, None, None]
            x = np.maximum(x, 0) if act == 'relu' else np.where(x > 0, x, np.float32(0.1) * x)
        return x.astype(np.float32)
    d_cross = gather(xyz2, cross) - xyz1[..., None]
    w0 = p['cost_mlp.convs.0.conv_fn.weight'][:, :, 0, 0]
    a = np.einsum('oc,bcn->bon', w0[:, :c], f1).astype(np.float32)
    bm = np.einsum('oc,bcn->bon', w0[:, c:2 * c], f2).astype(np.float32)
    e = (np.einsum('oc,bcnk->bonk', w0[:, 2 * c:], d_cross) + p['cost_mlp.convs.0.conv_fn.bias'][None, :, None, None]).astype(np.float32)
    h1 = oracle_lib.pwc3d_pair_fwd(a, bm, e, cross, 0.1)
    w1 = p['cost_mlp.convs.1.conv_fn.weight'][:, :, 0, 0]
    h2 = np.einsum('oc,bcnk->bonk', w1, h1) + p['cost_mlp.convs.1.conv_fn.bias'][None, :, None, None]
    h2 = np.where(h2 > 0, h2, np.float32(0.1) * h2).astype(np.float32)
    to_patch = oracle_lib.ksum_fwd(mlp(d_cross, 'weight_net2', ['relu'] * 3), h2)
    d_own = gather(xyz1, own) - xyz1[..., None]
    patch = oracle_lib.gather_wsum_fwd(mlp(d_own, 'weight_net1', ['relu'] * 3), to_patch, own)
    wa = p['feat_aligner.conv_fn.weight'][:, :, 0]
    out = np.einsum('oc,bcn->bon', wa, patch) + p['feat_aligner.conv_fn.bias'][None, :, None]
    out = np.where(out > 0, out, np.float32(0.1) * out)
    assert close(out, g['out'], 1e-4)


# ---- (2) host-side mirror vs the reference modules -------------------------------------------------------------------
def run_pointconv(g, device):
    from camliflow_amd.cores.setconv import PointConv
    mod = load_params(PointConv(13, 24, norm=None, k=16), g).to(device).eval()
    feat = _t(g['feat']).to(device).requires_grad_(True)
    out = mod(_t(g['xyz']).to(device), feat, _t(g['sampled']).to(device))
    out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(feat.grad, g['gfeat'])
    check_grad_norms(mod, g)


def run_pointconv_dw(g, device):
    from camliflow_amd.cores.setconv import PointConvDW, pass_cache
    mod = load_params(PointConvDW(20, 32, k=int(g['k'])), g).to(device)
    feat = _t(g['feat']).to(device).requires_grad_(True)
    with pass_cache():
        out = mod(_t(g['xyz']).to(device), feat, knn_indices=_t(g['knn']).to(device))
        out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(feat.grad, g['gfeat'])
    check_grad_norms(mod, g)


def run_corr3d_raft(g, device):
    from camliflow_amd.cores.raft3d import Correlation3D
    from camliflow_amd.cores.setconv import pass_cache
    mod = load_params(Correlation3D(out_channels=128, k=16), g).to(device)
    xyz1, xyz2 = _t(g['xyz1']).to(device), _t(g['xyz2']).to(device)
    xyzs2 = [xyz2[:, :, :m].contiguous() for m in (256, 128, 64, 32)]
    f1, f2 = _t(g['f1']).to(device).requires_grad_(True), _t(g['f2']).to(device).requires_grad_(True)
    with pass_cache():
        mod.build_cost_volume_pyramid(f1, f2, xyzs2)
        assert close(mod.cost_volume_pyramid[3], g['level3'], 1e-5)
        out = mod(xyz1, xyzs2)
        out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(f1.grad, g['gf1']) and gclose(f2.grad, g['gf2'])
    check_grad_norms(mod, g)


def run_corr3d_pwc(g, device):
    from camliflow_amd.cores.pwc3d import Correlation3D
    mod = load_params(Correlation3D(32, 32, 64), g).to(device)
    xyz2 = _t(g['xyz2']).to(device).requires_grad_(True)
    f1, f2 = _t(g['f1']).to(device).requires_grad_(True), _t(g['f2']).to(device).requires_grad_(True)
    out = mod(_t(g['xyz1']).to(device), f1, xyz2, f2, _t(g['own']).to(device))
    out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(f1.grad, g['gf1']) and gclose(f2.grad, g['gf2'])
    assert gclose(xyz2.grad, g['gxyz2'], 2e-3)
    check_grad_norms(mod, g)


def run_clfm(g, device):
    from camliflow_amd.cores.fusion import CLFM
    from camliflow_amd.cores.setconv import pass_cache
    mod = load_params(CLFM(32, 32, fusion_fn='sk', norm=None), g).to(device)
    f2d, f3d = _t(g['f2d']).to(device).requires_grad_(True), _t(g['f3d']).to(device).requires_grad_(True)
    with pass_cache():
        o2d, o3d = mod(_t(g['uv']).to(device), f2d, f3d)
        torch.autograd.backward([o2d, o3d], [_t(g['g2d']).to(device), _t(g['g3d']).to(device)])
    assert close(o2d, g['out2d']) and close(o3d, g['out3d'])
    assert gclose(f2d.grad, g['gf2d']) and gclose(f3d.grad, g['gf3d'])
    check_grad_norms(mod, g)


def run_gru3d(g, device):
    from camliflow_amd.cores.raft3d import GRU3D
    from camliflow_amd.cores.setconv import pass_cache
    mod = load_params(GRU3D(input_dim=64, hidden_dim=32), g).to(device)
    h, x = _t(g['h']).to(device).requires_grad_(True), _t(g['x']).to(device).requires_grad_(True)
    with pass_cache():
        out = mod(_t(g['xyz']).to(device), h, x, _t(g['knn']).to(device))
        out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(h.grad, g['gh']) and gclose(x.grad, g['gx'])
    check_grad_norms(mod, g)


def run_motion3d(g, device):
    from camliflow_amd.cores.raft3d import MotionEncoder3D
    from camliflow_amd.cores.setconv import pass_cache
    mod = load_params(MotionEncoder3D(corr_dim=128), g).to(device)
    flow, corr = _t(g['flow']).to(device).requires_grad_(True), _t(g['corr']).to(device).requires_grad_(True)
    with pass_cache():
        out = mod(_t(g['xyz']).to(device), flow, corr, _t(g['knn']).to(device))
        out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(flow.grad, g['gflow']) and gclose(corr.grad, g['gcorr'])
    check_grad_norms(mod, g)


def run_flowhead3d(g, device):
    from camliflow_amd.cores.raft3d import FlowHead3D
    from camliflow_amd.cores.setconv import pass_cache
    mod = load_params(FlowHead3D(input_dim=128), g).to(device)
    feat = _t(g['feat']).to(device).requires_grad_(True)
    with pass_cache():
        out = mod(_t(g['xyz']).to(device), feat, _t(g['knn']).to(device))
        out.backward(_t(g['grad_out']).to(device))
    assert close(out, g['out']) and gclose(feat.grad, g['gfeat'])
    check_grad_norms(mod, g)


MODULE_RUNS = {'module_pointconv': run_pointconv, 'module_pointconv_dw_bwd': run_pointconv_dw,
               'module_corr3d_raft': run_corr3d_raft, 'module_corr3d_pwc': run_corr3d_pwc, 'module_clfm': run_clfm,
               'module_gru3d': run_gru3d, 'module_motion3d': run_motion3d, 'module_flowhead3d': run_flowhead3d}


@pytest.mark.parametrize('name', sorted(MODULE_RUNS))
def test_mirror_module_matches_reference_golden_on_cpu(name, golden):
    with oracle_boundary():
        MODULE_RUNS[name](golden(name), 'cpu')

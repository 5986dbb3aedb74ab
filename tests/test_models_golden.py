"""Host-side model mirror (camliflow_amd/cores) against model-level golden vectors recorded from the
REFERENCE's own model code (tests/golden/make_model_golden.py).  CPU: the boundary operators are the
oracle-backed ones, composite ops run in their torch-composed form.  All six model families."""
import numpy as np
import pytest
import torch

from modelutils import MODEL_CASES, grad_fingerprint, hashed_fill_, oracle_boundary, synthetic_inputs


@pytest.mark.parametrize('name', sorted(MODEL_CASES))
def test_model_matches_reference_golden(name, golden):
    import camliflow_amd.cores as cores
    g = golden('model_' + name)
    _, cls, cfg_fn, shape = MODEL_CASES[name]
    torch.manual_seed(0)
    model = hashed_fill_(getattr(cores, cls)(cfg_fn()), scale=0.5)
    inputs = synthetic_inputs(*shape)
    checksum = np.array([float(v.double().sum()) for v in inputs.values()])
    assert np.allclose(checksum, g['input_checksum'], rtol=1e-9), 'synthetic input generator drifted'
    with oracle_boundary():
        for mode in ('eval', 'train'):
            getattr(model, mode)()
            model.zero_grad()
            res = model(inputs)
            loss = model.get_loss()
            for k, v in res.items():
                want = g['%s_%s' % (mode, k)]
                # EPE between this mirror and the reference's recorded flow (north star: <= 1e-4)
                epe = np.linalg.norm(v.detach().numpy() - want, axis=1).mean()
                assert epe <= 1e-4, (mode, k, epe)
            assert abs(loss.item() - float(g['%s_loss' % mode])) <= 1e-4 * max(1.0, abs(float(g['%s_loss' % mode])))
            if mode == 'train':
                loss.backward()
                names, norms = grad_fingerprint(model)
                assert names == list(g['grad_names'])
                assert np.allclose(norms, g['grad_norms'], rtol=2e-3, atol=1e-6)


@pytest.mark.needs_reference
def test_reference_cores_load_unchanged_over_this_boundary():
    """SURVEY 8b 'load unchanged': the reference's models/camliraft_core.py and camlipwc_core.py import
    and run with THIS repo's operator package substituted for models.csrc (here: its CPU stand-in with
    the kernels' index semantics), and this repo's mirror reproduces them exactly."""
    import refmodels
    refmodels.install(native_semantics=True)
    from models.camliraft_core import CamLiRAFT_Core as RefCore
    import models.utils as ref_utils
    from oracle import torch_ops
    assert ref_utils.k_nearest_neighbor is torch_ops.k_nearest_neighbor     # the substitution took effect
    from camliflow_amd.cores import CamLiRAFT_Core
    from camliflow_amd.cores.camliraft import _camera_pair
    from camliflow_amd.cores.geometry import persp2paral
    from modelutils import camliraft_cfg
    torch.manual_seed(0)
    ref = hashed_fill_(RefCore(camliraft_cfg(2)), scale=0.5).eval()
    mine = CamLiRAFT_Core(camliraft_cfg(2)).eval()
    mine.load_state_dict(ref.state_dict(), strict=True)       # identical parameter names
    inp = synthetic_inputs(1, 128, 160, 4608, with_targets=False)
    persp, paral = _camera_pair(128, 160, inp['intrinsics'])
    pc1 = persp2paral(inp['pcs'][:, :3], persp, paral)
    pc2 = persp2paral(inp['pcs'][:, 3:], persp, paral)
    img1, img2 = inp['images'][:, :3] / 255.0, inp['images'][:, 3:] / 255.0
    with torch.no_grad():
        f2d_ref, f3d_ref = ref(img1, img2, pc1, pc2, paral)
        with oracle_boundary():
            f2d, f3d = mine(img1, img2, pc1, pc2, paral)
    for a, b in zip(f2d_ref + f3d_ref, f2d + f3d):
        assert torch.equal(a, b)

"""Shared helpers for model-level tests and fixture generation (TEST INFRASTRUCTURE)."""
import contextlib
import zlib
from types import SimpleNamespace as NS

import torch


def camliraft_cfg(n_iters=2, **over):
    cfg = dict(name='camliraft', batch_size=1, freeze_bn=False, backbone=NS(depth=50, pretrained=None),
               n_iters_train=n_iters, n_iters_eval=n_iters, fuse_fnet=True, fuse_cnet=True, fuse_corr=True,
               fuse_motion=True, fuse_hidden=False, loss2d=NS(gamma=0.8, order='l2-norm'),
               loss3d=NS(gamma=0.8, order='l2-norm'))
    cfg.update(over)
    return NS(**cfg)


def camliraft_l_cfg(n_iters=2):
    return NS(name='camliraft_l', batch_size=1, n_iters_train=n_iters, n_iters_eval=n_iters,
              ids=NS(enabled=True), loss=NS(gamma=0.8, order='l2-norm'))


def hashed_fill_(module, scale=1.0):
    """Deterministic, name-keyed parameter fill: seed = crc32(name); weights ~ N(0, 1/fan_in);
    BatchNorm running stats are filled too.  No weights are ever shipped."""
    with torch.no_grad():
        persistent = set(module.state_dict().keys())      # constants kept as non-persistent buffers stay as they are
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if name not in persistent or t.dtype not in (torch.float32, torch.float64):
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith('running_var'):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith('running_mean'):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() <= 1:
                if 'norm' in name.split('.')[-2] or name.split('.')[-2].startswith('bn') or name.split('.')[-2] in ('1',):
                    base = 1.0 if name.endswith('weight') else 0.0
                    t.copy_(base + torch.randn(t.shape, generator=g) * 0.1)
                else:
                    t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * fan_in ** -0.5 * scale)
    return module


def synthetic_inputs(b, h, w, n_points, seed=0, f=1050.0, with_targets=True, zmax=35.0):
    """FlyingThings3D-shaped synthetic sample: projections land inside the image."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randint(0, 256, (b, 6, h, w), generator=g).float()
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    z = torch.rand(b, n_points, generator=g) * (zmax - 5.0) + 5.0
    u = torch.rand(b, n_points, generator=g) * (w - 1)
    v = torch.rand(b, n_points, generator=g) * (h - 1)
    pc1 = torch.stack([(u - cx) * z / f, (v - cy) * z / f, z], dim=1)
    pc2 = pc1 + torch.randn(b, 3, n_points, generator=g) * 0.05
    inputs = {'images': images, 'pcs': torch.cat([pc1, pc2], dim=1),
              'intrinsics': torch.tensor([[f, cx, cy]]).repeat(b, 1)}
    if with_targets:
        inputs['flow_2d'] = torch.cat([torch.randn(b, 2, h, w, generator=g), torch.ones(b, 1, h, w)], dim=1)
        inputs['flow_3d'] = torch.randn(b, 3, n_points, generator=g) * 0.05
    return inputs


@contextlib.contextmanager
def oracle_boundary():
    """Patch the four boundary operators of camliflow_amd with the oracle-backed CPU versions and
    select the torch-composed composite ops.  CPU tests / cpu_baseline only."""
    from camliflow_amd.csrc import wrapper
    from camliflow_amd.cores import runtime
    from oracle import torch_ops
    saved = {n: getattr(wrapper, n) for n in ('k_nearest_neighbor', 'furthest_point_sampling', 'correlation2d')}
    for n in saved:
        setattr(wrapper, n, getattr(torch_ops, n))
    try:
        with runtime.use_backend('composed'):
            yield
    finally:
        for n, fn in saved.items():
            setattr(wrapper, n, fn)


def camlipwc_cfg():
    norm2d = NS(feature_pyramid='batch_norm', flow_estimator=None, context_network=None)
    norm3d = NS(feature_pyramid='batch_norm', correlation=None, flow_estimator=None)
    weights = NS(level_weights=[8, 4, 2, 1, 0.5], order='l2-norm')
    return NS(name='camlipwc', batch_size=1, freeze_bn=False,
              pwc2d=NS(norm=norm2d, max_displacement=4, lite_estimator=False, fixed=False),
              pwc3d=NS(norm=norm3d, fixed=False, k=16),
              fusion=NS(fuse_pyramid=True, fuse_correlation=True, fuse_estimator=True),
              loss2d=weights, loss3d=weights)


def camlipwc_l_cfg():
    return NS(name='camlipwc_l', batch_size=1, ids=NS(enabled=True),
              norm=NS(feature_pyramid='batch_norm', correlation=None, flow_estimator=None),
              loss=NS(level_weights=[8, 4, 2, 1, 0.5], order='l2-norm'))


def pwc_cfg():
    return NS(name='pwc', batch_size=1, max_displacement=4, lite_estimator=False,
              norm=NS(feature_pyramid='batch_norm', flow_estimator=None, context_network=None),
              loss=NS(level_weights=[8, 4, 2, 1, 0.5], order='l2-norm'))


def raft_cfg():
    return NS(name='raft', batch_size=1, backbone=NS(depth=50, pretrained=None), n_iters_train=2, n_iters_eval=2,
              loss=NS(gamma=0.8, order='l2-norm'))


# fixture name -> (reference module, class name, cfg factory, synthetic_inputs args (b, h, w, n_points))
MODEL_CASES = {
    'camliraft': ('camliraft', 'CamLiRAFT', lambda: camliraft_cfg(2), (1, 128, 160, 4608)),
    'camliraft_l': ('camliraft_l', 'CamLiRAFT_L', lambda: camliraft_l_cfg(2), (1, 128, 160, 4608)),
    'camlipwc': ('camlipwc', 'CamLiPWC', camlipwc_cfg, (1, 128, 192, 4608)),
    'camlipwc_l': ('camlipwc_l', 'CamLiPWC_L', camlipwc_l_cfg, (1, 128, 192, 4608)),
    'pwc': ('pwc', 'PWC', pwc_cfg, (1, 128, 192, 4608)),
    'raft': ('raft', 'RAFT', raft_cfg, (1, 128, 160, 4608)),
    # round 4: every golden above is batch 1 -- a batch-stride slip in an orchestration path would pass them.
    # Two distinct samples per batch through the fused 2D+3D models of both families.
    'camliraft_b2': ('camliraft', 'CamLiRAFT', lambda: camliraft_cfg(2), (2, 128, 160, 4608)),
    'camlipwc_b2': ('camlipwc', 'CamLiPWC', camlipwc_cfg, (2, 128, 192, 4608)),
}


def grad_fingerprint(model, every=7):
    """(names, L2 norms) of every `every`-th parameter gradient -- a compact backward-pass check."""
    import numpy as np
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None][::every]
    return [n for n, _ in named], np.array([p.grad.double().norm().item() for _, p in named])


def share_clouds(monkeypatch, pc1=None, pc2=None, rel_tol=1e-4):
    """Make ``build_pc_pyramid`` of every core module receive given post-IDS clouds on the GPU instead of the ones the
    GPU transform produced: FPS is only reproducible on bit-identical inputs, the IDS transform's log / divide differ
    in the last ulp between CPU and GPU.  The clouds are either passed in (reference-recorded) or RECORDED from the
    first CPU call that passes through (the CPU port's run).  The GPU's own clouds must agree with them to rounding."""
    import importlib
    state = {'pc1': pc1, 'pc2': pc2}
    for name in ('camlipwc', 'camliraft', 'raft3d', 'pwc3d'):
        mod = importlib.import_module('camliflow_amd.cores.' + name)
        if not hasattr(mod, 'build_pc_pyramid'):
            continue
        original = mod.build_pc_pyramid

        def shared(_pc1, _pc2, *args, _orig=original, **kwargs):
            if not _pc1.is_cuda:          # the CPU port's own call: untouched, remembered
                if state['pc1'] is None:
                    state['pc1'], state['pc2'] = _pc1.detach().clone(), _pc2.detach().clone()
                return _orig(_pc1, _pc2, *args, **kwargs)
            assert state['pc1'] is not None and _pc1.shape == state['pc1'].shape
            ref1, ref2 = state['pc1'].cuda(), state['pc2'].cuda()
            assert (_pc1 - ref1).abs().max().item() <= rel_tol * max(1.0, ref1.abs().max().item())
            return _orig(ref1, ref2, *args, **kwargs)
        monkeypatch.setattr(mod, 'build_pc_pyramid', shared)
    return state

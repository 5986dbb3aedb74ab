"""CamLiRAFT end to end on the GPU.

(1) product path ('hip' backend: HIP kernels + fused autograd) vs the torch-composed formulation
    on the same device: flows within 1e-4 EPE, loss and parameter gradients agree.
(2) GPU core vs the CPU core driven by the oracle operators, on IDENTICAL core inputs.  The
    inverse-depth-scaling transform (log / divide) is evaluated once on the CPU and shared, because
    CPU and GPU transcendental functions differ in the last ulp and furthest-point sampling -- a
    chain of 4096 arg-max decisions -- is only reproducible on bit-identical inputs (the reference
    has the same property between its CPU and CUDA paths; tools/debug_fps_inputs.py shows it).
    Criterion (north star): |EPE2D_gpu - EPE2D_cpu| and |EPE3D_gpu - EPE3D_cpu| <= 1e-4, fp32.
"""
import pytest
import torch

from modelutils import camliraft_cfg, hashed_fill_, oracle_boundary, synthetic_inputs

pytestmark = pytest.mark.gpu


def _to(inputs, device):
    return {k: v.to(device) for k, v in inputs.items()}


def _epe(a, b):
    return torch.linalg.norm(a - b, dim=1).mean().item()


def _models(n_iters, mode):
    from camliflow_amd.cores import CamLiRAFT
    torch.manual_seed(0)
    cfg = camliraft_cfg(n_iters=n_iters)
    cpu_model = hashed_fill_(CamLiRAFT(cfg))
    gpu_model = CamLiRAFT(cfg)
    gpu_model.load_state_dict(cpu_model.state_dict())
    gpu_model.cuda()
    getattr(cpu_model, mode)()
    getattr(gpu_model, mode)()
    return cpu_model, gpu_model


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_hip_backend_vs_composed_backend(mode):
    from camliflow_amd.cores import runtime
    _, model = _models(3, mode)
    inputs = _to(synthetic_inputs(2, 128, 160, 4608), 'cuda')
    res = {}
    for backend in ('hip', 'composed'):
        with runtime.use_backend(backend):
            out = model(inputs)
            loss = model.get_loss()
            grads = None
            if mode == 'train':
                model.zero_grad()
                loss.backward()
                grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        res[backend] = (out, loss.item(), grads)
    (oh, lh, gh), (oc, lc, gc) = res['hip'], res['composed']
    for key in ('flow_2d', 'flow_3d'):
        assert _epe(oh[key], oc[key]) <= 1e-4, key
    assert abs(lh - lc) <= 1e-4 * max(1.0, abs(lc))
    if mode == 'train':
        assert gh.keys() == gc.keys() and len(gh) > 400
        # composed-path backward uses atomics (index_put / grid_sampler): compare in norm, not elementwise
        num = sum(((gh[n] - gc[n]).double() ** 2).sum().item() for n in gh) ** 0.5
        den = sum((gc[n].double() ** 2).sum().item() for n in gh) ** 0.5
        assert num / den < 1e-3, num / den
        worst = max(((gh[n] - gc[n]).norm() / (gc[n].norm() + 1e-3 * den / len(gh))).item() for n in gh)
        assert worst < 2e-2, worst


def _core_inputs(inputs):
    """CamLiRAFT.forward's preprocessing, evaluated on the CPU (camliraft.py:32-64 of the reference)."""
    from camliflow_amd.cores.camliraft import _camera_pair, _IMAGENET_MEAN, _IMAGENET_STD
    from camliflow_amd.cores.geometry import InputPadder, persp2paral
    images = inputs['images'].float()
    padder = InputPadder(images.shape, x=8)
    image1, image2 = padder.pad(images[:, :3], images[:, 3:])
    mean = torch.tensor(_IMAGENET_MEAN).reshape(1, 3, 1, 1)
    std = torch.tensor(_IMAGENET_STD).reshape(1, 3, 1, 1)
    persp, paral = _camera_pair(image1.shape[-2], image1.shape[-1], inputs['intrinsics'])
    pc1 = persp2paral(inputs['pcs'][:, :3], persp, paral)
    pc2 = persp2paral(inputs['pcs'][:, 3:], persp, paral)
    return (image1 - mean) / std, (image2 - mean) / std, pc1, pc2, paral


def test_gpu_core_vs_cpu_oracle_core_epe_parity():
    from camliflow_amd.cores import runtime
    cpu_model, gpu_model = _models(4, 'eval')
    inputs = synthetic_inputs(1, 128, 160, 4608)
    image1, image2, pc1, pc2, paral = _core_inputs(inputs)
    with torch.no_grad():
        with oracle_boundary():
            f2d_cpu, f3d_cpu = cpu_model.core(image1, image2, pc1, pc2, paral)
        with runtime.use_backend('hip'):
            f2d_gpu, f3d_gpu = gpu_model.core(image1.cuda(), image2.cuda(), pc1.cuda(), pc2.cuda(), paral)
    tgt2d, tgt3d = inputs['flow_2d'][:, :2], inputs['flow_3d']
    for it in range(len(f2d_cpu)):
        epe2d_cpu, epe2d_gpu = _epe(f2d_cpu[it], tgt2d), _epe(f2d_gpu[it].cpu(), tgt2d)
        epe3d_cpu, epe3d_gpu = _epe(f3d_cpu[it], tgt3d), _epe(f3d_gpu[it].cpu(), tgt3d)
        assert abs(epe2d_cpu - epe2d_gpu) <= 1e-4, (it, epe2d_cpu, epe2d_gpu)
        assert abs(epe3d_cpu - epe3d_gpu) <= 1e-4, (it, epe3d_cpu, epe3d_gpu)
    print('final-iteration flow difference: 2d %.2e px, 3d %.2e' %
          (_epe(f2d_cpu[-1], f2d_gpu[-1].cpu()), _epe(f3d_cpu[-1], f3d_gpu[-1].cpu())))


def test_two_lane_stream_overlap_matches_single_stream():
    """runtime.set_overlap(True): point branch on a side HIP stream.  Same kernels, same inputs ->
    flows agree to float-atomic noise and the gradients in norm; repeated to shake out races."""
    from camliflow_amd.cores import runtime
    _, model = _models(3, 'train')
    inputs = _to(synthetic_inputs(2, 128, 160, 4608), 'cuda')

    def run():
        model.zero_grad()
        out = model(inputs)
        loss = model.get_loss()
        loss.backward()
        torch.cuda.synchronize()
        return out, loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    with runtime.use_backend('hip'):
        base_out, base_loss, base_grads = run()
        runtime.set_overlap(True)
        try:
            for rep in range(4):
                out, loss, grads = run()
                epes = {key: _epe(out[key], base_out[key]) for key in ('flow_2d', 'flow_3d')}
                num = sum(((grads[n] - base_grads[n]).double() ** 2).sum().item() for n in grads) ** 0.5
                den = sum((base_grads[n].double() ** 2).sum().item() for n in grads) ** 0.5
                print('overlap rep %d: epe %s loss %.8f vs %.8f grad rel %.3e' % (rep, epes, loss, base_loss, num / den))
                # run-to-run noise floor of the forward itself is ~1e-6 EPE / 1e-7 relative loss (library
                # GEMM / convolution kernels with split accumulation, and which algorithm the library picks per call: a
                # sibling bound at 1x the floor flaked once in round 6); a stream race would be far above it
                for key, epe in epes.items():
                    assert epe <= 5e-5, (key, epe)
                assert abs(loss - base_loss) <= 5e-5 * max(1.0, abs(base_loss))
                # the backward accumulates with float atomics whose order differs between the schedules;
                # observed 4e-7 .. 1e-4 through 3 recurrent iterations
                assert num / den < 1e-3, num / den
        finally:
            runtime.set_overlap(False)


@pytest.mark.parametrize('name', ['camlipwc', 'camlipwc_l', 'pwc', 'raft', 'camliraft_l'])
def test_other_model_families_hip_vs_composed(name):
    """Every model family of the reference (factory.py:21-35) through the product path vs the
    torch-composed formulation on the same device: final flows within 1e-4 EPE, same loss."""
    import camliflow_amd.cores as cores
    from camliflow_amd.cores import runtime
    from modelutils import MODEL_CASES
    _, cls, cfg_fn, shape = MODEL_CASES[name]
    torch.manual_seed(0)
    model = hashed_fill_(getattr(cores, cls)(cfg_fn()), scale=0.5).cuda().train()
    inputs = _to(synthetic_inputs(*shape), 'cuda')
    res = {}
    for backend in ('hip', 'composed'):
        with runtime.use_backend(backend):
            model.zero_grad()
            out = model(inputs)
            loss = model.get_loss()
            loss.backward()
            res[backend] = (out, loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    (oh, lh, gh), (oc, lc, gc) = res['hip'], res['composed']
    for key in oh:
        assert _epe(oh[key], oc[key]) <= 1e-4, key
    assert abs(lh - lc) <= 1e-4 * max(1.0, abs(lc))
    num = sum(((gh[n] - gc[n]).double() ** 2).sum().item() for n in gh) ** 0.5
    den = sum((gc[n].double() ** 2).sum().item() for n in gh) ** 0.5
    assert num / den < 1e-3, num / den


def test_camlipwc_config2_shape_runs():
    """BASELINE configs[1]: CamLiPWC 960x540 + 8192 points, fp32, batch 1 (forward + backward)."""
    import camliflow_amd.cores as cores
    from camliflow_amd.cores import runtime
    from modelutils import camlipwc_cfg
    torch.manual_seed(0)
    model = cores.CamLiPWC(camlipwc_cfg()).cuda().train()
    inputs = _to(synthetic_inputs(1, 540, 960, 8192), 'cuda')
    with runtime.use_backend('hip'):
        out = model(inputs)
        model.get_loss().backward()
    assert out['flow_2d'].shape == (1, 2, 540, 960) and out['flow_3d'].shape == (1, 3, 8192)
    assert torch.isfinite(out['flow_2d']).all() and torch.isfinite(out['flow_3d']).all()


@pytest.mark.parametrize('overlap', [False, True], ids=['one_lane', 'two_lanes'])
def test_deferred_parameter_gradients_match_autograd_accumulation(overlap):
    """runtime.set_deferred_param_grads(True): the 1x1-convolution weights and the fused biases accumulate
    inside their kernels and are moved into .grad by one callback at the end of backward().  Every
    parameter's gradient must equal the per-call autograd accumulation (same forward, same adjoint
    kernels, different summation order -> 1e-5 per tensor in norm; whole model 1e-3 like the overlap
    test, float atomics)."""
    from camliflow_amd.cores import runtime
    _, model = _models(3, 'train')
    inputs = _to(synthetic_inputs(2, 128, 160, 4608), 'cuda')

    def run():
        model.zero_grad()
        model(inputs)
        loss = model.get_loss()
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    runtime.set_overlap(overlap)
    try:
        with runtime.use_backend('hip'):
            base_loss, base = run()
            runtime.set_deferred_param_grads(True)
            try:
                for _ in range(2):
                    loss, grads = run()
                    assert not runtime.PARAM_GRADS.entries and not runtime.PARAM_GRADS.armed
                    assert grads.keys() == base.keys()
                    assert abs(loss - base_loss) <= 5e-5 * max(1.0, abs(base_loss))
                    num = sum(((grads[n] - base[n]).double() ** 2).sum().item() for n in grads) ** 0.5
                    den = sum((base[n].double() ** 2).sum().item() for n in grads) ** 0.5
                    assert num / den < 1e-3, num / den
                    worst = max(((grads[n] - base[n]).norm() / (base[n].norm() + 1e-12)).item() for n in grads
                                if base[n].norm() > 1e-6)
                    print('deferred grads: whole-model rel %.2e, worst tensor %.2e' % (num / den, worst))
            finally:
                runtime.set_deferred_param_grads(False)
    finally:
        runtime.set_overlap(False)

"""Headline benchmark: CamLiRAFT training step (forward + sequence losses + backward + clip + AdamW)
on synthetic FlyingThings3D-shaped inputs, 960x540 images + 8192 points (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W [--config camliraft|camlipwc|kitti|eval]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU, batch-dimension data parallel (SyncBatchNorm + one flat-bucket gradient all-reduce over
RCCL/xGMI per step); per-GPU batch is fixed, so the scaling is weak.  Rank 0 prints ONE JSON line.  `value` =
global frame-pairs per second with the inputs resident in HBM before the timed region.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks (one process per
GPU, the role of train.py:307's mp.spawn) and refuses cleanly when the box has fewer than N GPUs.

The ONE line on stdout is compact (< 4 KB, tests/test_bench_line.py): the driver contract + `config` + (N = 1, default
config only):
  roofline      the dominant north-star kernel (most device time in the timed region): in situ figure (HIP events on
                the launch stream over the timed region), `single_lane` (same kernel with the chip to itself), `traffic`
  cpu_baseline  the CPU port timed on the host cores: 1 warm-up + 2 timed steps at 8 threads (~25 s)
  parity        EPE2D / EPE3D of the HIP path against the CPU port (this repo's cores driven by the C oracle
                operators) on one sample of the SAME workload with shared post-IDS core inputs, plus bit-equality
                of the FPS / KNN indices; the run FAILS when |dEPE| > 1e-4 or an index differs
Everything bulky goes to `gpurun_out/bench_detail.json` (path in the line's `detail`): `hip_kernels` (every timed entry
point in situ), `roofline_rows` (one row per north-star kernel alone on the idle GPU, tools/kernel_bench.py), `census`
(fused launches vs composed fall-backs under the 'hip' backend, cores/runtime.py) and the long forms of the above.

Other configurations (own bench lines, not the headline; the batch-1 ones replay a HIP graph by default): --config camlipwc (configs[1]), kitti (configs[4],
bf16 autocast, 32 iterations), eval (batch 8, 20 iterations, inference).
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')     # see camliflow_amd/__init__.py: stream-K GEMMs on two streams

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from types import SimpleNamespace as NS  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# entry point -> roofline kind, for the kernels north_star names.  Glue kernels either side of the path
# (bias_act, gru_*, sk_*, ids_*, masked_l2, convex_upsample) are listed in `hip_kernels` but never selected.
NORTH_STAR = {
    'camli_corr2d_fwd': 'hbm', 'camli_corr2d_bwd': 'hbm',
    'camli_allpairs_build_fwd': 'mfma', 'camli_allpairs_build_bwd': 'mfma', 'camli_allpairs_fold_bwd': 'hbm',
    'camli_allpairs_lookup_fwd': 'hbm', 'camli_allpairs_lookup_bwd': 'hbm',
    'camli_knn': 'valu', 'camli_fps': 'fps',
    'camli_gather_cf_fwd': 'hbm', 'camli_gather_cf_bwd': 'hbm', 'camli_gather_cl_fwd': 'hbm', 'camli_gather_cl_bwd': 'hbm',
    'camli_pointconv_mix_fwd': 'hbm', 'camli_pointconv_mix_bwd': 'hbm',
    'camli_pointconv_dw_fwd': 'hbm', 'camli_pointconv_dw_bwd': 'hbm', 'camli_pointconv_dw_expand': 'hbm',
    'camli_weightnet_fwd': 'hbm', 'camli_weightnet_bwd': 'mfma',
    'camli_knn_interp_fwd': 'hbm', 'camli_knn_interp_bwd': 'hbm', 'camli_knn_interp_bwd_xyz': 'hbm',
    'camli_corr3d_gather_fwd': 'hbm', 'camli_corr3d_gather_bwd': 'hbm',
    'camli_corr3d_mlp_fwd': 'fma', 'camli_corr3d_mlp_bwd': 'fma',      # plain fp32 FMA on the vector ALU (registers only)
    'camli_convcl_gru_gates': 'mfma', 'camli_convcl_gru_blend': 'mfma', 'camli_convcl_fwd': 'mfma', 'camli_convcl_wrw': 'mfma',
    'camli_wino_conv3x3': 'mfma', 'camli_wino_wrw': 'mfma',       # flop = the transform-domain MFMA work (4/9 | 1/4 of the direct form's)
    'camli_wino1d_gru_gates': 'mfma', 'camli_wino1d_gru_blend': 'mfma', 'camli_wino1d_conv': 'mfma', 'camli_wino1d_wrw': 'mfma',       # 1-D F(4,5): 2/5 of the 5-tap form's
    'camli_pwc3d_pair_fwd': 'hbm', 'camli_pwc3d_pair_bwd': 'hbm', 'camli_gather_wsum_fwd': 'hbm', 'camli_gather_wsum_bwd': 'hbm',
}
# SURVEY 8(f)2 ("the 2-D convolution side"): the update block's convolutions on own matrix-core kernels.  Everything else in
# NORTH_STAR is a SURVEY 8(a) row (A1-A15), the path BASELINE.json's north_star names.
CONV_SIDE = {'camli_convcl_gru_gates', 'camli_convcl_gru_blend', 'camli_convcl_fwd', 'camli_convcl_wrw', 'camli_wino_conv3x3',
             'camli_wino_wrw', 'camli_wino1d_gru_gates', 'camli_wino1d_gru_blend', 'camli_wino1d_conv', 'camli_wino1d_wrw'}
MFMA_F32_PEAK_TFLOPS = 157.3
VALU_PAIR_PEAK_G = 7865.0
FPS_STEP_IDEAL_US = 0.35      # one dependent selection step with the cloud resident in registers (tools/kernel_bench.py)

CONFIGS = {
    # name: (model, height, width, points, iters, batch, mode, autocast dtype, BASELINE config it stands for)
    'camliraft': ('camliraft', 540, 960, 8192, 12, 8, 'train', None, 'configs[2]'),
    'camlipwc': ('camlipwc', 540, 960, 8192, 0, 1, 'train', None, 'configs[1]'),
    'kitti': ('camliraft', 375, 1242, 16384, 32, 1, 'train', torch.bfloat16, 'configs[4]'),
    'eval': ('camliraft', 540, 960, 8192, 20, 8, 'eval', None, 'SURVEY 8f rank 1 (eval_things.py: batch 8, n_iters_eval 20)'),
}


def model_cfg(n_iters):
    return NS(name='camliraft', batch_size=1, freeze_bn=False, backbone=NS(depth=50, pretrained=None),
              n_iters_train=n_iters, n_iters_eval=n_iters, fuse_fnet=True, fuse_cnet=True, fuse_corr=True,
              fuse_motion=True, fuse_hidden=False, loss2d=NS(gamma=0.8, order='l2-norm'),
              loss3d=NS(gamma=0.8, order='l2-norm'))


def build_model(args):
    from camliflow_amd.cores import CamLiPWC, CamLiRAFT
    if args.model == 'camlipwc':
        from modelutils import camlipwc_cfg
        return CamLiPWC(camlipwc_cfg())
    return CamLiRAFT(model_cfg(args.iters))


def synthetic_batch(b, h, w, n_points, seed, kitti=False):
    """uint8-valued images, points whose projections land inside the image, small flows."""
    g = torch.Generator().manual_seed(seed)
    f, cx, cy, zmax = (721.5, 609.6, 172.9, 90.0) if kitti else (1050.0, 479.5, 269.5, 35.0)
    images = torch.randint(0, 256, (b, 6, h, w), generator=g).float()
    z = torch.rand(b, n_points, generator=g) * (zmax - 5.0) + 5.0
    u = torch.rand(b, n_points, generator=g) * (w - 1)
    v = torch.rand(b, n_points, generator=g) * (h - 1)
    pc1 = torch.stack([(u - cx) * z / f, (v - cy) * z / f, z], dim=1)
    pc2 = pc1 + torch.randn(b, 3, n_points, generator=g) * 0.05
    return {'images': images, 'pcs': torch.cat([pc1, pc2], dim=1),
            'intrinsics': torch.tensor([[f, cx, cy]]).repeat(b, 1),
            'flow_2d': torch.cat([torch.randn(b, 2, h, w, generator=g), torch.ones(b, 1, h, w)], dim=1),
            'flow_3d': torch.randn(b, 3, n_points, generator=g) * 0.05}


class GraphedStep:
    """The whole training step (forward, losses, backward, clip, AdamW) captured once into a HIP graph
    and replayed.  Both HIP streams of the two-lane execution are captured (the side stream forks from and joins
    the capture stream).  Single GPU only (collectives stay outside graphs here)."""

    def __init__(self, step_fn, warmup=3):
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # the first of these steps is the one-lane priming pass of runtime.Lanes, and it runs on THIS side stream: the
            # parameters' AccumulateGrad nodes remember the stream of their first use, and a first use on the legacy
            # stream makes the later capture fail ("legacy stream would depend on a capturing stream")
            for _ in range(warmup + 1):
                step_fn()
                side.synchronize()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn()

    def __call__(self):
        self.graph.replay()
        return self.loss


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


_T_START = time.perf_counter()


def _log(msg):
    """Progress notes on stderr (never stdout: the one JSON line is the only thing there), stamped with the seconds
    since the process started so that a run that is cut off shows which leg it was in."""
    sys.stderr.write('bench.py [%6.1f s]: %s\n' % (time.perf_counter() - _T_START, msg))
    sys.stderr.flush()


def _elapsed():
    return time.perf_counter() - _T_START


def run_with_deadline(fn, seconds):
    """fn() on a worker thread, waited for at most `seconds`: (result, None) or (None, reason).  The CPU legs are single
    long torch calls that cannot be interrupted; a thread that overruns is abandoned (daemon) and main() leaves through
    os._exit after printing its line, so an overrun costs its leg, never the bench line."""
    import threading
    box = {}

    def work():
        try:
            box['result'] = fn()
        except BaseException as exc:      # noqa: BLE001 -- reported on the main thread
            box['error'] = exc

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(max(seconds, 1.0))
    if th.is_alive():
        return None, 'not finished %.0f s after it started (time budget, --time-budget)' % seconds
    if 'error' in box:
        raise box['error']
    return box['result'], None


# ------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` (the reference: train.py:307 mp.spawn, one process per GPU)
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def launch_ranks(n, argv, device_count=None, backend_env=None):
    """Start `n` ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), wait for them,
    pass rank 0's stdout through.  Refuses (exit code 2, nothing started) when the box has fewer than `n` GPUs: two
    ranks on one device is not a measurement.  A rank that fails takes the others down (exact PIDs, no patterns)."""
    if device_count is None:
        device_count = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if device_count < n:
        sys.stderr.write('bench.py: --gpus %d needs %d GPUs, this box has %d; refusing to oversubscribe\n'
                         % (n, n, device_count))
        return 2
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.update(backend_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    code = 0
    pending = list(procs)
    while pending:
        for proc in list(pending):
            rc = proc.poll()
            if rc is None:
                continue
            pending.remove(proc)
            if rc != 0 and code == 0:
                code = rc
                for other in pending:       # one rank failed: the collective of the others would never complete
                    other.terminate()
        time.sleep(0.2)
    return code


def make_optimizer(model, capturable=False):
    """AdamW with the reference's split learning rates (conf/training/flyingthings3d_subset/camliraft.yaml,
    factory.py:50-58: parameters under core.branch_3d get lr_3d)."""
    p3d = [p for n, p in model.named_parameters() if 'core.branch_3d' in n]
    p2d = [p for n, p in model.named_parameters() if 'core.branch_3d' not in n]
    return torch.optim.AdamW([{'params': p2d, 'lr': 2e-4}, {'params': p3d, 'lr': 2e-3}], weight_decay=1e-6,
                             capturable=capturable)


def allreduce_gradients(model, world, force=False):
    """Data-parallel gradient averaging as ONE flat bucket (33.5 MB for CamLiRAFT): a single RCCL
    all-reduce over xGMI after backward.  The payload is latency-, not bandwidth-bound, so
    there is nothing to gain from DDP's bucketed overlap -- and a plain collective issued after
    ``backward()`` (which joins every stream it used) stays correct when the point branch runs on
    its own HIP stream."""
    if world <= 1 and not force:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat)
    flat.div_(world)
    torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))     # one multi-tensor launch


def train_step(model, optimizer, batch, world=1, force_dist=False, autocast=None):
    if autocast is not None:
        with torch.autocast('cuda', dtype=autocast):
            model(batch)
            loss = model.get_loss()
    else:
        model(batch)
        loss = model.get_loss()
    loss.backward()
    allreduce_gradients(model, world, force_dist)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    model.clear_metrics()
    # the step's autograd graph is spent: let go of it (the model keeps `loss`, base.py:28-31) before the next forward pass
    # builds its own -- while it lives, the parameters' AccumulateGrad nodes of THIS step are reused by the next one with the
    # streams they were created on (torch warns about exactly that: "AccumulateGrad node's stream does not match ...")
    loss = loss.detach()
    model.loss = loss
    return loss


def eval_step(model, batch, autocast=None):
    with torch.no_grad():
        if autocast is not None:
            with torch.autocast('cuda', dtype=autocast):
                out = model({k: v for k, v in batch.items() if k not in ('flow_2d', 'flow_3d')})
        else:
            out = model({k: v for k, v in batch.items() if k not in ('flow_2d', 'flow_3d')})
    return out['flow_2d'].sum()


# ------------------------------------------------------------------------------------------------------------
# CPU port: timing (cpu_baseline) and the parity reference
# ------------------------------------------------------------------------------------------------------------
def _epe(pred, target):
    return torch.linalg.norm(pred - target, dim=1).mean().item()


def _cpu_model_name():
    try:
        out = subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if line.startswith('Model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline_and_reference(args, state_dict, deadline=None):
    """The CPU restatement path (this repo's cores driven by the C oracle operators) on the host cores: one
    sample (batch 1) of the same workload.  1 warm-up + 2 timed training steps at 8 threads -- the thread count of the
    in-container reference timing and the port's best: measured on the 128-core GPU box in round 2,
    8 threads 6.9 s, 32 threads 8.6 s, 128 threads 50 s per step (CAMLI_CPU_THREAD_SWEEP=1 repeats that sweep).  The
    warm-up step's forward is also the parity reference: it returns the final flows of that sample.  Reported, not
    the target.  `deadline` (seconds on the _elapsed() clock): a timed step is only started when one more step of the
    duration just seen fits before it; when not even one does, the warm-up step is the sample (and the line says so)."""
    from modelutils import oracle_boundary
    all_threads = torch.get_num_threads()
    model = build_model(args).train()
    model.load_state_dict(state_dict)
    opt = make_optimizer(model)
    batch = synthetic_batch(1, args.height, args.width, args.points, seed=1, kitti=args.config == 'kitti')
    times = {}
    ref = None
    if args.config == 'camliraft':
        plan = [('warmup', min(8, all_threads), 1), ('t8', min(8, all_threads), 2)]
        if all_threads > 8 and os.environ.get('CAMLI_CPU_THREAD_SWEEP') == '1':
            plan += [('t32', min(32, all_threads), 1), ('all', all_threads, 1)]
    else:       # the other configurations are side lines: parity reference + one timed step
        plan = [('warmup', min(8, all_threads), 1), ('t8', min(8, all_threads), 1)]
    with oracle_boundary():
        for label, threads, reps in plan:
            torch.set_num_threads(threads)
            for _ in range(reps):
                if deadline is not None and times and _elapsed() + 1.2 * max(t for runs in times.values() for t, _ in runs) > deadline:
                    _log('CPU port: no room for another %s step before the time budget ends' % label)
                    break
                t0 = time.perf_counter()
                if ref is None:       # the very first step: weights == state_dict, keep its forward as the reference
                    out = model(batch)
                    ref = {k: v.detach().clone() for k, v in out.items()}
                    loss = model.get_loss()
                    loss.backward()
                    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                    opt.step()
                    opt.zero_grad(set_to_none=True)
                    model.clear_metrics()
                else:
                    train_step(model, opt, batch)
                times.setdefault(label, []).append((time.perf_counter() - t0, threads))
                _log('CPU port: %s step at %d threads took %.1f s' % (label, threads, times[label][-1][0]))
    torch.set_num_threads(all_threads)
    timed = [(t, n) for label, runs in times.items() if label != 'warmup' for t, n in runs]
    cut = not timed
    if cut:        # the time budget left no room after the warm-up step: it is the sample
        timed = list(times['warmup'])
    best_t, best_n = min(timed)
    base = {'value': round(1.0 / best_t, 5), 'unit': 'frame-pairs/s', 'cores': best_n, 'kind': 'port',
            'sample': 'batch-1 training step (fwd+bwd+clip+AdamW) of the same workload, %dx%d + %d pts, %d iters: '
                      '1 warm-up (%.1f s) + %s; best taken'
                      % (args.width, args.height, args.points, args.iters, times['warmup'][0][0],
                         'no timed step (time budget): the warm-up step is the sample' if cut else
                         ', '.join('%.1f s @ %d thr' % (t, n) for t, n in timed)),
            'cpu_model': _cpu_model_name(), 'os_cpu_count': os.cpu_count()}
    return base, batch, ref


def cpu_all_cores_point(args, state_dict):
    """SURVEY 8d also asks for the CPU path at all physical cores: one more batch-1 training step of the port with torch's
    thread count = the physical core count (the 8-thread figure above is its best: round 2 measured 6.9 s at 8 threads,
    50 s at 128).  Runs LAST, under its own deadline."""
    from modelutils import oracle_boundary
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:      # noqa: BLE001
        cores = os.cpu_count()
    model = build_model(args).train()
    model.load_state_dict(state_dict)
    opt = make_optimizer(model)
    batch = synthetic_batch(1, args.height, args.width, args.points, seed=1, kitti=args.config == 'kitti')
    before = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        with oracle_boundary():
            t0 = time.perf_counter()
            train_step(model, opt, batch)
            dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    return {'value': round(1.0 / dt, 5), 'cores': cores, 'seconds_per_step': round(dt, 1), 'sample': 'one batch-1 training step, no warm-up'}


def parity_check(args, state_dict, batch, ref, device):
    """The HIP path on the sample the CPU port just ran, with SHARED post-IDS core inputs (the IDS transform uses
    log / divide, which differ in the last ulp between CPU and GPU, and FPS -- 4096 chained arg-max decisions -- is
    only reproducible on bit-identical inputs; tests/test_model_gpu.py does the same).  EPE as in eval_things.py:62,88."""
    import numpy as np
    import oracle
    from camliflow_amd import csrc
    from camliflow_amd.cores import runtime
    from camliflow_amd.cores.camliraft import _IMAGENET_MEAN, _IMAGENET_STD, _camera_pair
    from camliflow_amd.cores.geometry import InputPadder, flows_paral2persp, persp2paral
    model = build_model(args)
    model.load_state_dict(state_dict)
    model = model.to(device).train()
    images = batch['images'].float()
    padder = InputPadder(images.shape, x=8)
    image1, image2 = padder.pad(images[:, :3], images[:, 3:])
    mean = torch.tensor(_IMAGENET_MEAN).reshape(1, 3, 1, 1)
    std = torch.tensor(_IMAGENET_STD).reshape(1, 3, 1, 1)
    persp, paral = _camera_pair(image1.shape[-2], image1.shape[-1], batch['intrinsics'])
    pc1 = persp2paral(batch['pcs'][:, :3], persp, paral)          # on the CPU: identical to what the CPU port saw
    pc2 = persp2paral(batch['pcs'][:, 3:], persp, paral)
    with torch.no_grad(), runtime.use_backend('hip'):
        f2d, f3d = model.core(((image1 - mean) / std).to(device), ((image2 - mean) / std).to(device),
                              pc1.to(device), pc2.to(device), paral)
        flow_2d = padder.unpad(f2d[-1]).cpu()
        persp_cpu = {k: v for k, v in persp.items()}
        flow_3d = flows_paral2persp(pc1, [f3d[-1].cpu()], persp_cpu, paral)[0]
    tgt2d, tgt3d = batch['flow_2d'][:, :2], batch['flow_3d'][:, :3]
    epe2d_cpu, epe2d_gpu = _epe(ref['flow_2d'], tgt2d), _epe(flow_2d, tgt2d)
    epe3d_cpu, epe3d_gpu = _epe(ref['flow_3d'], tgt3d), _epe(flow_3d, tgt3d)
    # index parity on the sample's own (IDS-transformed) clouds
    both = torch.cat([pc1, pc2], dim=0).transpose(1, 2).contiguous()
    picks = csrc.furthest_point_sampling(both.to(device), 4096).cpu().numpy()
    fps_equal = bool(np.array_equal(picks, oracle.fps(both.numpy(), 4096)))
    lvl1 = torch.gather(both, 1, torch.from_numpy(picks)[:, :, None].expand(-1, -1, 3)).contiguous()
    lvl2 = lvl1[:, :2048].contiguous()
    knn_equal = True
    for inp, qry, k in ((both, lvl1, 16), (lvl1, lvl2, 16), (lvl2, lvl2, 32), (lvl2, both, 3)):
        got = csrc.k_nearest_neighbor(inp.to(device), qry.to(device), k).cpu().numpy()
        knn_equal = knn_equal and bool(np.array_equal(got, oracle.knn(inp.numpy(), qry.numpy(), k)))
    res = {'epe2d_cpu': round(epe2d_cpu, 6), 'epe2d_gpu': round(epe2d_gpu, 6), 'epe2d_abs_diff': abs(epe2d_cpu - epe2d_gpu),
           'epe3d_cpu': round(epe3d_cpu, 6), 'epe3d_gpu': round(epe3d_gpu, 6), 'epe3d_abs_diff': abs(epe3d_cpu - epe3d_gpu),
           'flow2d_mean_diff_px': _epe(flow_2d, ref['flow_2d']), 'flow3d_mean_diff': _epe(flow_3d, ref['flow_3d']),
           'fps_equal': fps_equal, 'knn_equal': knn_equal, 'tolerance': 1e-4,
           'sample': 'batch 1, %dx%d + %d pts, %d iters, train-mode forward, fp32, shared post-IDS core inputs; '
                     'reference = CPU port (cores + C oracle operators)' % (args.width, args.height, args.points, args.iters)}
    res['ok'] = bool(res['epe2d_abs_diff'] <= 1e-4 and res['epe3d_abs_diff'] <= 1e-4 and fps_equal and knn_equal)
    return res


SIDE_CONFIGS = ('camlipwc', 'kitti', 'ddp4')   # BASELINE configs[1], configs[4] and the per-rank cost of configs[3]
# ddp4 = what ONE rank of configs[3] (batch 32 over 8 GPUs, train.py:99-101,307) runs: batch 4, SyncBatchNorm, broadcast,
# the flat gradient all-reduce, on a 1-rank RCCL group (CAMLI_FORCE_DIST=1) -- the collectives cross no link here, so this is
# the per-rank compute + launch cost of that configuration, not its scaling


def side_configs(budget):
    """ms per step of the other two single-GPU configurations of BASELINE.json, each as its own process (its own model,
    graph replay, library warm-up) with a hard time limit taken from what is left of the time budget; a configuration
    that does not fit is reported as skipped.  < 300 bytes in the line."""
    out = {}
    for name in SIDE_CONFIGS:
        left = budget - _elapsed()
        if left < 75:
            out[name] = {'skipped': 'time budget'}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), '--config', name, '--steps', '5', '--warmup', '2',
               '--no-cpu-baseline', '--no-isolated']
        env = dict(os.environ, CAMLI_BENCH_DETAIL='bench_detail_%s.json' % name)
        if name == 'ddp4':
            cmd = [sys.executable, os.path.abspath(__file__), '--config', 'camliraft', '--batch', '4', '--steps', '5', '--warmup', '2',
                   '--no-cpu-baseline', '--no-isolated', '--no-side-configs']
            env.update(CAMLI_FORCE_DIST='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left - 15)
            rec = json.loads(res.stdout.strip().splitlines()[-1])
            out[name] = {'ms_per_step': rec['ms_per_step'], 'value': rec['value'], 'dtype': rec['dtype'], 'steps': rec['steps']}
            if name == 'ddp4':
                out[name].update(batch=4, n_iters=12, sync_bn=True, ranks=1, collectives='1-rank RCCL group',
                                 hip_graph=rec.get('config', {}).get('hip_graph'),
                                 host_enqueue_ms=rec.get('config', {}).get('host_enqueue_ms_per_step'))
        except subprocess.TimeoutExpired:
            out[name] = {'skipped': 'did not finish in %.0f s' % (left - 15)}
        except Exception as exc:      # noqa: BLE001 -- an optional leg never costs the line
            out[name] = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:80])}
        _log('side configuration %s: %s (%.0f s)' % (name, out[name], time.perf_counter() - t0))
    return out


def pmc_traffic(entry_point, args):
    """(HBM bytes per launch of `entry_point`, where the figure comes from): the committed PMC measurement of this same
    workload (profiles/roofline_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py, corrected
    as MI355X_MICROARCH.md prescribes) -- a STORED figure of an earlier pass, not a measurement of this run, and the line says
    so.  (None, None) when no matching record exists."""
    path = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
    if not os.path.exists(path):
        return None, None
    want = {'batch': args.batch, 'iters': args.iters, 'height': args.height, 'width': args.width, 'points': args.points}
    for rec in json.load(open(path)):
        if rec.get('entry_point') == entry_point and rec.get('workload') == want:
            return rec['traffic_bytes_per_launch'], 'stored PMC pass %s (profiles/roofline_traffic.json), not this run' % rec.get('round', 'r03')
    return None, None


def roofline_report(summary, steps, args, step_ms):
    """summary: _lib.TIMER.summary().  Returns (roofline object, per-kernel table).  The roofline kernel is the
    DOMINANT north-star entry point: the one holding the most device time in the timed region (a stable choice -- the
    lowest-fraction one flips between kernels of similar fraction from run to run; every kernel's fraction is in
    `hip_kernels` and, measured alone, in `roofline_rows`)."""
    table = {}
    fracs = {}
    for name, rec in sorted(summary.items(), key=lambda kv: -kv[1]['total_ms']):
        secs = rec['total_ms'] * 1e-3
        rate = rec['work'] / secs if secs > 0 else 0.0
        entry = {'launches_per_step': round(rec['launches'] / steps, 1), 'ms_per_step': round(rec['total_ms'] / steps, 3),
                 'avg_launch_us': round(rec['total_ms'] / rec['launches'] * 1e3, 2),
                 'rate': round(rate / 1e9, 2), 'rate_unit': 'G' + rec['unit'] + '/s'}
        kind = NORTH_STAR.get(name)
        if kind == 'hbm':
            entry['frac'] = round(rate / 1e9 / HBM_PEAK_GBS, 4)
        elif kind == 'mfma' and rec.get('flop', 0) > 0:
            if name == 'camli_allpairs_build_bwd':
                # the adjoint skips the gradient blocks no lookup visited: the flop it EXECUTES is only known from the
                # visit marks on the device (tools/kernel_bench.py reads them back -> roofline_rows); in situ only the
                # dense-product equivalent is known, and that is not a roofline fraction
                entry['dense_equivalent_tflops'] = round(rec['flop'] / secs / 1e12, 2)
            else:
                entry['tflops'] = round(rec['flop'] / secs / 1e12, 2)
                entry['frac'] = round(entry['tflops'] / MFMA_F32_PEAK_TFLOPS, 4)
        elif kind == 'fma' and rec.get('flop', 0) > 0:      # un-packed fp32 FMA: half the quoted (v_pk_fma_f32) vector peak
            entry['tflops'] = round(rec['flop'] / secs / 1e12, 2)
            entry['frac'] = round(entry['tflops'] / (MFMA_F32_PEAK_TFLOPS / 2), 4)
        elif kind == 'valu':
            entry['frac'] = round(rate / 1e9 / VALU_PAIR_PEAK_G, 4)
        elif kind == 'fps':          # dependent selection steps: ideal time per step / achieved time per step
            us_per_step = secs * 1e6 / rec['flop'] if rec.get('flop') else None     # `flop` carries the dependent steps (wrapper.py)
            if us_per_step:
                entry['us_per_step'] = round(us_per_step, 3)
                entry['frac'] = round(min(FPS_STEP_IDEAL_US / us_per_step, 1.0), 4)
        if 'frac' in entry:
            fracs[name] = rec['total_ms']
        table[name] = entry
    if not fracs:
        return None, table
    # the DOMINANT north-star entry point (most device time in the timed region), whatever bounds it: round 3 admitted only
    # hbm / mfma kinds here and camli_knn (valu) could never be named although it was the largest entry
    name = max(fracs, key=fracs.get)
    rec = summary[name]
    secs = rec['total_ms'] * 1e-3
    kind = NORTH_STAR[name]
    if kind == 'hbm':
        achieved, peak, unit, per_launch = rec['work'] / secs / 1e9, HBM_PEAK_GBS, 'GB/s', rec['work'] / rec['launches']
    elif kind == 'mfma':
        achieved, peak, unit, per_launch = rec['flop'] / secs / 1e12, MFMA_F32_PEAK_TFLOPS, 'TFLOP/s', rec['flop'] / rec['launches']
    elif kind == 'fma':
        achieved, peak, unit, per_launch = rec['flop'] / secs / 1e12, MFMA_F32_PEAK_TFLOPS / 2, 'TFLOP/s', rec['flop'] / rec['launches']
    elif kind == 'valu':     # candidate pairs per second against the fp32 vector rate at ~10 lane-operations per pair
        achieved, peak, unit, per_launch = rec['work'] / secs / 1e9, VALU_PAIR_PEAK_G, 'Gpairs/s', rec['work'] / rec['launches']
    else:                    # fps: fraction of the register-resident ideal step time
        achieved, peak, unit, per_launch = table[name]['frac'] * 100.0, 100.0, '% of ideal step rate', rec['work'] / rec['launches']
    # the dominant entry point of the path north_star names, next to the overall dominant one (which since
    # round 5 is a convolution-side kernel, 8(f)2)
    star = [n for n in fracs if n not in CONV_SIDE]
    star = max(star, key=fracs.get) if star else None
    traffic, traffic_source = pmc_traffic(name, args)
    roofline = {'kernel': name, 'bound': {'fma': 'valu', 'fps': 'latency'}.get(kind, kind), 'achieved': round(achieved, 2), 'peak': peak, 'unit': unit,
                'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_source': traffic_source, 'launches': rec['launches'],
                'avg_launch_us': round(rec['total_ms'] / rec['launches'] * 1e3, 2),
                'algorithmic_work_per_launch': round(per_launch), 'measured': 'in situ: HIP events on the launch stream, timed region'}
    # the north-star entry point FURTHEST below its roofline among those that hold at least 1 % of the step
    heavy = [n for n in fracs if summary[n]['total_ms'] / steps >= 0.01 * step_ms]
    if heavy:
        worst = min(heavy, key=lambda n: table[n]['frac'])
        roofline['worst'] = {'kernel': worst, 'bound': {'fma': 'valu', 'fps': 'latency'}.get(NORTH_STAR[worst], NORTH_STAR[worst]),
                             'frac': table[worst]['frac'], 'ms_per_step': table[worst]['ms_per_step']}
    if kind == 'mfma' and 'wino' in name and rec.get('flop', 0) > 0 and rec.get('unit') == 'B' and rec.get('work', 0) > 0:
        # an entry point of several kernels (transforms + contraction: the Winograd families) declares its matrix flop AND the bytes
        # its transform domain moves; its floor is both, un-overlapped -- `frac` above prices the flop alone
        floor_s = rec['flop'] / (MFMA_F32_PEAK_TFLOPS * 1e12) + rec['work'] / (HBM_PEAK_GBS * 1e9)
        roofline['floor'] = {'mfma_us': round(rec['flop'] / rec['launches'] / (MFMA_F32_PEAK_TFLOPS * 1e6), 1),
                             'hbm_us': round(rec['work'] / rec['launches'] / (HBM_PEAK_GBS * 1e3), 1), 'frac_in_situ': round(floor_s / secs, 4)}
    if star is not None:
        roofline['north_star'] = {'kernel': star, 'bound': {'fma': 'valu', 'fps': 'latency'}.get(NORTH_STAR[star], NORTH_STAR[star]),
                                  'frac_in_situ': table[star]['frac'], 'avg_launch_us_in_situ': table[star]['avg_launch_us'],
                                  'ms_per_step': table[star]['ms_per_step']}
    return roofline, table


def kernel_frac(name, rec):
    """roofline fraction of entry point `name` from a TIMER record (the arithmetic of roofline_report's table)"""
    kind = NORTH_STAR.get(name)
    secs = rec['total_ms'] * 1e-3
    if not secs or not rec['launches']:
        return None
    if kind == 'hbm':
        return rec['work'] / secs / 1e9 / HBM_PEAK_GBS
    if kind == 'mfma':
        return rec['flop'] / secs / 1e12 / MFMA_F32_PEAK_TFLOPS
    if kind == 'fma':
        return rec['flop'] / secs / 1e12 / (MFMA_F32_PEAK_TFLOPS / 2)
    if kind == 'valu':
        return rec['work'] / secs / 1e9 / VALU_PAIR_PEAK_G
    if kind == 'fps' and rec.get('flop'):
        return min(FPS_STEP_IDEAL_US / (secs * 1e6 / rec['flop']), 1.0)
    return None


def step_floor(step_fn, step_ms):
    """Where the STEP stands against the hardware: one extra, instrumented step after the timed region.
      mfma_flop  = floating-point work of every library convolution / GEMM of the step, forward and backward
                   (torch.utils.flop_counter over the aten operators) + the flop the own matrix-core kernels declare
      hbm_bytes  = the ALGORITHMIC bytes the own HBM-bound kernels declare (DESIGN section 5 formulas; the library's
                   streamed bytes are not counted -- its kernels are priced by their flop alone)
      floor_ms   = mfma_flop / 157.3 TFLOP/s + hbm_bytes / 8 TB/s       (no overlap of the two assumed)
      frac       = floor_ms / ms_per_step"""
    from torch.utils.flop_counter import FlopCounterMode
    from camliflow_amd.csrc import _lib
    _lib.TIMER.reset()
    only = _lib.TIMER.only
    _lib.TIMER.only = None
    _lib.TIMER.enabled = True
    counter = FlopCounterMode(display=False)
    with counter:
        step_fn()
    torch.cuda.synchronize()
    _lib.TIMER.enabled = False
    _lib.TIMER.only = only
    summary = _lib.TIMER.summary()
    _lib.TIMER.reset()
    lib_flop = float(counter.get_total_flops())
    # (the pyramid adjoint executes only the K steps the lookups marked -- ~20 % of its dense product -- and is left out: the
    # floor errs low, never high)
    own_flop = sum(r['flop'] for n, r in summary.items() if NORTH_STAR.get(n) == 'mfma' and n != 'camli_allpairs_build_bwd')
    own_bytes = sum(r['work'] for n, r in summary.items() if r['unit'] == 'B' and NORTH_STAR.get(n, 'hbm') == 'hbm')
    floor_ms = (lib_flop + own_flop) / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3 + own_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
    return {'mfma_flop': round(lib_flop + own_flop), 'library_flop': round(lib_flop), 'hbm_bytes': round(own_bytes),
            'floor_ms': round(floor_ms, 1), 'frac': round(min(floor_ms / step_ms, 1.0), 4)}


def write_detail(full, config='camliraft'):
    """The bulky parts of the report (per-kernel tables, isolated rows, census, long-form parity / baseline) go to a file;
    `gpurun_out/` is the directory that travels back from a GPU box."""
    out_dir = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        default = 'bench_detail.json' if config == 'camliraft' else 'bench_detail_%s.json' % config
        path = os.path.join(out_dir, os.environ.get('CAMLI_BENCH_DETAIL', default))
        with open(path, 'w') as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as err:
        sys.stderr.write('bench.py: could not write the detail file: %s\n' % err)
        return None


CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                 'vs_baseline', 'dtype', 'data', 'config')
LINE_LIMIT_BYTES = 4096


def compact_line(full, detail_path=None):
    """The ONE stdout line: contract fields + config + roofline (one kernel) + cpu_baseline + parity, < 4 KB by
    construction (the round-2 line was 20 KB and the driver's tail cut it).  No fraction above 1 is ever printed."""
    line = {k: full[k] for k in CONTRACT_KEYS if k in full}
    roof = full.get('roofline')
    if roof is not None:
        keep = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_in_situ', 'traffic', 'traffic_source', 'frac_single_lane',
                'avg_launch_us', 'avg_launch_us_in_situ', 'achieved_in_situ', 'launches', 'algorithmic_work_per_launch', 'measured', 'step', 'floor')
        line['roofline'] = {k: roof[k] for k in keep if k in roof}
        for sub in ('north_star', 'worst'):
            if sub in roof:
                line['roofline'][sub] = roof[sub]
        assert line['roofline']['frac'] <= 1.0, 'a roofline fraction above 1 is a mis-stated work figure'
    base = full.get('cpu_baseline')
    if base is not None:
        line['cpu_baseline'] = {k: base[k] for k in ('value', 'unit', 'cores', 'kind', 'sample', 'cpu_model', 'os_cpu_count', 'all_cores') if k in base}
    par = full.get('parity')
    if par is not None:
        line['parity'] = {k: par[k] for k in ('epe2d_abs_diff', 'epe3d_abs_diff', 'fps_equal', 'knn_equal', 'tolerance', 'ok')}
        line['parity']['vs'] = 'CPU port (cores + C oracle), batch-1 sample of the workload, shared post-IDS inputs'
    if full.get('side_configs'):
        line['side_configs'] = full['side_configs']
    if detail_path:
        line['detail'] = detail_path
    size = len(json.dumps(line))
    if size >= LINE_LIMIT_BYTES:      # never let free-text fields push the contract fields out of the driver's tail
        for key in ('parity', 'cpu_baseline'):
            if key in line and 'sample' in line[key]:
                line[key]['sample'] = line[key]['sample'][:120]
            if key in line and 'vs' in line[key]:
                line[key].pop('vs')
        line['config'].pop('lanes_note', None)
    assert len(json.dumps(line)) < LINE_LIMIT_BYTES
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: a 10-step window after 3 warm-up steps (+1.3 s over 5 / 2): the two lanes need a few steps to settle into their
    # steady alignment, and 5-step windows read 4-6 ms per step high (profiles/README.md, r05h vs r05b)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='camliraft')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (configs[2]: 8; configs[3] would be 4)')
    ap.add_argument('--iters', type=int, default=None)
    ap.add_argument('--height', type=int, default=None)
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--points', type=int, default=None)
    ap.add_argument('--mode', choices=['train', 'eval'], default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU port (cpu_baseline AND parity)')
    ap.add_argument('--no-isolated', action='store_true', help='skip the isolated per-kernel rows (roofline_rows)')
    ap.add_argument('--no-side-configs', action='store_true', help='skip the camlipwc / kitti side lines (side_configs)')
    ap.add_argument('--time-budget', type=float, default=float(os.environ.get('CAMLI_BENCH_BUDGET_S', 420)),
                    help='seconds the whole run may take: the legs after the timed region (isolated rows, CPU port, parity) '
                         'are skipped / cut when they would not fit, and the line says so; the timed region never is')
    ap.add_argument('--graph', action='store_true', default=None,
                    help='capture the whole training step in one HIP graph and replay it (single GPU); default for the '
                         'batch-1 configurations camlipwc / kitti, whose steps are bound by host enqueue time')
    ap.add_argument('--no-graph', dest='graph', action='store_false')
    ap.add_argument('--launch-check', action='store_true',
                    help='rendezvous only (gloo, no GPU): every rank joins, one all-reduce, rank 0 prints a line; '
                         'tests/test_bench_line.py uses it to cover the self-launcher on CPU')
    args = ap.parse_args()
    if os.environ.get('CAMLI_FAULT_DUMP'):      # debugging aid: dump every thread's Python stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['CAMLI_FAULT_DUMP']), exit=True)
    model_name, h, w, pts, iters, batch, mode, autocast, stands_for = CONFIGS[args.config]
    args.model = model_name
    args.height, args.width = args.height or h, args.width or w
    args.points, args.mode = args.points or pts, args.mode or mode
    args.iters = iters if args.iters is None else args.iters
    args.batch = args.batch or batch

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this process becomes the launcher of N ranks (train.py:307)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], device_count=args.gpus if args.launch_check else None))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.launch_check:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({'launch_check': True, 'n_gpus': world, 'rank_sum': t.item()}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path is the product, there is no CPU fallback')
    if local_rank >= torch.cuda.device_count():
        raise SystemExit('rank %d: LOCAL_RANK %d but only %d GPUs visible; refusing to oversubscribe'
                         % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # CAMLI_FORCE_DIST=1 under a 1-rank torch.distributed.run exercises the whole multi-GPU path
    # (RCCL init, broadcast, SyncBatchNorm, flat all-reduce) on a single-GPU box
    dist_on = world > 1 or (os.environ.get('CAMLI_FORCE_DIST') == '1' and 'RANK' in os.environ)
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # no device_id: binding the group to the device at construction (eager communicator) was measured to cost 8 % of
        # EVERY step in this process (293 vs 270 ms with identical work, idle communicator included); the communicator
        # is created by the first collective instead
        dist.init_process_group('nccl', rank=rank, world_size=world)  # RCCL over xGMI
    # lanes: two by default at every N (point branch on a side HIP stream; the first pass of a process is primed one-lane
    # by runtime.Lanes), so that the 1/2/4/8-GPU points of a scaling run are the same configuration.  CAMLI_OVERLAP=0 -> one.
    two_lane = os.environ.get('CAMLI_OVERLAP', '1') == '1'

    from camliflow_amd.cores import runtime
    from camliflow_amd.csrc import _lib
    _lib.load()
    runtime.set_backend('hip')
    runtime.set_overlap(two_lane)
    # parameter gradients of the iteration-shared 1x1 convolutions / biases accumulate inside their kernels and
    # reach .grad once per backward() (cores/runtime.py)
    runtime.set_deferred_param_grads(os.environ.get('CAMLI_DEFER_GRADS', '1') == '1')
    tuned_gemms = runtime.use_tuned_gemms()      # TunableOp with the shipped table (no tuning at run time)
    torch.backends.cudnn.benchmark = os.environ.get('CAMLI_MIOPEN_FIND', '0') == '1'

    torch.manual_seed(0)
    raw_model = build_model(args)
    state_dict = {k: v.clone() for k, v in raw_model.state_dict().items()}      # the parity legs start from these weights
    if dist_on and args.mode == 'train':
        raw_model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(raw_model)
    model = raw_model.to(device)
    model = model.train() if args.mode == 'train' else model.eval()
    if dist_on:   # identical replicas: broadcast rank 0's parameters and buffers once
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    if args.graph is None:
        # host-bound steps replay faster than they enqueue: the batch-1 configurations 1.6x (camlipwc) / 2.9x (kitti), the
        # batch-4 CamLiRAFT step (configs[3]'s per-rank batch) 175 -> 128 ms (r6, profiles/r06_experiments.txt); the batch-8
        # headline step is device-bound and eager there keeps the auxiliary streams (213.7 replayed vs 199.7 eager)
        args.graph = (args.config in ('camlipwc', 'kitti') or args.batch <= 4) and args.mode == 'train'
    use_graph = args.graph and world == 1
    optimizer = make_optimizer(model, capturable=use_graph) if args.mode == 'train' else None
    batch = {k: v.to(device) for k, v in synthetic_batch(args.batch, args.height, args.width, args.points,
                                                         seed=100 + rank, kitti=args.config == 'kitti').items()}

    def step():
        if args.mode == 'train':
            return train_step(model, optimizer, batch, world, dist_on, autocast)
        return eval_step(model, batch, autocast)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The first pass of this input signature runs on ONE stream whatever `lanes` says (runtime.set_overlap); it is untimed
    # and comes before the W warm-up steps.  It is also where a process pays the libraries' first-use work -- MIOpen /
    # hipBLASLt solution look-ups and code-object loads: 20-60 s on most boxes, 231 s measured on a slow one
    # (profiles/r03_first_step_probe.txt).  Round 2 mistook that stall for a two-stream dead-lock and restarted the run
    # from a watchdog; there is no watchdog any more, only this log line.
    if os.environ.get('CAMLI_PRIO_MAIN') and device.type == 'cuda':       # experiment: the step's own stream at a HIP priority
        torch.cuda.set_stream(torch.cuda.Stream(device, priority=int(os.environ['CAMLI_PRIO_MAIN'])))
    t_first = time.perf_counter()
    if not use_graph:
        step()
        torch.cuda.synchronize()
        _log('first step of the process (one lane, first-use work of the libraries included): %.1f s' % (time.perf_counter() - t_first))
    graphed = GraphedStep(step) if use_graph else None       # primes on its own side stream
    # The step is close to host-bound (169 ms of enqueue in a 178 ms step, profiles/r06i_ab_timer_graph.txt), and an event pair
    # around a launch costs ~6 us of host time: bracketing all ~780 north-star launches of a step inside the timed region cost
    # 5 ms of the step itself.  So the per-entry-point table (`hip_kernels`) is measured in the last warm-up steps (same
    # two-lane execution, every north-star entry point bracketed), and inside the timed region only the roofline kernel (the
    # dominant entry point of that table) and the dominant point-lane one (`roofline.north_star`) are bracketed.
    # CAMLI_TIME_ALL=1: every launch, in the timed region; CAMLI_NO_TIMER=1: none.
    warm_timed = 0
    time_all = os.environ.get('CAMLI_TIME_ALL') == '1'
    # (events cannot be recorded through a graph replay; only rank 0 reports, the other ranks record none)
    timer_on = graphed is None and rank == 0 and os.environ.get('CAMLI_NO_TIMER') != '1'
    if timer_on and not time_all:
        warm_timed = min(2, args.warmup)
    _lib.TIMER.reset()
    for i in range(args.warmup):
        if warm_timed and i == args.warmup - warm_timed:
            _lib.TIMER.only = set(NORTH_STAR)
            _lib.TIMER.enabled = True
        graphed() if graphed else step()
    warm_summary = None
    if warm_timed:
        torch.cuda.synchronize()
        _lib.TIMER.enabled = False
        warm_summary = _lib.TIMER.summary()
    barrier()
    runtime.set_census(True)
    runtime.reset_census()
    _lib.TIMER.reset()
    if time_all:
        _lib.TIMER.only = None
    elif warm_summary:
        picked, _ = roofline_report(warm_summary, warm_timed, args, 0.0)
        _lib.TIMER.only = {picked['kernel']} | ({picked['north_star']['kernel']} if picked.get('north_star') else set()) if picked else set()
    else:
        _lib.TIMER.only = set(NORTH_STAR)
    _lib.TIMER.enabled = timer_on
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = graphed() if graphed else step()
        host_s += time.perf_counter() - h0
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.TIMER.enabled = False
    census = runtime.census()
    runtime.set_census(False)
    if dist_on:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    roofline_steps, roofline_how = args.steps, None
    if graphed is not None and rank == 0:
        # events cannot be recorded through a replay: the per-kernel timings come from two eager steps after the
        # timed region (same model state, same streams)
        _lib.TIMER.reset()
        _lib.TIMER.enabled = True
        runtime.set_census(True)
        runtime.reset_census()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        _lib.TIMER.enabled = False
        census = runtime.census()
        runtime.set_census(False)
        roofline_steps, roofline_how = 2, 'two eager steps after the graph-replayed timed region (HIP events on the launch stream)'

    _flush_c_stdio()          # every rank: nothing of theirs may follow rank 0's line
    if dist_on:
        dist.barrier()
    failed = False
    if rank == 0:
        global_batch = args.batch * world
        step_ms = elapsed / args.steps * 1e3
        roofline, kernel_table = roofline_report(_lib.TIMER.summary(), roofline_steps, args, step_ms)
        if warm_summary:
            # the whole table from the warm-up steps; the rows bracketed in the timed region replace theirs, and `worst` is
            # taken over the whole table
            warm_roofline, warm_table = roofline_report(warm_summary, warm_timed, args, step_ms)
            warm_table.update(kernel_table)
            kernel_table = warm_table
            if roofline is None:
                roofline = warm_roofline
            elif warm_roofline is not None:
                if 'worst' in warm_roofline:
                    roofline['worst'] = warm_roofline['worst']
                roofline['table_measured'] = ('hip_kernels: the last %d warm-up steps (two lanes, every north-star entry point bracketed by HIP '
                                              'events); %s: the timed region' % (warm_timed, ' and '.join(sorted(_lib.TIMER.only))))
        if roofline is not None and roofline_how:
            roofline['measured'] = roofline_how
        if roofline is not None and runtime.overlap() and graphed is None and not dist_on:
            # the point-branch kernels of the timed region share the chip with the image branch's convolutions (two-lane
            # execution), so their in-situ durations carry that contention; two more steps single-lane give the same
            # kernel's duration with the chip to itself inside the same step -- the setting the committed rocprofv3
            # trace of this command is taken in (the profiler stalls the two-stream run on this image, profiles/README.md)
            runtime.set_overlap(False)
            _lib.TIMER.reset()
            _lib.TIMER.enabled = True
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            _lib.TIMER.enabled = False
            runtime.set_overlap(True)
            single = _lib.TIMER.summary()
            rec = single.get(roofline['kernel'])
            if rec and rec['launches']:
                secs = rec['total_ms'] * 1e-3
                rate = (rec['flop'] / secs / 1e12) if roofline['unit'] == 'TFLOP/s' else (rec['work'] / secs / 1e9)
                if roofline['bound'] == 'latency':
                    rate = min(FPS_STEP_IDEAL_US / (secs * 1e6 / rec['flop']), 1.0) * 100.0
                # WHICH FIGURE IS WHICH (round-5 review): `frac` / `achieved` / `avg_launch_us` are the SINGLE-LANE ones -- the
                # kernel with the chip to itself inside the step, the setting the committed rocprofv3 trace of this command is
                # taken in, so they are the ones to check against profiles/; `*_in_situ` are the two-lane figures of the timed
                # region (HIP events while the point lane shares the chip)
                roofline.update(frac_in_situ=roofline['frac'], achieved_in_situ=roofline['achieved'],
                                avg_launch_us_in_situ=roofline['avg_launch_us'])
                roofline.update(frac=round(rate / roofline['peak'], 4), achieved=round(rate, 2),
                                avg_launch_us=round(rec['total_ms'] / rec['launches'] * 1e3, 2),
                                measured='frac / achieved / avg_launch_us: 2 extra steps after the timed region, one lane (CAMLI_OVERLAP=0 '
                                         'semantics, = the rocprofv3 setting); *_in_situ: HIP events in the two-lane timed region')
                roofline['frac_single_lane'] = roofline['frac']          # (round-4 / round-5 name of the same figure)
                if 'floor' in roofline:
                    roofline['floor']['frac'] = round((rec['flop'] / (MFMA_F32_PEAK_TFLOPS * 1e12) + rec['work'] / (HBM_PEAK_GBS * 1e9)) / secs, 4)
            star = roofline.get('north_star')
            if star and single.get(star['kernel']) and single[star['kernel']]['launches']:
                rec = single[star['kernel']]
                f = kernel_frac(star['kernel'], rec)
                if f is not None:
                    star['frac'] = round(f, 4)
                    star['avg_launch_us'] = round(rec['total_ms'] / rec['launches'] * 1e3, 2)
        if roofline is not None and graphed is None and not dist_on and os.environ.get('CAMLI_NO_STEP_FLOOR') != '1':
            try:
                roofline['step'] = step_floor(step, step_ms)
            except Exception as exc:      # noqa: BLE001 -- an optional figure never costs the line
                _log('step floor FAILED: %s: %s' % (type(exc).__name__, str(exc)[:200]))
        what = {'train': 'training step (fwd + losses + bwd + clip + AdamW)', 'eval': 'inference forward'}[args.mode]
        metric = 'frame-pairs/sec (fwd+bwd) 960x540 + 8192 pts, CamLiRAFT' if args.config == 'camliraft' else \
                 'frame-pairs/sec, %s %s, %dx%d + %d pts' % (args.model, args.mode, args.width, args.height, args.points)
        line = {
            'metric': metric,
            'value': round(global_batch * args.steps / elapsed, 4),
            'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(step_ms, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if autocast is not None else 'f32', 'data': 'synthetic',
            'config': {'workload': '%s %s, %dx%d + %d pts, %s, batch %d per GPU (BASELINE %s)'
                                   % (args.model, what, args.width, args.height, args.points,
                                      ('%d GRU iters' % args.iters) if args.iters else 'coarse-to-fine pyramid',
                                      args.batch, stands_for),
                       'global_batch': global_batch, 'parallelism': 'dp%d' % world,
                       'lanes': 2 if runtime.overlap() else 1, 'hip_graph': bool(graphed), 'tuned_gemms': bool(tuned_gemms),
                       'loss': round(float(loss.detach()), 4),
                       'host_enqueue_ms_per_step': round(host_s / args.steps * 1e3, 1),
                       'hip_launches_per_step': round(sum(census['fused'].values()) / roofline_steps, 1)},
            'roofline': roofline,
            'hip_kernels': kernel_table,
            'census': {'fused_launches_per_step': {k: round(v / roofline_steps, 1) for k, v in sorted(census['fused'].items())},
                       'composed_calls_per_step': {k: round(v / roofline_steps, 1) for k, v in sorted(census['composed'].items())}},
        }
        _log('timed region done: %.2f ms per step' % step_ms)
        budget = args.time_budget
        if world == 1 and not args.no_isolated and args.config == 'camliraft':
            if _elapsed() < 0.5 * budget:
                import kernel_bench
                try:        # an optional leg: its failure is reported in the line, it never costs the line itself
                    line['roofline_rows'] = kernel_bench.run(batch=args.batch, reps=5)
                    _log('isolated kernel rows done')
                except Exception as exc:      # noqa: BLE001
                    line['roofline_rows_error'] = '%s: %s' % (type(exc).__name__, str(exc)[:300])
                    _log('isolated kernel rows FAILED: %s' % line['roofline_rows_error'])
            else:
                line['roofline_rows_skipped'] = 'time budget: %.0f of %.0f s were gone after the timed region' % (_elapsed(), budget)
                _log('isolated kernel rows skipped (time budget)')
        overrun = False
        if world == 1 and not args.no_cpu_baseline and args.model == 'camliraft':
            # free the bench model before the parity model is built
            del optimizer, graphed
            from camliflow_amd.csrc import fused as _fused
            _fused.release_cached_buffers()         # the clean gradient pyramid of the timed steps (2.8 GB)
            deadline = budget - 15.0
            got, why = run_with_deadline(lambda: cpu_baseline_and_reference(args, state_dict, deadline), deadline - _elapsed())
            if got is None:
                overrun = True
                line['cpu_baseline'] = {'value': None, 'unit': 'frame-pairs/s', 'cores': min(8, torch.get_num_threads()),
                                        'kind': 'port', 'sample': 'CPU port %s' % why}
                _log('CPU port abandoned: %s' % why)
            else:
                base, sample, ref = got
                line['cpu_baseline'] = base
                if autocast is None:
                    line['parity'] = parity_check(args, state_dict, sample, ref, device)
                    failed = not line['parity']['ok']
                    _log('parity check done')
        if world == 1 and args.config == 'camliraft' and not args.no_side_configs:
            line['side_configs'] = side_configs(budget)
        if (world == 1 and not args.no_cpu_baseline and args.model == 'camliraft' and not overrun and line.get('cpu_baseline', {}).get('value')
                and os.environ.get('CAMLI_CPU_ALL_CORES', '1') == '1'):
            # the all-physical-cores point of the CPU port, after everything else (it occupies every core): only when a
            # generous margin is left -- 128 threads took 50 s per step in round 2
            left = budget - 10.0 - _elapsed()
            if left >= 90.0:
                got, why = run_with_deadline(lambda: cpu_all_cores_point(args, state_dict), left)
                if got is None:
                    overrun = True
                    line['cpu_baseline']['all_cores'] = {'skipped': why}
                else:
                    line['cpu_baseline']['all_cores'] = got
                _log('CPU port at all physical cores: %s' % line['cpu_baseline']['all_cores'])
            else:
                line['cpu_baseline']['all_cores'] = {'skipped': 'time budget: %.0f s left' % left}
        detail_path = write_detail(line, args.config)
        # RCCL writes its version banner through C stdio when the communicator is created; flush it so that the JSON
        # line is the LAST line on stdout
        _flush_c_stdio()
        print(json.dumps(compact_line(line, detail_path)), flush=True)
        if overrun:         # a CPU-port thread is still inside a torch call: leave without joining it
            sys.stderr.flush()
            os._exit(0)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit('parity check FAILED: %s' % json.dumps(line['parity']))


if __name__ == '__main__':
    main()

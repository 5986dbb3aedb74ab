"""Headline benchmark: CamLiRAFT training step (forward + sequence losses + backward + clip + AdamW)
on synthetic FlyingThings3D-shaped inputs, 960x540 images + 8192 points (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU, batch-dimension data parallel (SyncBatchNorm + one flat-bucket gradient
all-reduce over RCCL/xGMI per step); per-GPU batch is fixed, so the scaling is weak.  Rank 0 prints ONE JSON line.  `value` = global frame-pairs per second with the
inputs resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from types import SimpleNamespace as NS  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# Every C-ABI launch of the timed region is bracketed by HIP events on its launch stream and carries
# its algorithmic work.  The roofline object reports the HBM-bound entry point
# with the largest total time; KNN (VALU-bound) and FPS (latency-bound) are listed beside it.


def model_cfg(n_iters):
    return NS(name='camliraft', batch_size=1, freeze_bn=False, backbone=NS(depth=50, pretrained=None),
              n_iters_train=n_iters, n_iters_eval=n_iters, fuse_fnet=True, fuse_cnet=True, fuse_corr=True,
              fuse_motion=True, fuse_hidden=False, loss2d=NS(gamma=0.8, order='l2-norm'),
              loss3d=NS(gamma=0.8, order='l2-norm'))


def synthetic_batch(b, h, w, n_points, seed):
    """uint8-valued images, points whose projections land inside the image, small flows."""
    g = torch.Generator().manual_seed(seed)
    f, cx, cy = 1050.0, 479.5, 269.5
    images = torch.randint(0, 256, (b, 6, h, w), generator=g).float()
    z = torch.rand(b, n_points, generator=g) * 30.0 + 5.0
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
    and replayed: removes the ~16k kernel launches per step from the host.  Both HIP streams of the
    two-lane execution are captured (the side stream forks from and joins the capture stream).
    Single GPU only (collectives stay outside graphs here)."""

    def __init__(self, model, optimizer, batch, warmup=3):
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                train_step(model, optimizer, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            self.loss = train_step(model, optimizer, batch)

    def __call__(self):
        self.graph.replay()
        return self.loss


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
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)


def train_step(model, optimizer, batch, world=1, force_dist=False):
    model(batch)
    loss = model.get_loss()
    loss.backward()
    allreduce_gradients(model, world, force_dist)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    model.clear_metrics()
    return loss


def cpu_baseline(args):
    """The CPU restatement path (this repo's cores driven by the C oracle operators) timed on the
    host cores: ONE training step at batch 1 of the same workload.  Reported, not the target."""
    from modelutils import oracle_boundary
    from camliflow_amd.cores import CamLiRAFT
    threads = torch.get_num_threads()
    torch.manual_seed(0)
    model = CamLiRAFT(model_cfg(args.iters)).train()
    opt = make_optimizer(model)
    batch = synthetic_batch(1, args.height, args.width, args.points, seed=1)
    with oracle_boundary():
        t0 = time.perf_counter()
        train_step(model, opt, batch)
        dt = time.perf_counter() - t0
    return {'value': 1.0 / dt, 'unit': 'frame-pairs/s', 'cores': threads, 'kind': 'port',
            'sample': '1 training step (fwd+bwd+AdamW), batch 1, %dx%d + %d pts, %d iters, %.1f s, cold'
                      % (args.width, args.height, args.points, args.iters, dt)}


def pmc_traffic(entry_point, args):
    """HBM bytes per launch of `entry_point` from the committed PMC measurement of this same workload
    (profiles/roofline_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    bench.py, corrected as MI355X_MICROARCH.md prescribes).  None when no matching record exists."""
    path = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
    if not os.path.exists(path):
        return None
    want = {'batch': args.batch, 'iters': args.iters, 'height': args.height, 'width': args.width, 'points': args.points}
    for rec in json.load(open(path)):
        if rec.get('entry_point') == entry_point and rec.get('workload') == want:
            return rec['traffic_bytes_per_launch']
    return None


def isolated_dw_fwd(batch, device):
    """The roofline kernel alone on an idle GPU (its largest shape in the step: C=128, k=32,
    2048 points): context for the in-situ figure, which is measured while the image branch's
    convolutions share the chip with it on another stream."""
    from camliflow_amd.csrc import fused
    c, n, k = 128, 2048, 32
    feat = torch.randn(batch, c, n, device=device)
    shared = fused.SharedSetConvWeights(torch.rand(batch, c, n, k, device=device))
    idx = torch.randint(0, n, (batch, n, 32), device=device)
    for _ in range(3):
        fused.pointconv_dw(feat, shared, idx, k)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    start.record()
    for _ in range(reps):
        fused.pointconv_dw(feat, shared, idx, k)
    end.record()
    torch.cuda.synchronize()
    us = start.elapsed_time(end) / reps * 1e3
    byt = 4.0 * batch * c * n * k + 4.0 * batch * c * n + 8.0 * batch * n * k + 5.0 * batch * c * n
    return {'shape': 'B%d C%d N%d k%d, inference form' % (batch, c, n, k), 'avg_launch_us': round(us, 2),
            'achieved': round(byt / us / 1e3, 1), 'frac': round(byt / us / 1e3 / HBM_PEAK_GBS, 4)}


def roofline_report(summary, steps, args):
    """summary: _lib.TIMER.summary().  Returns (roofline object, per-kernel table)."""
    table = {}
    for name, rec in sorted(summary.items(), key=lambda kv: -kv[1]['total_ms']):
        rate = rec['work'] / (rec['total_ms'] * 1e-3) if rec['total_ms'] > 0 else 0.0
        table[name] = {'launches_per_step': round(rec['launches'] / steps, 1),
                       'ms_per_step': round(rec['total_ms'] / steps, 3),
                       'avg_launch_us': round(rec['total_ms'] / rec['launches'] * 1e3, 2),
                       'rate': round(rate / 1e9, 2), 'rate_unit': 'G' + rec['unit'] + '/s'}
    hbm = [(n, r) for n, r in summary.items() if r['unit'] == 'B' and r['total_ms'] > 0]
    if not hbm:
        return None, table
    name, rec = max(hbm, key=lambda kv: kv[1]['total_ms'])
    achieved = rec['work'] / (rec['total_ms'] * 1e-3) / 1e9
    roofline = {'kernel': name, 'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(name, args),
                'launches': rec['launches'], 'avg_launch_us': round(rec['total_ms'] / rec['launches'] * 1e3, 2),
                'algorithmic_bytes_per_launch': round(rec['work'] / rec['launches'])}
    return roofline, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch (configs[2]: 8)')
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--height', type=int, default=540)
    ap.add_argument('--width', type=int, default=960)
    ap.add_argument('--points', type=int, default=8192)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='capture the whole training step in one HIP graph (single GPU)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path is the product, there is no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # CAMLI_FORCE_DIST=1 under a 1-rank torch.distributed.run exercises the whole multi-GPU path
    # (RCCL init, broadcast, SyncBatchNorm, flat all-reduce) on a single-GPU box
    dist_on = world > 1 or (os.environ.get('CAMLI_FORCE_DIST') == '1' and 'RANK' in os.environ)
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)  # RCCL over xGMI

    from camliflow_amd.cores import CamLiRAFT, runtime
    from camliflow_amd.csrc import _lib
    _lib.load()
    runtime.set_backend('hip')
    runtime.set_overlap(os.environ.get('CAMLI_OVERLAP', '1') == '1')
    # parameter gradients of the iteration-shared 1x1 convolutions / biases accumulate inside their kernels and
    # reach .grad once per backward() (cores/runtime.py): 119 parameters, ~1,200 fewer add / sum launches per
    # step; same-box A/B at batch 8: 306.0 vs 308.0 ms (3 runs each), 162 vs 180 ms at batch 2
    runtime.set_deferred_param_grads(os.environ.get('CAMLI_DEFER_GRADS', '1') == '1')
    torch.backends.cudnn.benchmark = os.environ.get('CAMLI_MIOPEN_FIND', '0') == '1'

    torch.manual_seed(0)
    raw_model = CamLiRAFT(model_cfg(args.iters))
    if dist_on:
        raw_model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(raw_model)
    model = raw_model.to(device).train()
    if dist_on:   # identical replicas: broadcast rank 0's parameters and buffers once
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    use_graph = args.graph and world == 1
    optimizer = make_optimizer(model, capturable=use_graph)
    batch = {k: v.to(device) for k, v in synthetic_batch(args.batch, args.height, args.width, args.points,
                                                         seed=100 + rank).items()}

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    graphed = GraphedStep(model, optimizer, batch) if use_graph else None
    for _ in range(args.warmup):
        graphed() if graphed else train_step(model, optimizer, batch, world, dist_on)
    barrier()
    _lib.TIMER.reset()
    _lib.TIMER.only = None
    _lib.TIMER.enabled = graphed is None and os.environ.get('CAMLI_NO_TIMER') != '1'   # events cannot be recorded through a graph replay
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = graphed() if graphed else train_step(model, optimizer, batch, world, dist_on)
        host_s += time.perf_counter() - h0
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.TIMER.enabled = False
    if dist_on:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        global_batch = args.batch * world
        roofline, kernel_table = roofline_report(_lib.TIMER.summary(), args.steps, args)
        if roofline and roofline['kernel'] == 'camli_pointconv_dw_fwd':
            roofline['isolated'] = isolated_dw_fwd(args.batch, device)
        line = {
            'metric': 'frame-pairs/sec (fwd+bwd) 960x540 + 8192 pts, CamLiRAFT',
            'value': round(global_batch * args.steps / elapsed, 4),
            'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'CamLiRAFT training step (fwd + sequence losses + bwd + clip + AdamW), '
                                   '%dx%d + %d pts, %d GRU iters, batch %d per GPU (BASELINE configs[2])'
                                   % (args.width, args.height, args.points, args.iters, args.batch),
                       'global_batch': global_batch, 'parallelism': 'dp%d' % world, 'hip_graph': bool(graphed),
                       'loss': round(float(loss.detach()), 4),
                       'host_enqueue_ms_per_step': round(host_s / args.steps * 1e3, 1)},
            'roofline': roofline,
            'hip_kernels': kernel_table,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Isolated per-kernel roofline rows of the north-star operators at the shapes of the headline step
(BASELINE configs[2]: CamLiRAFT 960x540 + 8192 points, batch 8) plus the PWC correlation shapes of configs[1].

Every case drives the C-ABI entry point through the same Python wrappers the cores use; each launch is bracketed
by HIP events on its launch stream (csrc/_lib.KernelTimer) and carries its ALGORITHMIC work,
so a row is   achieved = algorithmic work / average launch duration   on an otherwise idle GPU (no second lane).

  python tools/kernel_bench.py [--batch 8] [--reps 20] [--only substr] [--json out.json]

bench.py imports ``run()`` for its ``roofline_rows``; the committed rocprofv3 --kernel-trace --stats summary of
THIS command (profiles/r02_kernel_bench_*) is what the per-row durations are checked against.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK = 8000.0          # GB/s, MI355X_MICROARCH.md (spec; ~6.3 TB/s is the measured copy ceiling)
MFMA_F32_PEAK = 157.3      # TFLOP/s, fp32-input MFMA = the fp32 vector rate
VALU_PAIR_PEAK = 7865.0    # Gpairs/s: 157.3 TFLOP/s / 2 flop per lane-op / ~10 lane-ops per candidate pair (3-D)
LDS_STEP_IDEAL_US = 0.35   # FPS: one dependent step with the cloud resident in registers (update + arg-max + 1 barrier)


def _rand(g, *shape, scale=1.0):
    return (torch.rand(*shape, generator=g) * scale).cuda()


def _randn(g, *shape):
    return torch.randn(*shape, generator=g).cuda()


def cases(batch):
    """yield (row name, callable running the op once, {entry point: bound kind})"""
    from camliflow_amd import csrc
    from camliflow_amd.csrc import fused, wrapper
    g = torch.Generator(device='cpu').manual_seed(0)
    b = batch

    # ---- A1 correlation2d, PWC pyramid shapes (configs[1], batch 1) + the reference self-check shape -------------
    for (cb, c, h, w) in [(1, 32, 144, 240), (1, 64, 72, 120), (1, 96, 36, 60), (1, 128, 18, 30), (1, 192, 9, 15),
                          (32, 128, 144, 240)]:
        in1 = _randn(g, cb, h, w, c).requires_grad_(True)
        in2 = _randn(g, cb, h, w, c).requires_grad_(True)
        go = _randn(g, cb, 81, h, w)

        def corr(in1=in1, in2=in2, go=go):
            out = wrapper.CorrelationFunction.apply(in1, in2, 4)
            torch.autograd.grad(out, [in1, in2], go)
        yield 'corr2d B%d C%d %dx%d' % (cb, c, h, w), corr, {'camli_corr2d_fwd': 'hbm', 'camli_corr2d_bwd': 'hbm'}

    # ---- A2 / A3 all-pairs volume: build (+ adjoint) and the radius-4 lookup (+ adjoint) --------------------------
    h, w = 68, 120
    f1 = _randn(g, b, 256, h, w).requires_grad_(True)
    f2 = _randn(g, b, 256, h, w).requires_grad_(True)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    # a flow field as the GRU produces it: smooth (a coarse random field up-sampled 8x, +-6 px) plus sub-pixel noise.
    # White noise of several pixels per source pixel would scatter the windows of neighbouring pixels over 3x as many
    # rows of the volume as any real flow does (it matters for the adjoint, which skips the never-visited blocks).
    coarse = torch.randn(b, 2, (h + 7) // 8 + 1, (w + 7) // 8 + 1, generator=g) * 3
    flow = torch.nn.functional.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=True)
    coords = (torch.stack([xs, ys])[None].repeat(b, 1, 1, 1) + flow + torch.randn(b, 2, h, w, generator=g) * 0.25).cuda()
    go_lookup = _randn(g, b, 324, h, w)

    def allpairs():
        pyr = fused.allpairs_pyramid(f1, f2, 4)
        out = fused.allpairs_lookup(pyr, coords, 4)
        torch.autograd.grad(out, [f1, f2], go_lookup)
    def visited_flop():
        """fp32 flop the adjoint GEMMs actually execute: K steps of 128x128x32 whose gradient tile holds a visit mark
        (2 row tiles each), from the marks of one backward pass -- the rest of the dense product is never computed."""
        keep = {}
        pyr = fused.allpairs_pyramid(f1, f2, 4)
        out = fused.allpairs_lookup(pyr, coords, 4)
        hook = pyr.token.register_hook(lambda _g: keep.update(marks=[m.clone() for m in pyr.marks]))
        torch.autograd.grad(out, [f1, f2], go_lookup)
        hook.remove()
        steps = 0
        for m in keep['marks']:
            nb, sb, tb = m.shape
            live = m != 0
            src4 = torch.nn.functional.pad(live, (0, 0, 0, (-sb) % 4)).reshape(nb, -1, 4, tb).any(dim=2)     # g_f1 tiles
            tgt4 = torch.nn.functional.pad(live, (0, (-tb) % 4)).reshape(nb, sb, -1, 4).any(dim=3)            # g_f2 tiles
            steps += int(src4.sum()) + int(tgt4.sum())
        return steps * 2 * (2.0 * 128 * 128 * 32)
    allpairs.flop_override = {'camli_allpairs_build_bwd': visited_flop}
    yield 'allpairs B%d 68x120' % b, allpairs, {'camli_allpairs_build_fwd': 'mfma', 'camli_allpairs_build_bwd': 'mfma',
                                                'camli_allpairs_fold_bwd': 'hbm',
                                                'camli_allpairs_lookup_fwd': 'hbm', 'camli_allpairs_lookup_bwd': 'hbm'}

    # ---- (f)2 GRU2D: one update (both half-steps) on the channels-last matrix-core kernels, forward + backward (r5) ----
    # (batch 8 = the headline step; batch 4 = configs[3]'s per-rank batch, 47 x 156 = KITTI's batch-1 map: there the pixel tiles
    # alone leave the chip half empty -- the shapes the r6 tile selection is about)
    from camliflow_amd.cores.raft2d import GRU2D
    gru = GRU2D(hidden_dim=128, input_dim=256).cuda()
    for (gb, gh_, gw_) in ([(b, h, w), (4, 68, 120), (1, 47, 156)] if b == 8 else [(b, h, w)]):
        gh0 = torch.tanh(_randn(g, gb, 128, gh_, gw_)).requires_grad_(True)
        gctx = _randn(g, gb, 128, gh_, gw_)
        gmot = _randn(g, gb, 128, gh_, gw_).requires_grad_(True)
        ggo = _randn(g, gb, 128, gh_, gw_)

        def gru2d(gh0=gh0, gctx=gctx, gmot=gmot, ggo=ggo, gstate={}):
            if not gstate:
                gstate['s'] = gru.prepare(gctx)
            out = gru.step(gh0, gmot, gstate['s'])
            torch.autograd.grad(out, [gh0, gmot] + [p_ for p_ in gru.parameters() if p_.dim() == 4], ggo, retain_graph=True, allow_unused=True)
        yield 'gru2d B%d %dx%d' % (gb, gh_, gw_), gru2d, {'camli_convcl_gru_gates': 'mfma', 'camli_convcl_gru_blend': 'mfma',
                                                        'camli_convcl_fwd': 'mfma', 'camli_convcl_wrw': 'mfma',
                                                        'camli_wino1d_gru_gates': 'mfma', 'camli_wino1d_gru_blend': 'mfma',
                                                        'camli_wino1d_conv': 'mfma', 'camli_wino1d_wrw': 'mfma'}

    # ---- (f)2 the update block's 3x3 convolutions as Winograd F(2x2,3x3) (csrc/hip/winograd.hip): forward, data gradient,
    # weight gradient per launch of the entry point (three / four kernels each; flop = the transform-domain MFMA work)
    for (cin, cout) in [(256, 192), (256, 126), (128, 256)]:
        wx = _randn(g, b, cin, h, w).requires_grad_(True)
        ww_ = (_randn(g, cout, cin, 3, 3) * (9 * cin) ** -0.5).requires_grad_(True)
        wgo = _randn(g, b, cout, h, w)

        def wino(wx=wx, ww_=ww_, wgo=wgo):
            out = fused.conv3x3_wino(wx, ww_)
            torch.autograd.grad(out, [wx, ww_], wgo)
        yield 'wino3x3 B%d %d->%d 68x120' % (b, cin, cout), wino, {'camli_wino_conv3x3': 'mfma', 'camli_wino_wrw': 'mfma'}

    # ---- A4 furthest point sampling ---------------------------------------------------------------------------
    xyz = _rand(g, 2 * b, 8192, 3, scale=10.0)
    yield 'fps B%d 8192->4096' % (2 * b), (lambda: csrc.furthest_point_sampling(xyz, 4096)), {'camli_fps': 'fps'}

    # ---- A6 KNN at every shape of the step -----------------------------------------------------------------------
    for (m, nq, d, k) in [(8192, 4096, 3, 16), (4096, 2048, 3, 16), (2048, 2048, 3, 32), (2048, 2048, 3, 16),
                          (1024, 2048, 3, 16), (512, 2048, 3, 16), (256, 2048, 3, 16), (2048, 2048, 3, 3),
                          (2048, 1024, 3, 3), (2048, 8192, 3, 3), (2048, 8160, 2, 1)]:
        inp, qry = _rand(g, b, m, d, scale=10.0), _rand(g, b, nq, d, scale=10.0)
        yield ('knn B%d M%d Nq%d D%d k%d' % (b, m, nq, d, k),
               (lambda inp=inp, qry=qry, k=k: csrc.k_nearest_neighbor(inp, qry, k)), {'camli_knn': 'valu'})

    # the four nested cross searches of a GRU iteration, one launch (camli_knn_prefixes)
    inp_p, qry_p = _rand(g, b, 2048, 3, scale=10.0), _rand(g, b, 2048, 3, scale=10.0)
    yield ('knn prefixes B%d 2048/1024/512/256 Nq2048 k16' % b,
           (lambda: wrapper.k_nearest_neighbor_prefixes(inp_p, qry_p, (2048, 1024, 512, 256), 16)), {'camli_knn': 'valu'})

    # the same search one GRU iteration later (11 of the 12 per step): the target cloud moved by ~1 % of its extent, the previous
    # result passed as prior (camli_knn_prefixes_prior: a bound on every k-th distance, identical indices)
    inp_m = (inp_p + _rand(g, b, 2048, 3, scale=0.2) - 0.1).contiguous()
    prior_p = wrapper.k_nearest_neighbor_prefixes(inp_p, qry_p, (2048, 1024, 512, 256), 16)
    yield ('knn prefixes B%d 2048/1024/512/256 Nq2048 k16, prior = the result one iteration earlier' % b,
           (lambda: wrapper.k_nearest_neighbor_prefixes(inp_m, qry_p, (2048, 1024, 512, 256), 16, prior=prior_p)), {'camli_knn': 'valu'})

    # the dense-query interpolation of kitti_submission.py:89-93 (every pixel of a 375x1242 map)
    inp1, qry1 = _rand(g, 1, 8192, 3, scale=10.0), _rand(g, 1, 465750, 3, scale=10.0)
    yield 'knn B1 M8192 Nq465750 D3 k3', (lambda: csrc.k_nearest_neighbor(inp1, qry1, 3)), {'camli_knn': 'valu'}

    # ---- A8 gather / scatter (the cost-volume pooling gathers of Correlation3D.build) ---------------------------
    for (c, m, i) in [(2048, 2048, 3072), (128, 2048, 2048 * 16)]:
        data = _randn(g, b, c, m).requires_grad_(True)
        idx = torch.randint(0, m, (b, i), generator=g).cuda()
        go = _randn(g, b, c, i)

        def gather(data=data, idx=idx, go=go):
            out = fused.gather_points(data, idx)
            torch.autograd.grad(out, data, go)
        yield 'gather_cf B%d C%d M%d I%d' % (b, c, m, i), gather, {'camli_gather_cf_fwd': 'hbm', 'camli_gather_cf_bwd': 'hbm'}

    # ---- A9 PointConv mixing (Encoder3D) -------------------------------------------------------------------------
    for (m, n, ch) in [(8192, 4096, 99), (4096, 2048, 131)]:
        feat = _randn(g, b, m, ch).requires_grad_(True)
        wgt = _rand(g, b, 16, n, 16).requires_grad_(True)
        idx = torch.randint(0, m, (b, n, 16), generator=g).cuda()
        go = _randn(g, b, n, 16, ch)

        def mix(feat=feat, wgt=wgt, idx=idx, go=go):
            out = fused.pointconv_mix(feat, wgt, idx, 16)
            torch.autograd.grad(out, [feat, wgt], go)
        yield 'pointconv_mix B%d M%d n%d CH%d' % (b, m, n, ch), mix, {'camli_pointconv_mix_fwd': 'hbm', 'camli_pointconv_mix_bwd': 'hbm'}

    # ---- A10 PointConvDW: fused gather * weight -> max, and the weight network on the matrix cores ---------------
    n = 2048
    # weights k-major [B,C,k,N] -- the layout of the product path (cores/setconv.py) -- and, as '(nk)' rows, the
    # reference's [B,C,N,k] layout the other kernels of the file serve
    for (c, k, k_major) in [(128, 32, True), (128, 16, True), (128, 4, True), (128, 32, False), (128, 16, False), (128, 4, False)]:
        feat = _randn(g, b, c, n).requires_grad_(True)
        wgt = (_rand(g, b, c, k, n) if k_major else _rand(g, b, c, n, k)).requires_grad_(True)
        idx = torch.randint(0, n, (b, n, 32), generator=g).cuda()
        go = _randn(g, b, c, n)

        def dw(feat=feat, wgt=wgt, idx=idx, go=go, k=k, k_major=k_major):
            shared = fused.SharedSetConvWeights(wgt, k_major=k_major)
            out = fused.pointconv_dw(feat, shared, idx, k)
            torch.autograd.grad(out, [feat, wgt], go)
        yield 'pointconv_dw B%d C%d k%d%s' % (b, c, k, '' if k_major else ' (nk)'), dw, {
            'camli_pointconv_dw_fwd': 'hbm', 'camli_pointconv_dw_bwd': 'hbm', 'camli_pointconv_dw_expand': 'hbm'}
    from camliflow_amd.cores.blocks import MLP2d
    xyzc = _rand(g, b, 3, n, scale=10.0)
    for k in (16, 32):
        mlp = MLP2d(3, [8, 32, 128], act='relu').cuda()
        idx = torch.randint(0, n, (b, n, 32), generator=g).cuda()
        go = _randn(g, b, 128, k, n)

        def wn(mlp=mlp, idx=idx, go=go, k=k):
            out = fused.weightnet(xyzc, xyzc, idx, k, mlp, k_major=True)
            torch.autograd.grad(out, list(mlp.parameters()), go)
        yield 'weightnet B%d C128 k%d' % (b, k), wn, {'camli_weightnet_fwd': 'hbm', 'camli_weightnet_bwd': 'mfma_wn'}

    # ---- A11 / A12 interpolation and the point cost-volume gather ------------------------------------------------
    in_xyz, q_xyz = _rand(g, b, 3, 2048, scale=10.0), _rand(g, b, 3, 8192, scale=10.0)
    feat = _randn(g, b, 3, 2048).requires_grad_(True)
    knn3 = torch.randint(0, 2048, (b, 8192, 3), generator=g).cuda()
    go = _randn(g, b, 3, 8192)

    def interp():
        out = fused.knn_interpolate(in_xyz, feat, q_xyz, knn3, 3, invariant=True)      # the GRU loops' case: same clouds every call
        torch.autograd.grad(out, feat, go)
    yield 'knn_interp B%d C3 M2048 Nq8192' % b, interp, {'camli_knn_interp_fwd': 'hbm', 'camli_knn_interp_bwd': 'hbm'}

    cost = _randn(g, b, 2048, 2048).requires_grad_(True)
    x1, x2 = _rand(g, b, 3, 2048), _rand(g, b, 3, 2048)
    knn16 = torch.randint(0, 2048, (b, 2048, 16), generator=g).cuda()
    go4 = _randn(g, b, 4, 2048, 16)

    def c3d():
        out = fused.corr3d_lookup_input(cost, x1, x2, knn16)
        torch.autograd.grad(out, cost, go4)
    yield 'corr3d_gather B%d N2048 k16' % b, c3d, {'camli_corr3d_gather_fwd': 'hbm', 'camli_corr3d_gather_bwd': 'hbm'}

    # ---- A12 cost MLP (4 -> 32 -> 32, ReLU) + neighbour sum of the lookup: fp32 FMA work, everything in registers ------
    conv1, conv2 = torch.nn.Conv2d(4, 32, 1).cuda(), torch.nn.Conv2d(32, 32, 1).cuda()
    look = _randn(g, b, 4, 2048, 64).requires_grad_(True)
    go_mlp = _randn(g, b, 128, 2048)

    def cost_mlp():
        out = fused.corr3d_cost_mlp(look, conv1, conv2, 4)
        torch.autograd.grad(out, [look, conv1.weight, conv1.bias, conv2.weight, conv2.bias], go_mlp)
    yield 'corr3d_mlp B%d N2048 4x16' % b, cost_mlp, {'camli_corr3d_mlp_fwd': 'fma', 'camli_corr3d_mlp_bwd': 'fma'}


def _row(case, name, kind, rec, fps_steps=None):
    us = rec['total_ms'] / rec['launches'] * 1e3
    work = rec['work'] / rec['launches']
    row = {'case': case, 'kernel': name, 'avg_launch_us': round(us, 2), 'launches': rec['launches'],
           'algorithmic_work_per_launch': work, 'work_unit': rec['unit']}
    if kind == 'hbm':
        ach = work / us / 1e3
        row.update(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK, unit='GB/s', frac=round(ach / HBM_PEAK, 4))
    elif kind in ('mfma', 'mfma_wn'):
        flop = rec.get('flop', 0.0) / rec['launches']
        ach = flop / us / 1e6
        row.update(bound='mfma', achieved=round(ach, 2), peak=MFMA_F32_PEAK, unit='TFLOP/s', frac=round(ach / MFMA_F32_PEAK, 4),
                   flop_per_launch=flop)
    elif kind == 'fma':
        # plain (un-packed) fp32 FMA on the vector ALU: half the quoted 157.3 TFLOP/s, which counts v_pk_fma_f32
        flop = rec.get('flop', 0.0) / rec['launches']
        ach = flop / us / 1e6
        row.update(bound='valu-fma', achieved=round(ach, 2), peak=MFMA_F32_PEAK / 2, unit='TFLOP/s',
                   frac=round(ach / (MFMA_F32_PEAK / 2), 4), flop_per_launch=flop)
    elif kind == 'valu':
        ach = work / us / 1e3
        row.update(bound='valu', achieved=round(ach, 1), peak=VALU_PAIR_PEAK, unit='Gpairs/s', frac=round(ach / VALU_PAIR_PEAK, 4))
    elif kind == 'fps':
        steps = fps_steps or 4096
        row.update(bound='latency', achieved=round(us / steps, 3), peak=LDS_STEP_IDEAL_US, unit='us/dependent-step',
                   frac=round(LDS_STEP_IDEAL_US / (us / steps), 4))
    return row


def run(batch=8, reps=10, only=None):
    from camliflow_amd.cores import runtime
    from camliflow_amd.csrc import _lib
    _lib.load()
    runtime.set_backend('hip')
    # the rows differentiate with torch.autograd.grad(), which the deferred parameter gradients bypass by design
    # (bench.py switches them on for its step: "One of the differentiated Tensors appears to not have been used")
    deferred = runtime.deferred_param_grads()
    runtime.set_deferred_param_grads(False)
    try:
        return _run(batch, reps, only)
    finally:
        runtime.set_deferred_param_grads(deferred)


KINETO = os.environ.get('CAMLI_KB_KINETO', '1') == '1'      # kernel_us / frac_kernel from a kineto trace of every case


def _train_us(fn, per_graph=20, replays=7):
    """Device time of one call inside a train: `per_graph` back-to-back calls captured in a HIP graph and replayed (median of
    `replays`).  An event pair around ONE launch has a floor of 17-19 us on this stack (marker packets on either side), more than
    the kernels of the small rows take; a replayed train leaves 2-3 us of dispatch gap per call.  Includes every kernel the
    callable launches (an upper bound of the entry point's own time).  None if the callable cannot be captured."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(per_graph):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(replays)]
        for a, b in evs:
            a.record()
            graph.replay()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 / per_graph for a, b in evs)
        return ts[len(ts) // 2]
    except Exception:       # noqa: BLE001 -- a case that allocates / synchronises under capture just keeps its event figure
        torch.cuda.synchronize()
        return None


def _kernel_us(fn, reps):
    """entry point -> mean DEVICE time of the kernels one call of it launches, from a kineto trace of `reps` calls of the case:
    the timer's event pair brackets every entry point on the CPU timeline (hipEventRecord ... hipLaunchKernel ... hipEventRecord),
    the launches in between carry correlation ids, the GPU activities with those ids carry the kernels' own begin / end
    timestamps -- what rocprofv3 reports, without the 14-19 us floor of an event pair.  None when the trace does not pair up
    (some other hipEventRecord in between)."""
    from camliflow_amd.csrc import _lib
    try:
        from torch.profiler import ProfilerActivity, profile
        _lib.TIMER.reset()
        _lib.TIMER.enabled = True
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')         # "Profiler clears events at the end of each cycle": one cycle is all there is
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
        _lib.TIMER.enabled = False
        order = list(_lib.TIMER.order)
        events = prof.profiler.kineto_results.events()
        gpu = {}
        for e in events:
            if 'CUDA' in str(e.device_type()) or 'PrivateUse' in str(e.device_type()):
                gpu[e.correlation_id()] = gpu.get(e.correlation_id(), 0.0) + e.duration_ns() * 1e-3
        cpu = sorted((e for e in events if 'CPU' in str(e.device_type()) and
                      (e.name() == 'hipEventRecord' or 'Launch' in e.name())), key=lambda e: e.start_ns())
        if sum(e.name() == 'hipEventRecord' for e in cpu) != 2 * len(order):
            return None
        out, inside, acc, i = {}, False, 0.0, 0
        for e in cpu:
            if e.name() == 'hipEventRecord':
                if inside:
                    out.setdefault(order[i], []).append(acc)
                    i += 1
                inside, acc = not inside, 0.0
            elif inside:
                acc += gpu.get(e.correlation_id(), 0.0)
        return {name: sum(v) / len(v) for name, v in out.items() if v}
    except Exception:       # noqa: BLE001 -- no profiler, no figure
        _lib.TIMER.enabled = False
        return None


def _with_kernel(row, kernel_us):
    """the row's rate on the kernels' own duration, next to the event figure"""
    if kernel_us is None or kernel_us <= 0:
        return row
    scale = row['avg_launch_us'] / kernel_us
    if kernel_us < 1.0 or (row.get('bound') != 'latency' and row['frac'] * scale > 1.0):
        # the kineto trace did not see this entry point's kernels (under rocprofv3 it sees none and reads ~0.6 us for everything:
        # the r06c / r06g rows, taken by collect_profiles.sh's trace stage, carried fractions of 28 x the peak): no figure rather
        # than a wrong one
        return row
    row['kernel_us'] = round(kernel_us, 2)
    row['frac_kernel'] = round(min(row['frac'] * scale, 1.0) if row.get('bound') == 'latency' else row['frac'] * scale, 4)
    return row


def _with_train(row, train_us):
    """the row's rate re-priced on the train figure (kept next to the event figure, never replacing it)"""
    if train_us is None or train_us <= 0 or train_us >= row['avg_launch_us']:
        return row
    scale = row['avg_launch_us'] / train_us
    row['train_us'] = round(train_us, 2)
    if row.get('bound') == 'latency':
        row['frac_train'] = round(min(row['frac'] * scale, 1.0), 4)
    else:
        row['frac_train'] = round(row['frac'] * scale, 4)
    return row


def _run(batch, reps, only):
    from camliflow_amd.csrc import _lib
    rows = []
    for case, fn, kinds in cases(batch):
        if only and not any(o in case for o in only.split('|')):       # '|' separates alternatives
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        _lib.TIMER.reset()
        _lib.TIMER.only = None
        _lib.TIMER.enabled = True
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        _lib.TIMER.enabled = False
        summary = _lib.TIMER.summary()
        # short single-kernel rows: the event pair's floor is most of the figure -> add the replayed-train time
        timed_names = [n for n in summary if n in kinds]
        kernel_us = _kernel_us(fn, min(reps, 3)) if KINETO else None
        train = None
        if len(timed_names) == 1 and summary[timed_names[0]]['launches'] == reps and \
                summary[timed_names[0]]['total_ms'] / reps < 0.06 and not getattr(fn, 'flop_override', None):
            train = _train_us(fn)
        for name, rec in summary.items():
            if name in kinds:
                override = getattr(fn, 'flop_override', {}).get(name)
                if override is not None:
                    # the adjoint follows the lookups' visit marks: count the flop of the K steps it executes, and keep
                    # the dense product of the same shapes next to it
                    dense = rec['flop'] / rec['launches']
                    rec = dict(rec, flop=override() * rec['launches'])
                    row = _row(case, name, kinds[name], rec)
                    row['dense_flop_per_launch'] = dense
                    row['dense_equivalent_tflops'] = round(dense / row['avg_launch_us'] / 1e6, 2)
                    rows.append(_with_kernel(row, (kernel_us or {}).get(name)))
                    continue
                rows.append(_with_kernel(_with_train(_row(case, name, kinds[name], rec), train), (kernel_us or {}).get(name)))
    _lib.TIMER.reset()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default=None)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    rows = run(args.batch, args.reps, args.only)
    print('%-34s %-28s %10s %12s %-18s %6s' % ('case', 'kernel', 'us', 'achieved', 'unit', 'frac'))
    for r in rows:
        print('%-34s %-28s %10.1f %12.1f %-18s %6.3f' % (r['case'], r['kernel'], r['avg_launch_us'], r['achieved'], r['unit'], r['frac']))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()

"""Host-side profile of the bench step (the step is host-bound: ~250 ms of Python / dispatcher time for ~7,400 launches).
cProfile over K steps with the autograd engine kept on the calling thread, so the custom Functions' backward bodies are
seen too.  Run on the GPU box:  python tools/host_profile.py [--steps 2] [--top 45]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--top', type=int, default=45)
    a = ap.parse_args()
    from camliflow_amd.cores import runtime
    from camliflow_amd.csrc import _lib
    _lib.load()
    runtime.set_backend('hip')
    runtime.set_overlap(True)
    runtime.set_deferred_param_grads(True)
    args = argparse.Namespace(model='camliraft', iters=12, config='camliraft')
    torch.manual_seed(0)
    model = bench.build_model(args).cuda().train()
    opt = bench.make_optimizer(model)
    batch = {k: v.cuda() for k, v in bench.synthetic_batch(8, 540, 960, 8192, seed=100).items()}
    for _ in range(2):
        bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bench.train_step(model, opt, batch)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('un-profiled: host enqueue %.1f ms / step, wall %.1f ms / step' % (host / a.steps * 1e3, (time.perf_counter() - t0) / a.steps * 1e3))
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        bench.train_step(model, opt, batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.strip_dirs()
    print('---- by own time (per %d steps) ----' % a.steps)
    st.sort_stats('tottime').print_stats(a.top)
    print('---- by cumulative time ----')
    st.sort_stats('cumulative').print_stats(a.top)


if __name__ == '__main__':
    main()

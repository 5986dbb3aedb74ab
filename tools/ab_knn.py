"""A/B of the KNN kernels on one GPU: lane-per-query (CAMLI_KNN=lane) vs candidates-across-lanes (CAMLI_KNN=xlane).  Every launch is bracketed by HIP events; rows = the KNN shapes of the headline step.

  python tools/ab_knn.py [--batch 8] [--reps 30] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VALU_PAIR_PEAK = 7865.0

SHAPES = [(8192, 4096, 3, 16), (4096, 2048, 3, 16), (2048, 2048, 3, 32), (2048, 2048, 3, 16), (1024, 2048, 3, 16),
          (512, 2048, 3, 16), (256, 2048, 3, 16), (2048, 2048, 3, 3), (2048, 1024, 3, 3), (2048, 8192, 3, 3),
          (2048, 8160, 2, 1), (16384, 4096, 3, 16), (2048, 16384, 3, 3)]
MODES = [('lane', {'CAMLI_KNN': 'lane'}), ('xlane', {'CAMLI_KNN': 'xlane'})]


def timed(fn, reps, per_graph=20):
    """median device time of one call: `per_graph` back-to-back launches captured in a HIP graph and replayed, so the
    Python / ctypes launch path (15-30 us per call, more than most of these kernels) is not in the figure"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(per_graph):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        graph.replay()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 / per_graph for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--json', default=None)
    ap.add_argument('--waves', default=None, help='comma list of CAMLI_KNN_XL_WAVES values to sweep for the xlane mode')
    args = ap.parse_args()
    from camliflow_amd import csrc
    from camliflow_amd.csrc import wrapper
    g = torch.Generator(device='cpu').manual_seed(0)
    rows = []
    for (m, nq, d, k) in SHAPES:
        b = args.batch
        inp = (torch.rand(b, m, d, generator=g) * 10).cuda()
        qry = (torch.rand(b, nq, d, generator=g) * 10).cuda()
        row = {'shape': 'B%d M%d Nq%d D%d k%d' % (b, m, nq, d, k), 'pairs': b * m * nq}
        ref = None
        for name, env in MODES:
            os.environ.update(env)
            out = csrc.k_nearest_neighbor(inp, qry, k)
            if ref is None:
                ref = out
            row[name + '_equal'] = bool(torch.equal(out, ref))
            us = timed(lambda: csrc.k_nearest_neighbor(inp, qry, k), args.reps)
            row[name + '_us'] = round(us, 2)
            row[name + '_frac'] = round(row['pairs'] / us * 1e-3 / VALU_PAIR_PEAK, 4)
        if args.waves:
            os.environ.update(MODES[1][1])
            for wv in args.waves.split(','):
                os.environ['CAMLI_KNN_XL_WAVES'] = wv
                row['xlane_w%s_us' % wv] = round(timed(lambda: csrc.k_nearest_neighbor(inp, qry, k), args.reps), 2)
            os.environ.pop('CAMLI_KNN_XL_WAVES')
        rows.append(row)
        print(json.dumps(row), flush=True)
    # the four nested cross searches of a GRU iteration
    b = args.batch
    inp = (torch.rand(b, 2048, 3, generator=g) * 10).cuda()
    qry = (torch.rand(b, 2048, 3, generator=g) * 10).cuda()
    sizes = (2048, 1024, 512, 256)
    row = {'shape': 'prefixes B%d 2048/1024/512/256 Nq2048 k16' % b, 'pairs': b * 2048 * sum(sizes)}
    ref = None
    for name, env in MODES:
        os.environ.update(env)
        out = wrapper.k_nearest_neighbor_prefixes(inp, qry, sizes, 16)
        if ref is None:
            ref = out
        row[name + '_equal'] = all(bool(torch.equal(a, c)) for a, c in zip(out, ref))
        us = timed(lambda: wrapper.k_nearest_neighbor_prefixes(inp, qry, sizes, 16), args.reps)
        row[name + '_us'] = round(us, 2)
        row[name + '_frac'] = round(row['pairs'] / us * 1e-3 / VALU_PAIR_PEAK, 4)
    rows.append(row)
    print(json.dumps(row), flush=True)
    for k_ in ('CAMLI_KNN',):
        os.environ.pop(k_, None)
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()

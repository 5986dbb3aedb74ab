"""Randomised equivalence check of the bucket-pruned FPS (Hilbert order, rotated wave assignment) against the full-update kernel:
random cloud sizes in (4096, 16384], sample counts, batch sizes and cloud kinds (uniform, clustered, planar, collinear, duplicated,
constant, huge-extent, with ties on a lattice).  The picks must be identical.   python tools/fps_fuzz.py [cases]"""
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cloud(rng, kind, b, n):
    if kind == 'uniform':
        x = rng.random((b, n, 3))
    elif kind == 'clustered':
        centres = rng.random((b, 8, 3)) * 10
        x = centres[:, rng.integers(0, 8, n)] + rng.normal(0, 0.05, (b, n, 3))
    elif kind == 'planar':
        x = rng.random((b, n, 3)); x[..., 2] = 1.0
    elif kind == 'line':
        t = rng.random((b, n, 1)); x = t * np.array([1.0, 2.0, -0.5])
    elif kind == 'dup':
        x = rng.random((b, n, 3)); x[:, rng.integers(0, n, n // 2)] = x[:, rng.integers(0, n, n // 2)]
    elif kind == 'constant':
        x = np.ones((b, n, 3)) * 0.25
    elif kind == 'huge':
        x = rng.random((b, n, 3)) * 1e4 - 5e3
    else:       # lattice: many exactly equal distances
        x = rng.integers(0, 12, (b, n, 3)).astype(np.float64)
    return torch.from_numpy(x.astype(np.float32)).cuda().contiguous()


def main():
    if os.environ.get('CAMLI_FPS') == 'legacy':
        # child: dump the legacy picks for the seeds given
        from camliflow_amd import csrc
        cases = int(sys.argv[1])
        out = []
        rng = np.random.default_rng(7)
        for i in range(cases):
            kind = KINDS[i % len(KINDS)]
            b, n = int(rng.integers(1, 5)), int(rng.integers(4097, 16385))
            s = int(rng.integers(1, n + 1)) if i % 3 else n // 2
            out.append(csrc.furthest_point_sampling(cloud(rng, kind, b, n), s).cpu())
        torch.save(out, '/tmp/fps_fuzz_legacy.pt')
        return
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    env = dict(os.environ, CAMLI_FPS='legacy')
    subprocess.run([sys.executable, os.path.abspath(__file__), str(cases)], check=True, env=env)
    want = torch.load('/tmp/fps_fuzz_legacy.pt')
    from camliflow_amd import csrc
    rng = np.random.default_rng(7)
    bad = 0
    for i in range(cases):
        kind = KINDS[i % len(KINDS)]
        b, n = int(rng.integers(1, 5)), int(rng.integers(4097, 16385))
        s = int(rng.integers(1, n + 1)) if i % 3 else n // 2
        got = csrc.furthest_point_sampling(cloud(rng, kind, b, n), s).cpu()
        ok = torch.equal(got, want[i])
        bad += not ok
        print('%-10s B%d N%5d -> %5d  %s' % (kind, b, n, s, 'equal' if ok else 'DIFFERENT'))
    print('%d of %d cases differ' % (bad, cases))
    sys.exit(1 if bad else 0)


KINDS = ['uniform', 'clustered', 'planar', 'line', 'dup', 'constant', 'huge', 'lattice']

if __name__ == '__main__':
    main()

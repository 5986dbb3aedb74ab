"""camli_fps variants on the clouds of the headline step (16 clouds of 8192 -> 4096, IDS-transformed frustum points as
bench.synthetic_batch makes them) and of the KITTI shape (2 x 16384 -> 8192): picks compared with the legacy kernel,
time per launch by HIP events.  The variant is read once per process (CAMLI_FPS), so each one runs in a child process.
python tools/ab_fps.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import bench
    from camliflow_amd import csrc
    from camliflow_amd.cores.camliraft import _camera_pair
    from camliflow_amd.cores.geometry import persp2paral
    out = {}
    for name, (b, h, w, n, ns, kitti) in {'things 16x8192->4096': (8, 540, 960, 8192, 4096, False),
                                          'kitti 2x16384->8192': (1, 375, 1242, 16384, 8192, True),
                                          'selfcheck 64x4096->1024': (32, 540, 960, 4096, 1024, False)}.items():
        batch = bench.synthetic_batch(b, h, w, n, seed=7, kitti=kitti)
        persp, paral = _camera_pair((h + 7) // 8 * 8, (w + 7) // 8 * 8, batch['intrinsics'])
        pcs = torch.cat([persp2paral(batch['pcs'][:, :3], persp, paral), persp2paral(batch['pcs'][:, 3:], persp, paral)], 0)
        xyz = pcs.transpose(1, 2).contiguous().cuda()
        for _ in range(2):
            idx = csrc.furthest_point_sampling(xyz, ns)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            idx = csrc.furthest_point_sampling(xyz, ns)
        e.record()
        torch.cuda.synchronize()
        out[name] = (s.elapsed_time(e) / 5, idx.cpu())
    torch.save(out, sys.argv[2])


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
        sys.exit(0)
    import torch
    results = {}
    for variant in os.environ.get('AB_FPS_VARIANTS', 'legacy,p16,p32').split(','):
        path = '/tmp/ab_fps_%s.pt' % variant
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child', path], check=True,
                       env=dict(os.environ, CAMLI_FPS=variant))
        results[variant] = torch.load(path)
    for name in results['legacy']:
        base_ms, base_idx = results['legacy'][name]
        for variant, res in results.items():
            ms, idx = res[name]
            print('%-26s %-8s %8.3f ms  picks equal to legacy: %s' % (name, variant, ms, bool(torch.equal(idx, base_idx))))

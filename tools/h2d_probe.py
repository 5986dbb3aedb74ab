"""Host-to-device time of one configs[2] batch (what a dataloader hands over: images [8,6,540,960] fp32 = 99.5 MB, two clouds, intrinsics,
both targets), pinned and pageable, for the PCIe-inclusive rate DESIGN.md quotes next to the HBM-resident `value` of bench.py.
    python tools/h2d_probe.py [step_ms]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

step_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 174.2
batch = bench.synthetic_batch(8, 540, 960, 8192, seed=100)
nbytes = sum(v.numel() * v.element_size() for v in batch.values())
for name, pin in (('pinned', True), ('pageable', False)):
    host = {k: (v.pin_memory() if pin else v) for k, v in batch.items()}
    for _ in range(3):
        dev = {k: v.to('cuda', non_blocking=True) for k, v in host.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        dev = {k: v.to('cuda', non_blocking=True) for k, v in host.items()}
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print('%-8s %.1f MB per batch: %.2f ms per batch = %.1f GB/s;  step %.1f ms + copy (not overlapped) = %.2f frame-pairs/s (HBM-resident: %.2f)'
          % (name, nbytes / 1e6, ms, nbytes / ms / 1e6, step_ms, 8e3 / (step_ms + ms), 8e3 / step_ms))

#!/usr/bin/env python3
"""us per 1024-instance call for several config-2 batches (seeds) with the library named by LA3D_LIB."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from labelany3d_amd import InstanceFitter
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = []
for seed in (1234, 3, 4, 5, 6, 7):
    depth, masks, K, _, _ = bench.make_inputs(B, dev, seed)
    f = InstanceFitter(B, bench.H, bench.W, dev)
    for _ in range(10):
        f.run(depth, masks, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(100):
            f.run(depth, masks, K)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    out.append(best)
print(os.environ.get("LA3D_LIB", "default"), f"B={B}", " ".join(f"{t:6.1f}" for t in out), f"| mean {np.mean(out):.1f}")

#!/usr/bin/env python3
"""Where the driver-style 20-step figure loses its ~5 us per step against the 1000-step loop: wall clock and HIP-event time of K
back-to-back config-2 steps entered from an idle, synchronised stream, for several K (fit: wall = a + b K)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from labelany3d_amd import InstanceFitter
dev = torch.device("cuda", 0)
B = 1024
depth, masks, K, _, _ = bench.make_inputs(B, dev, 1234)
f = InstanceFitter(B, bench.H, bench.W, dev)
for _ in range(50): f.run(depth, masks, K)
torch.cuda.synchronize()
rows = []
for Ksteps in (1, 5, 20, 40, 100, 400):
    w, e = [], []
    for rep in range(7):
        for _ in range(5): f.run(depth, masks, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(Ksteps): f.run(depth, masks, K)
        e1.record(); t_issue = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        w.append((t1 - t0) * 1e6); e.append(e0.elapsed_time(e1) * 1e3)
    rows.append((Ksteps, np.median(w), np.median(e), (t_issue - t0) * 1e6))
    print(f"K={Ksteps:4d}: wall {np.median(w):9.1f} us ({np.median(w)/Ksteps:7.2f} per step)  events {np.median(e):9.1f} us ({np.median(e)/Ksteps:7.2f} per step)  issue {(t_issue-t0)*1e6:8.1f} us")
ks = np.array([r[0] for r in rows[2:]], float); ws = np.array([r[1] for r in rows[2:]])
b, a = np.polyfit(ks, ws, 1)
print(f"fit over K >= 20: wall = {a:.1f} us + {b:.2f} us x K")

#!/usr/bin/env python3
"""us per 1024-instance call of la3d_masked_ratio_median on the bench_aux workload (config-2 rectangles & an 80 % random mask)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import labelany3d_amd as la
dev = torch.device("cuda", 0)
B = 1024
depth, masks, K, _, _ = bench.make_inputs(B, dev, 1234)
g0 = torch.Generator(device=dev); g0.manual_seed(4)
den = torch.rand((B, bench.H, bench.W), device=dev, generator=g0) * 2.8 + 0.2
g = torch.Generator(device=dev); g.manual_seed(5)
mb = (torch.rand((B, bench.H, bench.W), device=dev, generator=g) < 0.8).to(torch.uint8)
med = cnt = None
for _ in range(3):
    med, cnt = la.masked_ratio_median(depth, den, masks, mb)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        la.masked_ratio_median(depth, den, masks, mb)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
import hashlib
h = hashlib.sha1(med.cpu().numpy().tobytes() + cnt.cpu().numpy().tobytes()).hexdigest()[:10]
print(f"{os.environ.get('LA3D_LIB', 'default'):40s} {best:8.1f} us  [{h}]")

#!/usr/bin/env python3
"""la3d_fit_instances inside a HIP graph: capture one call (torch.cuda.CUDAGraph on a side stream), replay it, compare
the records bit for bit with the eager call and time both.   python profiles/graph_replay.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from labelany3d_amd import InstanceFitter  # noqa: E402

H, W = 480, 640
dev = torch.device("cuda", 0)
K = torch.tensor([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], dtype=torch.float64, device=dev)


def one(B):
    rs = np.random.RandomState(1234)
    depth = torch.rand((B, H, W), device=dev) * 9.5 + 0.5
    masks = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    for i in range(B):
        h, w = rs.randint(8, 301), rs.randint(8, 331)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = 1
    f = InstanceFitter(B, H, W, dev)
    b0, s0, a0 = f.run(depth, masks, K)
    torch.cuda.synchronize()
    b0, s0 = b0.clone(), s0.clone()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(3):
            f.run(depth, masks, K, stream=side)
        side.synchronize()
        f.boxes.zero_()
        with torch.cuda.graph(g, stream=side):
            f.run(depth, masks, K, stream=torch.cuda.current_stream())
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(f.boxes[0], b0) and torch.equal(f.status[0], s0)

    def timed(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    te = timed(lambda: f.run(depth, masks, K))
    tg = timed(g.replay)
    print(f"B={B:5d}  eager {te:7.1f} us/call   graph replay {tg:7.1f} us/call   records identical: {same}")


for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64, 256, 1024]:
    one(B)

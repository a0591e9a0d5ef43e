#!/usr/bin/env python3
"""The octile pattern found by order_search2.py at B = 1024, applied by position fraction to other batch sizes, against the
LPT table (r02 order) - config-2 and config-5 size mixes.    LA3D_LIB=build/abl/libla3d_dbg.so python profiles/r03/order_bsweep.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from labelany3d_amd import InstanceFitter  # noqa: E402
from labelany3d_amd._lib import lib  # noqa: E402

os.environ.setdefault("LA3D_STAGGER_US", "12")
dev = torch.device("cuda", 0)
lib.la3d_debug_set_block_order.argtypes = [C.c_void_p, C.c_int]
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
PI = (2, 6, 5, 0, 7, 1, 3, 4)


def run(B, config5, seed):
    depth, masks, K, _, _ = (bench.make_config5 if config5 else bench.make_inputs)(B, dev, seed)
    t = torch.nn.functional.max_pool2d(masks.float().view(B, 1, bench.H, bench.W), (8, 32)).view(B, -1).sum(1).cpu().numpy()
    rank_to_inst = np.argsort(-t, kind="stable")
    fit = InstanceFitter(B, bench.H, bench.W, dev)
    perm_dev = torch.zeros(B, dtype=torch.int32, device=dev)

    def time(rob, iters=40, reps=2):
        if rob is None:
            lib.la3d_debug_set_block_order(None, 0)
        else:
            assert sorted(rob.tolist()) == list(range(B))
            perm_dev.copy_(torch.as_tensor(rank_to_inst[rob].astype(np.int32)))
            lib.la3d_debug_set_block_order(C.c_void_p(perm_dev.data_ptr()), B)
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fit.run(depth, masks, K, stream=st)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        return best

    r = np.arange(B)
    R = min(B, 512)
    lpt = np.empty(B, np.int64)
    for rk in range(B):
        if rk < R:
            g, pos = rk >> 8, rk & 255
            ng = min(256, R - (g << 8))
            lpt[(g << 8) + (ng - 1 - pos if g >= 1 else pos)] = rk
        else:
            lpt[rk] = rk
    edges = [(k * B) // 8 for k in range(9)]
    octs = [r[edges[k]:edges[k + 1]] for k in range(8)]
    # block groups take the rank groups PI[k]; group sizes differ by at most one, so walk the blocks in order
    pat = np.concatenate([octs[PI[k]] for k in range(8)])
    # "spread" instead of "balance": the second static group in the SAME direction as the first, so that the round-1 finish
    # times of the CUs are spread out and the dynamically placed rest starts staggered
    spread = lpt.copy()
    if B > R:
        spread[256:R] = np.arange(256, R)
    time(None, 10, 1)
    return time(None), time(lpt), time(pat), time(spread)


print(f"{'B':>6s} {'mix':>4s} {'seed':>5s} {'library':>9s} {'LPT table':>10s} {'octile table':>13s} {'spread table':>13s}")
for B in (640, 768, 896, 1024, 1152, 1280):
    for c5 in (False, True):
        for seed in (1234, 3):
            a, b, c, d = run(B, c5, seed)
            print(f"{B:6d} {'c5' if c5 else 'c2':>4s} {seed:5d} {a:9.1f} {b:10.1f} {c:13.1f} {d:13.1f}", flush=True)

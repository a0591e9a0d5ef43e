#!/usr/bin/env python3
"""Where do the 3.6 us between the in-kernel order (order_select) and the same-looking order read from a table go?
Feeds the debug table with (a) exact global ranks, (b) the library's own chunk-local ranks of its ESTIMATED keys (read back
from the workspace), (c) chunk-local ranks of the exact sizes, (d) global ranks of the estimated keys.
    LA3D_LIB=build/abl/libla3d_dbg.so python profiles/r03/order_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from labelany3d_amd import InstanceFitter  # noqa: E402
from labelany3d_amd._lib import lib  # noqa: E402

B = 1024
dev = torch.device("cuda", 0)
depth, masks, K, _, _ = bench.make_inputs(B, dev, 1234)
tiles = torch.nn.functional.max_pool2d(masks.float().view(B, 1, bench.H, bench.W), (8, 32)).view(B, -1).sum(1).cpu().numpy()
fit = InstanceFitter(B, bench.H, bench.W, dev)
st = torch.cuda.current_stream()
lib.la3d_debug_set_block_order.argtypes = [C.c_void_p, C.c_int]
perm_dev = torch.zeros(B, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def measure(perm=None, iters=60, reps=3):
    if perm is None:
        lib.la3d_debug_set_block_order(None, 0)
    else:
        assert sorted(np.asarray(perm).tolist()) == list(range(B))
        perm_dev.copy_(torch.as_tensor(np.asarray(perm).astype(np.int32)))
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


def blk_of_rank(r, R=512):
    R = min(R, B)
    if r < R:
        g, pos = r >> 8, r & 255
        ng = min(256, R - (g << 8))
        return (g << 8) + (ng - 1 - pos if g >= 1 else pos)
    return r


def perm_from_keys(key, chunk=None):
    """key: larger = earlier.  chunk=None: global ranking; else chunk-local ranks merged round-robin like order_select."""
    perm = np.empty(B, np.int64)
    if chunk is None:
        for r, i in enumerate(np.argsort(-key, kind="stable")):
            perm[blk_of_rank(r)] = i
        return perm
    nch = -(-B // chunk)
    per, rem = divmod(B, nch)
    start = 0
    for c in range(nch):
        size = per + (1 if c < rem else 0)
        idx = np.arange(start, start + size)
        start += size
        for lr, i in enumerate(idx[np.argsort(-key[idx], kind="stable")]):
            perm[blk_of_rank(lr * nch + c)] = i
    return perm


for _ in range(3):
    measure(None, 20, 1)
print(f"library (estimate kernel + order_select in the fit kernel): {measure(None):.1f} us")
keys = fit.workspace[0][: 4 * B].view(torch.int32).cpu().numpy().astype(np.int64)
est = (keys >> 14).astype(float) + (16383 - (keys & 16383)) * 0          # estimated area (quantised)
ukey = keys.astype(float)                                                 # the library's unique keys (area, then index)
print(f"corr(estimated area, active tiles) = {np.corrcoef(est, tiles)[0, 1]:.3f}")
print(f"table, exact sizes, global ranking            : {measure(perm_from_keys(tiles)):.1f} us")
print(f"table, exact sizes, chunks of 64              : {measure(perm_from_keys(tiles, 64)):.1f} us")
print(f"table, estimated keys, global ranking         : {measure(perm_from_keys(ukey)):.1f} us")
print(f"table, estimated keys, chunks of 64 (= library): {measure(perm_from_keys(ukey, 64)):.1f} us")
print(f"library again                                  : {measure(None):.1f} us")

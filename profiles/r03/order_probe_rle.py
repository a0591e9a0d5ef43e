#!/usr/bin/env python3
"""Run-length input (compute-bound: no mask stream): how much does the chunk-local order cost against an exact global ranking?
Debug-table build (-DLA3D_DEBUG_ORDER).   LA3D_LIB=build/abl/libla3d_dbg.so python profiles/r03/order_probe_rle.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from labelany3d_amd import InstanceFitter  # noqa: E402
from labelany3d_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda", 0)
lib.la3d_debug_set_block_order.argtypes = [C.c_void_p, C.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream()


def blk_of_rank(r, B, R=1024):
    R = min(R, B)
    if r < R:
        g, pos = r >> 8, r & 255
        ng = min(256, R - (g << 8))
        return (g << 8) + (ng - 1 - pos if g >= 1 else pos)
    return r


def perm_from(key, B, chunk=None):
    perm = np.empty(B, np.int64)
    if chunk is None:
        for r, i in enumerate(np.argsort(-key, kind="stable")):
            perm[blk_of_rank(r, B)] = i
        return perm
    nch = -(-B // chunk)
    per, rem = divmod(B, nch)
    start = 0
    for c in range(nch):
        size = per + (1 if c < rem else 0)
        idx = np.arange(start, start + size)
        start += size
        for lr, i in enumerate(idx[np.argsort(-key[idx], kind="stable")]):
            perm[blk_of_rank(lr * nch + c, B)] = i
    return perm


for B in (512, 1024, 2048):
    for seed in (1234, 3):
        lib.la3d_debug_set_block_order(None, 0)     # the table of the previous batch is about to be freed
        depth, masks, K, _, rects = bench.make_inputs(B, dev, seed)
        rc_np, ro_np = bench.rect_rle(rects)
        rle_c, rle_o = torch.as_tensor(rc_np, device=dev), torch.as_tensor(ro_np, device=dev)
        kfull = K[None].expand(B, 3, 3).contiguous()
        f = InstanceFitter(B, bench.H, bench.W, dev)
        perm_dev = torch.zeros(B, dtype=torch.int32, device=dev)
        tiles = torch.nn.functional.max_pool2d(masks.float().view(B, 1, bench.H, bench.W), (8, 32)).view(B, -1).sum(1).cpu().numpy()
        area = masks.reshape(B, -1).sum(1).cpu().numpy().astype(float)

        def run():
            check(lib.la3d_fit_instances_rle(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(rle_c.data_ptr()),
                                             C.c_void_p(rle_o.data_ptr()), C.c_void_p(kfull.data_ptr()), 9, None, None, B, bench.H, bench.W,
                                             C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()),
                                             C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()),
                                             C.c_void_p(st.cuda_stream)), "rle")

        def measure(perm):
            if perm is None:
                lib.la3d_debug_set_block_order(None, 0)
            else:
                perm_dev.copy_(torch.as_tensor(perm.astype(np.int32)))
                lib.la3d_debug_set_block_order(C.c_void_p(perm_dev.data_ptr()), B)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); e0.record()
                for _ in range(60):
                    run()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 60 * 1e3)
            return best
        for _ in range(5):
            run()
        print(f"B={B} seed={seed}: library {measure(None):6.1f} | table: area/chunk64 {measure(perm_from(area, B, 64)):6.1f}  area/global {measure(perm_from(area, B)):6.1f}  "
              f"tiles/global {measure(perm_from(tiles, B)):6.1f}  area/chunk256 {measure(perm_from(area, B, 256)):6.1f}", flush=True)

#!/usr/bin/env python3
"""Structured (size-aware, few-parameter) launch orders measured through the debug table, on config-2 and config-5 batches.
    LA3D_LIB=build/abl/libla3d_dbg.so python profiles/r03/order_structured.py"""
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
B = 1024
dev = torch.device("cuda", 0)
lib.la3d_debug_set_block_order.argtypes = [C.c_void_p, C.c_int]
fit = InstanceFitter(B, bench.H, bench.W, dev)
st = torch.cuda.current_stream()
perm_dev = torch.zeros(B, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


class Batch:
    def __init__(self, seed, config5=False):
        self.depth, self.masks, self.K, _, _ = (bench.make_config5 if config5 else bench.make_inputs)(B, dev, seed)
        t = torch.nn.functional.max_pool2d(self.masks.float().view(B, 1, bench.H, bench.W), (8, 32)).view(B, -1).sum(1).cpu().numpy()
        self.rank_to_inst = np.argsort(-t, kind="stable")
        self.name = f"{'c5' if config5 else 'c2'}/s{seed}"

    def time(self, rob, iters=40, reps=2):
        if rob is None:
            lib.la3d_debug_set_block_order(None, 0)
        else:
            assert sorted(rob.tolist()) == list(range(B))
            perm_dev.copy_(torch.as_tensor(self.rank_to_inst[rob].astype(np.int32)))
            lib.la3d_debug_set_block_order(C.c_void_p(perm_dev.data_ptr()), B)
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fit.run(self.depth, self.masks, self.K, stream=st)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        return best


r = np.arange(B)
ev, od = r[0::2], r[1::2]
q = [r[i * 256:(i + 1) * 256] for i in range(4)]
o = [r[i * 128:(i + 1) * 128] for i in range(8)]
cands = {
    "library (in-kernel)": None,
    "LPT table (r02 order)": np.concatenate([q[0], q[1][::-1], q[2], q[3]]),
    "evens LPT | odds desc": np.concatenate([ev[:256], ev[256:][::-1], od]),
    "evens LPT | odds asc": np.concatenate([ev[:256], ev[256:][::-1], od[::-1]]),
    "evens: small on A, big on B(stag) | odds desc": np.concatenate([ev[256:], ev[:256][::-1], od]),
    "evens LPT | odds: small quarter first, then desc": np.concatenate([ev[:256], ev[256:][::-1], od[384:], od[:384]]),
    "thirds: A = r%3==1, B = r%3==0 rev, dyn = rest desc": None,
    "quartile interleave r%4: A=1 B=0rev dyn=2,3": np.concatenate([r[1::4], r[0::4][::-1], r[2::4], r[3::4]]),
    "quartile interleave r%4: A=0 B=2rev dyn=1,3": np.concatenate([r[0::4], r[2::4][::-1], r[1::4], r[3::4]]),
    "quartile interleave r%4: A=0 B=3rev dyn=1,2": np.concatenate([r[0::4], r[3::4][::-1], r[1::4], r[2::4]]),
    "search best octiles (2,6,5,0,7,1,3,4)/10011101": np.concatenate([o[2][::-1], o[6], o[5][::-1], o[0][::-1], o[7][::-1], o[1], o[3], o[4][::-1]]),
    "octiles (2,6,5,0,7,1,3,4) no flips": np.concatenate([o[2], o[6], o[5], o[0], o[7], o[1], o[3], o[4]]),
    "octiles A=(2,6) B=(5,0) dyn desc (1,3,4,7)": np.concatenate([o[2], o[6], o[5], o[0], o[1], o[3], o[4], o[7]]),
    "octiles A=(2,5) B=(6,0)": np.concatenate([o[2], o[5], o[6], o[0], o[7], o[1], o[3], o[4]]),
}
a = r[(r % 3) == 1][:256]
b = r[(r % 3) == 0][:256]
rest = np.setdiff1d(r, np.concatenate([a, b]))
cands["thirds: A = r%3==1, B = r%3==0 rev, dyn = rest desc"] = np.concatenate([a, b[::-1], rest])

batches = [Batch(1234), Batch(3), Batch(4), Batch(1234, True), Batch(6, True)]
for bt in batches:
    bt.time(None, 10, 1)
print(f"{'order':58s}" + "".join(f"{bt.name:>10s}" for bt in batches))
for name, rob in cands.items():
    print(f"{name:58s}" + "".join(f"{bt.time(rob):10.1f}" for bt in batches), flush=True)

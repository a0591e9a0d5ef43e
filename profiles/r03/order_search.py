#!/usr/bin/env python3
"""Empirical search over launch orders of the instance engine (measurement build -DLA3D_DEBUG_ORDER: the block -> instance
table comes from this script instead of order_select).  Objective = measured us per 1024-instance call on the GPU.
Families: (F1) the four size quartiles dealt to the four block groups in any assignment / direction, with a stagger sweep;
(F2) the same on octiles (random sample); (F3) hill climbing by swapping blocks from the best found.
    LA3D_LIB=build/abl/libla3d_dbg.so python profiles/r03/order_search.py [--plain] [--config5] [--seed S]"""
import argparse
import ctypes as C
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from labelany3d_amd import InstanceFitter  # noqa: E402
from labelany3d_amd._lib import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--plain", action="store_true")
ap.add_argument("--config5", action="store_true")
ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--climb", type=int, default=600)
args = ap.parse_args()
if args.plain:
    os.environ["LA3D_RETAIN"] = "0"
B = 1024
dev = torch.device("cuda", 0)
mk = bench.make_config5 if args.config5 else bench.make_inputs
depth, masks, K, _, _ = mk(B, dev, args.seed)
tiles = torch.nn.functional.max_pool2d(masks.float().view(B, 1, bench.H, bench.W), (8, 32)).view(B, -1).sum(1).cpu().numpy()
rank_to_inst = np.argsort(-tiles, kind="stable")          # rank 0 = largest
fit = InstanceFitter(B, bench.H, bench.W, dev)
st = torch.cuda.current_stream()
lib.la3d_debug_set_block_order.argtypes = [C.c_void_p, C.c_int]
perm_dev = torch.zeros(B, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def measure(perm=None, iters=30, reps=2):
    if perm is None:
        lib.la3d_debug_set_block_order(None, 0)
    else:
        assert sorted(perm.tolist()) == list(range(B))
        perm_dev.copy_(torch.as_tensor(perm.astype(np.int32)))
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


def deal(groups_of_ranks):
    """groups_of_ranks[g] = ranks (in the order they should occupy blocks g*len .. ) -> perm[block] = instance"""
    return rank_to_inst[np.concatenate(groups_of_ranks)]


for _ in range(3):
    measure(None)
base = measure(None, 60, 3)
print(f"library order ({'plain' if args.plain else 'retaining'} build): {base:.1f} us", flush=True)
r = np.arange(B)
lib_like = deal([r[0:256], r[256:512][::-1], r[512:768], r[768:1024]])
print(f"same order through the table: {measure(lib_like, 60, 3):.1f} us", flush=True)

results = []
q = [r[i * 256:(i + 1) * 256] for i in range(4)]
t0 = time.time()
for stag in (["0", "6", "12", "18"] if not args.plain else ["0"]):
    os.environ["LA3D_STAGGER_US"] = stag
    for pi in itertools.permutations(range(4)):
        for dirs in range(16):
            g = [q[pi[k]][::-1] if (dirs >> k) & 1 else q[pi[k]] for k in range(4)]
            t = measure(deal(g), 20, 1)
            results.append((t, stag, pi, dirs))
results.sort(key=lambda x: x[0])
print(f"F1: {len(results)} quartile orders in {time.time() - t0:.0f} s; best:")
for t, stag, pi, dirs in results[:12]:
    print(f"   {t:6.1f} us  stagger {stag:>2s}  quartile of block group 0..3 = {pi}  reversed mask {dirs:04b}")
print("   worst:", ", ".join(f"{x[0]:.0f}" for x in results[-3:]))
best_t, stag, pi, dirs = results[0]
os.environ["LA3D_STAGGER_US"] = stag
best_perm = deal([q[pi[k]][::-1] if (dirs >> k) & 1 else q[pi[k]] for k in range(4)])
best_t = measure(best_perm, 60, 3)
print(f"F1 best re-measured: {best_t:.1f} us")

# F2: octiles
rs = np.random.RandomState(0)
o = [r[i * 128:(i + 1) * 128] for i in range(8)]
res2 = []
for _ in range(1500):
    pi8 = rs.permutation(8)
    d8 = rs.randint(0, 256)
    g = [o[pi8[k]][::-1] if (d8 >> k) & 1 else o[pi8[k]] for k in range(8)]
    res2.append((measure(np.asarray(rank_to_inst[np.concatenate(g)]), 20, 1), tuple(pi8), d8))
res2.sort(key=lambda x: x[0])
print("F2 (octiles, 1500 random): best")
for t, pi8, d8 in res2[:8]:
    print(f"   {t:6.1f} us  octile of block group 0..7 = {pi8}  reversed mask {d8:08b}")
t2 = measure(rank_to_inst[np.concatenate([o[res2[0][1][k]][::-1] if (res2[0][2] >> k) & 1 else o[res2[0][1][k]] for k in range(8)])], 60, 3)
print(f"F2 best re-measured: {t2:.1f} us")
if t2 < best_t:
    best_t = t2
    best_perm = rank_to_inst[np.concatenate([o[res2[0][1][k]][::-1] if (res2[0][2] >> k) & 1 else o[res2[0][1][k]] for k in range(8)])]

# F3: hill climbing (swap two blocks; accept when the re-measured time improves by more than the noise)
cur, cur_t = best_perm.copy(), best_t
acc = 0
for it in range(args.climb):
    cand = cur.copy()
    k = rs.randint(1, 9)
    for _ in range(k):
        a, b = rs.randint(0, B, 2)
        cand[a], cand[b] = cand[b], cand[a]
    t = measure(cand, 20, 1)
    if t < cur_t - 0.4:
        t = measure(cand, 40, 2)
        if t < cur_t - 0.2:
            cur, cur_t = cand, t
            acc += 1
print(f"F3: {acc} accepted swaps, {best_t:.1f} -> {cur_t:.1f} us (re-measured {measure(cur, 60, 3):.1f})")
inv_rank = np.empty(B, int); inv_rank[rank_to_inst] = np.arange(B)
rk = inv_rank[cur]
print("rank of the instance on block b, mean per block group of 128:", [int(rk[i * 128:(i + 1) * 128].mean()) for i in range(8)])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", f"order_best_{'plain' if args.plain else 'ret'}.npy"), rk)
print(f"library {base:.1f} us | best found {cur_t:.1f} us")

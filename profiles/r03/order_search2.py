#!/usr/bin/env python3
"""Cross-validated search over octile launch orders (see order_search.py): objective = mean us per call over three batches
with different seeds; the winners are then measured on held-out batches (other seeds, the config-5 size mix).
    LA3D_LIB=build/abl/libla3d_dbg.so python profiles/r03/order_search2.py [--plain] [--groups 8]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from labelany3d_amd import InstanceFitter  # noqa: E402
from labelany3d_amd._lib import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--plain", action="store_true")
ap.add_argument("--groups", type=int, default=8)
ap.add_argument("--samples", type=int, default=2500)
ap.add_argument("--climb", type=int, default=800)
ap.add_argument("--stagger", default="12")
ap.add_argument("--both", action="store_true", help="objective over config-2 AND config-5 batches (normalised)")
args = ap.parse_args()
if args.plain:
    os.environ["LA3D_RETAIN"] = "0"
os.environ["LA3D_STAGGER_US"] = args.stagger
B, G = 1024, args.groups
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
        self.name = f"{'config5' if config5 else 'config2'}/seed{seed}"

    def time(self, rank_of_block, iters=20, reps=1):
        if rank_of_block is None:
            lib.la3d_debug_set_block_order(None, 0)
        else:
            perm_dev.copy_(torch.as_tensor(self.rank_to_inst[rank_of_block].astype(np.int32)))
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


def pattern(pi, flips):
    n = B // G
    return np.concatenate([np.arange(pi[k] * n, (pi[k] + 1) * n)[::-1] if (flips >> k) & 1 else np.arange(pi[k] * n, (pi[k] + 1) * n)
                           for k in range(G)])


train = [Batch(s) for s in (1234, 1, 2)] if not args.both else [Batch(1234), Batch(1), Batch(1234, True), Batch(2, True)]
NORM = [1.0] * len(train) if not args.both else [1.0, 1.0, 98.5 / 83.0, 98.5 / 83.0]   # config-5 times scaled to the config-2 level
for b in train:
    b.time(None)


def score(rob, iters=20, reps=1):
    return float(np.mean([b.time(rob, iters, reps) * n for b, n in zip(train, NORM)]))


r = np.arange(B)
lib_like = np.concatenate([r[0:256], r[256:512][::-1], r[512:]])
print(f"library order: {score(None, 40, 2):.1f} us | its table form (exact global ranks): {score(lib_like, 40, 2):.1f} us", flush=True)
rs = np.random.RandomState(7)
res = []
for _ in range(args.samples):
    pi, fl = tuple(rs.permutation(G)), int(rs.randint(0, 1 << G))
    res.append((score(pattern(pi, fl)), pi, fl))
res.sort(key=lambda x: x[0])
print("random sample, best:")
for t, pi, fl in res[:6]:
    print(f"   {t:6.1f} us  size group of block group 0..{G - 1} = {tuple(int(x) for x in pi)}  reversed {fl:0{G}b}")
cur_t, cur_pi, cur_fl = res[0]
cur_t = score(pattern(cur_pi, cur_fl), 40, 2)
acc = 0
for it in range(args.climb):
    pi, fl = list(cur_pi), cur_fl
    if rs.rand() < 0.6:
        a, b = rs.randint(0, G, 2)
        pi[a], pi[b] = pi[b], pi[a]
    else:
        fl ^= 1 << int(rs.randint(0, G))
    t = score(pattern(pi, fl))
    if t < cur_t - 0.3:
        t = score(pattern(pi, fl), 40, 2)
        if t < cur_t - 0.15:
            cur_t, cur_pi, cur_fl, acc = t, tuple(pi), fl, acc + 1
print(f"hill climb: {acc} moves accepted -> {cur_t:.1f} us: groups {tuple(int(x) for x in cur_pi)} reversed {cur_fl:0{G}b}", flush=True)
cands = [("library", None), ("table/global", lib_like), ("climbed", pattern(cur_pi, cur_fl))] + \
        [(f"sample#{k}", pattern(res[k][1], res[k][2])) for k in range(3)]
print("\nheld-out batches (us per call):")
tests = [Batch(s) for s in (3, 4, 5)] + [Batch(s, True) for s in (1234, 6)]
print(f"{'':14s}" + "".join(f"{b.name:>18s}" for b in tests) + f"{'train mean':>14s}")
for name, rob in cands:
    print(f"{name:14s}" + "".join(f"{b.time(rob, 40, 2):18.1f}" for b in tests) + f"{score(rob, 40, 2):14.1f}", flush=True)

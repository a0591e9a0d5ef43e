#!/usr/bin/env python3
"""What sets the pass time of the instance engine?  Times la3d_fit_instances (instance engine pinned) on batches of
1024 rectangular masks with controlled sizes.  Run on an MI355X: python profiles/exp_chain.py"""
import os
import sys

import numpy as np
import torch

os.environ["LA3D_ENGINE"] = "instance"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from labelany3d_amd import InstanceFitter  # noqa: E402

H, W, B = 480, 640, 1024
dev = torch.device("cuda", 0)
depth = torch.rand((B, H, W), device=dev) * 9.5 + 0.5
K = torch.tensor([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], dtype=torch.float64, device=dev)
f = InstanceFitter(B, H, W, dev)


def run(name, hh, ww):
    masks = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    for i in range(B):
        masks[i, 10:10 + int(hh[i]), 10:10 + int(ww[i])] = 1
    for _ in range(10):
        f.run(depth, masks, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        f.run(depth, masks, K)
    e1.record()
    torch.cuda.synchronize()
    tiles = ((np.asarray(ww) + 10 + 31) // 32 - 10 // 32) * ((np.asarray(hh) + 10 + 7) // 8 - 10 // 8)
    print(f"{name:42s} {e0.elapsed_time(e1) / 100 * 1e3:8.1f} us/step   tiles/instance mean {tiles.mean():6.1f} max {tiles.max()}")


ones = np.ones(B, int)
run("all tiny (8x8)", 8 * ones, 8 * ones)
run("all average (154x169)", 154 * ones, 169 * ones)
run("all huge (300x330)", 300 * ones, 330 * ones)
hh, ww = 8 * ones, 8 * ones
hh[0], ww[0] = 300, 330
run("one huge, 1023 tiny", hh, ww)
hh, ww = 8 * ones, 8 * ones
hh[:256], ww[:256] = 300, 330
run("256 huge (one per CU), 768 tiny", hh, ww)
hh, ww = 154 * ones, 169 * ones
hh[0], ww[0] = 300, 330
run("one huge, 1023 average", hh, ww)
rs = np.random.RandomState(1234)
run("bench distribution", rs.randint(8, 301, B), rs.randint(8, 331, B))

# --- does a size-balanced placement help?  Arrange the SAME multiset of sizes so that, under the measured
# placement (block b -> CU b%256 for the first 1024 blocks; instance = xcd_remap(b)), every CU hosts a snake-
# balanced quadruple.
hh, ww = rs.randint(8, 301, B), rs.randint(8, 331, B)
run("bench-like distribution (fresh draw)", hh, ww)
order = np.argsort(-(hh * ww), kind="stable")
inst_of_block = [(b % 8) * (B // 8) + b // 8 for b in range(B)]
h2, w2 = np.zeros(B, int), np.zeros(B, int)
for r, src in enumerate(order):
    g, pos = divmod(r, 256)
    blk = g * 256 + (255 - pos if g & 1 else pos)
    h2[inst_of_block[blk]], w2[inst_of_block[blk]] = hh[src], ww[src]
run("same sizes, snake-balanced per CU", h2, w2)
h3, w3 = np.zeros(B, int), np.zeros(B, int)
for r, src in enumerate(order):          # adversarial: the 4 largest on CU 0, next 4 on CU 1, ...
    cu, k = divmod(r, 4)
    blk = k * 256 + cu
    h3[inst_of_block[blk]], w3[inst_of_block[blk]] = hh[src], ww[src]
run("same sizes, worst case (sorted per CU)", h3, w3)
h4, w4 = np.zeros(B, int), np.zeros(B, int)
for r, src in enumerate(order):          # sorted descending in block order
    h4[inst_of_block[r]], w4[inst_of_block[r]] = hh[src], ww[src]
run("same sizes, descending block order", h4, w4)

# --- the in-library launch order (LA3D_BALANCE, default on): read est/perm back from the workspace
masks = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
for i in range(B):
    masks[i, 10:10 + int(hh[i]), 10:10 + int(ww[i])] = 1
f.run(depth, masks, K)
torch.cuda.synchronize()
wsi = f.workspace[0][: 8 * B].view(torch.int32).cpu().numpy()
est, perm = wsi[:B].astype(np.int64) >> 14, wsi[B:]
area = hh * ww
print("perm is a permutation:", sorted(perm.tolist()) == list(range(B)),
      " corr(est, area) = %.4f" % np.corrcoef(est, area)[0, 1])
cu = np.zeros(256)
for b in range(B):
    cu[b % 256] += area[perm[b]]
print("per-CU area  max/mean with library order: %.3f" % (cu.max() / cu.mean()))
cu = np.zeros(256)
for b in range(B):
    cu[b % 256] += area[inst_of_block[b]]
print("per-CU area  max/mean with plain order:   %.3f" % (cu.max() / cu.mean()))

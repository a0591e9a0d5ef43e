#!/usr/bin/env python3
"""Tiles pass B walks with / without culling (a -DLA3D_CULL_STATS build): per workload, by instance size."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from labelany3d_amd import InstanceFitter

dev = torch.device("cuda", 0)
for name, mk in (("config2", bench.make_inputs), ("config5", bench.make_config5)):
    B = 1024
    depth, masks, K, n_masked, rects = mk(B, dev, 1234)
    f = InstanceFitter(B, bench.H, bench.W, dev)
    f.workspace.zero_()
    f.run(depth, masks, K, build="plain")
    torch.cuda.synchronize()
    st = f.workspace[0].view(torch.int32)[B:3 * B].view(B, 2).cpu().numpy()
    na, ns = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
    print(f"{name}: active tiles {int(na.sum())}, pass-B tiles {int(ns.sum())} = {ns.sum() / na.sum():.3f}")
    for lo, hi in ((0, 96), (96, 160), (160, 256), (256, 2000)):
        sel = (na >= lo) & (na < hi)
        if sel.any():
            print(f"   {lo:4d} <= tiles < {hi:4d}: {int(sel.sum()):4d} instances, {int(na[sel].sum()):7d} tiles, pass B walks {ns[sel].sum() / na[sel].sum():.3f}")
# smooth depth (sphere-like objects: the nearest point lies inside the mask)
B = 256
depth, masks, K, n_masked, rects = bench.make_config5(B, dev, 7)
vv, uu = torch.meshgrid(torch.arange(bench.H, device=dev, dtype=torch.float32), torch.arange(bench.W, device=dev, dtype=torch.float32), indexing="ij")
r0, c0, hh, ww = (torch.as_tensor(x, device=dev, dtype=torch.float32).view(-1, 1, 1) for x in rects)
rr = torch.clamp(1 - ((vv - r0 - hh / 2) / (hh / 1.5)) ** 2 - ((uu - c0 - ww / 2) / (ww / 1.5)) ** 2, 0, 1)
depth = (5.0 - 1.5 * torch.sqrt(rr) + 0.003 * torch.randn_like(rr)).contiguous()
f = InstanceFitter(B, bench.H, bench.W, dev)
f.workspace.zero_()
f.run(depth, masks, K, build="plain")
torch.cuda.synchronize()
st = f.workspace[0].view(torch.int32)[B:3 * B].view(B, 2).cpu().numpy()
na, ns = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
big = na >= 96
print(f"smooth (sphere-like) depth, config-5 masks: all {ns.sum() / na.sum():.3f}; instances >= 96 tiles: {ns[big].sum() / max(na[big].sum(), 1):.3f}")

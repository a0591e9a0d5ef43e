#!/usr/bin/env python3
"""Which records differ between the plain build (pass-B culling) and the retaining build (every tile walked)?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from labelany3d_amd import InstanceFitter

dev = torch.device("cuda", 0)
for B in (512, 1024):
    depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
    f = InstanceFitter(B, bench.H, bench.W, dev, slots=2)
    b0, s0, a0 = f.run(depth, masks, K, slot=0, build="plain")
    b1, s1, a1 = f.run(depth, masks, K, slot=1, build="retaining")
    torch.cuda.synchronize()
    b0, b1 = b0.cpu().numpy(), b1.cpu().numpy()
    bad = np.nonzero((b0 != b1).any(1))[0]
    print(f"B={B}: {len(bad)} records differ; status equal: {bool((s0 == s1).all())}")
    r0, c0, hh, ww = rects
    for i in bad[:12]:
        cols = np.nonzero(b0[i] != b1[i])[0]
        ty0, ty1 = r0[i] // 8, (r0[i] + hh[i] - 1) // 8
        tx0, tx1 = c0[i] // 32, (c0[i] + ww[i] - 1) // 32
        print(f"  inst {i}: rect r0={r0[i]} c0={c0[i]} h={hh[i]} w={ww[i]} tiles={(ty1-ty0+1)*(tx1-tx0+1)} cols={cols.tolist()[:8]} "
              f"dims plain={b0[i,3:6]} ret={b1[i,3:6]} max|diff|={np.abs(b0[i]-b1[i]).max():.3e}")
        # brute force extents for this instance
        d = depth[i].cpu().numpy().astype(np.float64)
        m = masks[i].cpu().numpy().astype(bool)
        vs, us = np.nonzero(m)
        Kinv = np.linalg.inv(np.array(bench.K640))
        P = (Kinv @ np.stack([us, vs, np.ones_like(us)]).astype(np.float64)) * d[vs, us]
        yaw = a1[i, 0].item()
        c, s = np.cos(yaw), np.sin(yaw)
        xr = c * P[0] + s * P[2]; zr = -s * P[0] + c * P[2]
        print(f"     brute dims [dz,dy,dx] = {[zr.max()-zr.min(), P[1].max()-P[1].min(), xr.max()-xr.min()]}")
        for name, q in (("x", xr), ("y", P[1]), ("z", zr)):
            for which, j in (("min", q.argmin()), ("max", q.argmax())):
                print(f"       {name}{which} at px (v={vs[j]}, u={us[j]}) tile (ty={vs[j]//8}, tx={us[j]//32}) d={d[vs[j], us[j]]:.6f}")

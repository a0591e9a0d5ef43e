"""Per-wave start / end times of the two passes (measurement build with per-wave stamps: profiles/experiments/r03_wave_stamps.patch
on top of -DLA3D_TIMELINE; results in profiles/r03/r03_wave_skew.txt)."""
import os, sys
os.environ["LA3D_ENGINE"] = "instance"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from labelany3d_amd import InstanceFitter
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 1024]:
    depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
    f = InstanceFitter(B, bench.H, bench.W, dev)
    for _ in range(5):
        f.run(depth, masks, K)
    torch.cuda.synchronize()
    ws = f.workspace[0]
    tl = ws[8192: 8192 + B * 128].view(torch.float64).cpu().numpy().reshape(B, 16)
    wv = ws[8192 + B * 128: 8192 + B * 128 + B * 256].view(torch.float64).cpu().numpy().reshape(B, 32)
    t0 = tl[:, 8].min()
    a0 = (wv[:, 8:16] - t0) / 100.0
    a1 = (wv[:, 0:8] - t0) / 100.0
    b1 = (wv[:, 16:24] - t0) / 100.0
    r0, c0, hh, ww = rects
    big = np.argsort(-(np.asarray(hh) * np.asarray(ww)))[:3]
    print(f"== B={B}: pass A duration per wave (end - start), mean over instances: {np.round((a1 - a0).mean(0), 2)}")
    print(f"   pass A end skew (max - min over waves): mean {np.mean(a1.max(1) - a1.min(1)):.2f} us; wave that ends last (histogram): {np.bincount(a1.argmax(1), minlength=8)}")
    print(f"   pass B end skew: mean {np.mean(b1.max(1) - b1.min(1)):.2f} us; last wave histogram: {np.bincount(b1.argmax(1), minlength=8)}")
    for i in big:
        print(f"   instance {i} ({hh[i]}x{ww[i]} px): pass A start {np.round(a0[i], 1)} end {np.round(a1[i], 1)} | pass B end {np.round(b1[i], 1)}")

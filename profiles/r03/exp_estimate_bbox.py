"""Latency of the scalar drop-in estimate_bbox (reference src/util_3dbox.py:106-178: one (N,3) cloud per call, NumPy in / out)."""
import io, os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from labelany3d_amd import util_3dbox as U
from oracle import la3d_oracle as O
rs = np.random.RandomState(3)
for n in (500, 5000, 100000):
    pc = rs.randn(n, 3) * [2.0, 0.5, 1.0] + [0, 0, 6]
    g = np.array([0.02, -0.97, 0.1, 1.0])
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(10):
            U.estimate_bbox(pc, "x", g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            U.estimate_bbox(pc, "x", g)
        t = (time.perf_counter() - t0) / 200 * 1e6
        t0 = time.perf_counter()
        for _ in range(20):
            O.estimate_bbox(pc, "x", g)
        tr = (time.perf_counter() - t0) / 20 * 1e6
    print(f"N={n:6d}: drop-in estimate_bbox {t:7.1f} us per call | NumPy oracle {tr:8.1f} us")

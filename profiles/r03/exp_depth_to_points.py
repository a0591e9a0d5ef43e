"""Latency of the drop-in depth_to_points (reference src/util.py:52-75) for one 480x640 frame: NumPy in / NumPy out vs tensors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from labelany3d_amd.util import depth_to_points
from oracle import la3d_oracle as O
K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
d = np.random.RandomState(0).uniform(0.5, 10, (1, 480, 640)).astype(np.float32)
dt = torch.as_tensor(d, device="cuda")
def T(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"NumPy in -> NumPy out: {T(lambda: depth_to_points(d, K)):8.1f} us   (7.4 MB of float64 points cross PCIe)")
print(f"tensor in -> tensor out: {T(lambda: depth_to_points(dt, K)):8.1f} us")
print(f"NumPy oracle (reference arithmetic): {T(lambda: O.depth_to_points(d, K), 5):8.1f} us")

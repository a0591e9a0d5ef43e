import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
dev = torch.device("cuda", 0)
H, W, B = 480, 640, 8
rs = np.random.RandomState(5)
depth = torch.rand((H, W), device=dev) * 9 + 0.5
K = np.array([[500.0, 0, W / 2], [0, 500.0, H / 2], [0, 0, 1]])
def blob(n):
    ang = np.sort(rs.uniform(0, 2 * np.pi, n)); cx, cy = rs.uniform(0.3 * W, 0.7 * W), rs.uniform(0.3 * H, 0.7 * H); rad = rs.uniform(0.5, 1.0, n)
    return np.stack([cx + 0.25 * W * rad * np.cos(ang), cy + 0.3 * H * rad * np.sin(ang)], 1).round().ravel().tolist()
segs = [[blob(40)] for _ in range(B)]
polys = la.pack_polygons(segs, H, W)
dpolys = tuple(torch.as_tensor(x, device=dev) if isinstance(x, np.ndarray) else x for x in polys)
for name, f in (("host arrays", lambda: la.fit_instances_poly(depth, polys, K)), ("device arrays", lambda: la.fit_instances_poly(depth, dpolys, K))):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): f()
    torch.cuda.synchronize()
    print(name, "per call us:", round((time.perf_counter() - t0) / 500 * 1e6, 1))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): la.fit_instances_poly(depth, polys, K)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)

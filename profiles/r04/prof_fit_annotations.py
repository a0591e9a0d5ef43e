"""cProfile of fit_annotations on the reference's per-image pattern (8 polygon annotations of 40 vertices): which Python lines cost what."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
dev = torch.device("cuda", 0)
H, W = 480, 640
rs = np.random.RandomState(5)
K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
depth = torch.as_tensor(rs.uniform(0.5, 10, (H, W)).astype(np.float32), device=dev)
def blob(n):
    ang = np.sort(rs.uniform(0, 2 * np.pi, n)); cx, cy = rs.uniform(0.3 * W, 0.7 * W), rs.uniform(0.3 * H, 0.7 * H); rad = rs.uniform(0.5, 1.0, n)
    return np.stack([cx + 0.25 * W * rad * np.cos(ang), cy + 0.3 * H * rad * np.sin(ang)], 1).round().ravel().tolist()
anns = [{"iscrowd": 0, "bbox": [0.0, 0.0, 1.0, 1.0], "category_id": 1 + i % 5, "segmentation": [blob(40)], "area": 20000.0} for i in range(8)]
ground = np.array([[0.02, -0.97, 0.1, 1.0]] * 8)
f = lambda: la.fit_annotations(anns, (W, H), depth, K, ground=ground, to_host=True)
for _ in range(20): f()
pr = cProfile.Profile(); pr.enable()
for _ in range(500): f()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])

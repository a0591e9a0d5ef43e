import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import labelany3d_amd as la
dev = torch.device("cuda", 0)
H, W, B = 480, 640, 8
depth = torch.rand((H, W), device=dev) * 9 + 0.5
K = torch.tensor([[500.0, 0, W / 2], [0, 500.0, H / 2], [0, 0, 1]], dtype=torch.float64, device=dev)
masks = torch.zeros((B, H, W), dtype=torch.uint8, device=dev); masks[:, 100:300, 200:400] = 1
ii = torch.zeros(B, dtype=torch.int32, device=dev)
f = lambda: la.fit_instances(depth, masks, K, image_index=ii)
for _ in range(50): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): f()
torch.cuda.synchronize()
print("per call us:", (time.perf_counter() - t0) / 500 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): f()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

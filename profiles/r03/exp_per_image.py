"""End-to-end cost of the reference's per-image pattern through the Python drop-ins: annotations (polygons / run lengths) of one image
-> filter -> boxes.  Where does the time go: host packing, uploads, the fit, the read-back?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la

dev = torch.device("cuda", 0)
H, W = 480, 640
rs = np.random.RandomState(5)
K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
depth_np = rs.uniform(0.5, 10, (H, W)).astype(np.float32)
depth = torch.as_tensor(depth_np, device=dev)


def blob(n):
    ang = np.sort(rs.uniform(0, 2 * np.pi, n))
    cx, cy = rs.uniform(0.3 * W, 0.7 * W), rs.uniform(0.3 * H, 0.7 * H)
    rad = rs.uniform(0.5, 1.0, n)
    return np.stack([cx + 0.25 * W * rad * np.cos(ang), cy + 0.3 * H * rad * np.sin(ang)], 1).round().ravel().tolist()


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for nann, nv in ((8, 40), (8, 200), (32, 40)):
    anns = [{"iscrowd": 0, "bbox": [0.0, 0.0, 1.0, 1.0], "category_id": 1 + i % 5, "segmentation": [blob(nv)], "area": 20000.0} for i in range(nann)]
    ground = np.array([[0.02, -0.97, 0.1, 1.0]] * nann)
    t_all = timeit(lambda: la.fit_annotations(anns, (W, H), depth, K, ground=ground))
    segs = [a["segmentation"] for a in anns]
    t_pack = timeit(lambda: la.pack_polygons(segs, H, W))
    polys = la.pack_polygons(segs, H, W)
    t_fit = timeit(lambda: la.fit_instances_poly(depth, polys, K, ground=ground))
    dpolys = tuple(torch.as_tensor(x, device=dev) if isinstance(x, np.ndarray) else x for x in polys)
    try:
        t_fit_dev = timeit(lambda: la.fit_instances_poly(depth, dpolys, K, ground=ground))
    except Exception as e:   # the wrapper may want host arrays
        t_fit_dev = float("nan")
    t_np = timeit(lambda: la.fit_instances_poly(depth, polys, K, ground=ground)[0].cpu().numpy())
    print(f"{nann} annotations x {nv} vertices: fit_annotations {t_all:7.1f} us | pack_polygons {t_pack:6.1f} | fit_instances_poly (host arrays in) {t_fit:6.1f} "
          f"(device arrays in {t_fit_dev:6.1f}) | + boxes to host {t_np:6.1f}")

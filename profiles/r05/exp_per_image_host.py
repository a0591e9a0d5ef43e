"""Where the time of fit_annotations(to_host=True) goes on the reference's per-image pattern (depth plane resident, annotations on the
host): the Python steps one by one and the foreign call la3d_fit_annotations_host alone (argument block built once)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
from labelany3d_amd import masks as M
from labelany3d_amd._lib import FitArgs, lib

dev = torch.device("cuda", 0)
H, W = 480, 640
rs = np.random.RandomState(5)
K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
depth = torch.as_tensor(rs.uniform(0.5, 10, (H, W)).astype(np.float32), device=dev)


def blob(n):
    ang = np.sort(rs.uniform(0, 2 * np.pi, n))
    cx, cy = rs.uniform(0.3 * W, 0.7 * W), rs.uniform(0.3 * H, 0.7 * H)
    rad = rs.uniform(0.5, 1.0, n)
    return np.stack([cx + 0.25 * W * rad * np.cos(ang), cy + 0.3 * H * rad * np.sin(ang)], 1).round().ravel().tolist()


def timeit(fn, n=400):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


for nann, nv, grounded in ((8, 40, True), (8, 40, False), (8, 200, True), (32, 40, True), (1, 40, True)):
    anns = [{"iscrowd": 0, "bbox": [0.0, 0.0, 1.0, 1.0], "category_id": 1 + i % 5, "segmentation": [blob(nv)], "area": 20000.0} for i in range(nann)]
    ground = np.array([[0.02, -0.97, 0.1, 1.0]] * nann) if grounded else None
    t_all = timeit(lambda: la.fit_annotations(anns, (W, H), depth, K, ground=ground, to_host=True))
    t_split = timeit(lambda: M.split_annotations(anns))
    segs = [a["segmentation"] for a in anns]
    t_pack = timeit(lambda: la.pack_polygons(segs, H, W))
    xy, ro, ir, _, _ = la.pack_polygons(segs, H, W)
    a = FitArgs(); a.struct_size = C.sizeof(FitArgs); a.B, a.H, a.W = nann, H, W
    a.depth = depth.data_ptr(); a.poly_xy, a.ring_offsets, a.inst_rings = xy.ctypes.data, ro.ctypes.data, ir.ctypes.data
    Kc = np.ascontiguousarray(K.reshape(-1)); a.K = Kc.ctypes.data
    if grounded: a.ground = ground.ctypes.data
    hint = np.full(nann, 20000, np.int32); a.area_hint = hint.ctypes.data
    out = np.empty((nann, 39)); st = np.empty(nann, np.int32); a.out, a.status = out.ctypes.data, st.ctypes.data
    a.filter_boundary, a.filter_min_area, a.filter_max_edge = 10, 100, 10
    ref = C.byref(a)
    t_c = timeit(lambda: lib.la3d_fit_annotations_host(ref))
    a.filter_boundary = -1
    t_c_nf = timeit(lambda: lib.la3d_fit_annotations_host(ref))
    print(f"{nann:3d} annotations x {nv:3d} vertices, ground={grounded}: fit_annotations(to_host=True) {t_all:6.1f} us | split_annotations {t_split:5.1f} | pack_polygons {t_pack:5.1f} | "
          f"la3d_fit_annotations_host {t_c:6.1f} (no filter {t_c_nf:6.1f}) | status {st.tolist()[:4]}")

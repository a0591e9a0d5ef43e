# Cost of additional polygon parts per instance (MI355X): python profiles/exp_poly_rings.py
# Every instance: its rectangle split into `nr` side-by-side parts of `nv` vertices each (same total area as the one-part case).
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from labelany3d_amd import InstanceFitter, pack_polygons
from labelany3d_amd._lib import check, lib
dev = torch.device("cuda", 0); B = 1024
depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
r0, c0, hh, ww = rects
kfull = K[None].expand(B, 3, 3).contiguous()
f = InstanceFitter(B, bench.H, bench.W, dev); st = torch.cuda.current_stream()
for nr, nv in ([(2, 40), (3, 40), (2, 80)] if os.environ.get('STAR') else [(1, 4), (1, 20), (2, 4), (2, 20), (3, 20), (5, 20), (8, 12)]):
    rs = np.random.RandomState(7); segs = []
    for a, b, h, w in zip(r0, c0, hh, ww):
        parts = []
        for q in range(nr):
            x0, x1 = b + w * q / nr, b + w * (q + 1) / nr - 1 + (2 if os.environ.get('OVERLAP') else 0)   # OVERLAP=1: boxes overlap, the general form
            cx, cy, rx, ry = (x0 + x1) / 2, a + h / 2, max((x1 - x0) / 2, 1), max(h / 2 - 1, 1)
            if nv == 4:
                parts.append([x0, a, x1, a, x1, a + h - 1, x0, a + h - 1])
            else:
                ang = np.sort(rs.uniform(0, 2 * np.pi, nv))
                rad = rs.uniform(0.35, 1.0, nv) if os.environ.get("STAR") else 1.0     # STAR=1: non-convex parts (many crossings per row)
                parts.append(np.stack([cx + rx * rad * np.cos(ang), cy + ry * rad * np.sin(ang)], 1).ravel().tolist())
        segs.append(parts)
    xy, ro, ir, _, _ = pack_polygons(segs, bench.H, bench.W)
    xy, ro, ir = (torch.as_tensor(x, device=dev) for x in (xy, ro, ir))
    def run():
        check(lib.la3d_fit_instances_poly(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(xy.data_ptr()), C.c_void_p(ro.data_ptr()), C.c_void_p(ir.data_ptr()), C.c_void_p(kfull.data_ptr()), 9, None, None, B, bench.H, bench.W, C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()), C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()), C.c_void_p(st.cuda_stream)), "poly")
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    print(f"parts={nr} vertices/part={nv}: {e0.elapsed_time(e1)/100*1e3:.1f} us  n_masked mean {float(f.aux[0][:,2].mean()):.0f}", flush=True)

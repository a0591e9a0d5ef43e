# Polygon rasteriser cost vs vertex count / convexity (MI355X): python profiles/exp_poly_scale.py  [LA3D_LIB=<ablation build> for raster-only timing]
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
for nv, jit in [(4, 0.0), (8, 0.0), (16, 0.0), (31, 0.0), (33, 0.0), (60, 0.0), (60, 0.3), (120, 0.0), (120, 0.3)]:
    rs = np.random.RandomState(7); segs = []
    for a, b, h, w in zip(r0, c0, hh, ww):
        ang = np.sort(rs.uniform(0, 2 * np.pi, nv)); rad = rs.uniform(1 - jit, 1.0, nv)
        segs.append([np.stack([b + w / 2 + (w / 2 - 1) * rad * np.cos(ang), a + h / 2 + (h / 2 - 1) * rad * np.sin(ang)], 1).ravel().tolist()])
    xy, ro, ir, _, _ = pack_polygons(segs, bench.H, bench.W)
    xy, ro, ir = (torch.as_tensor(x, device=dev) for x in (xy, ro, ir))
    def run():
        check(lib.la3d_fit_instances_poly(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(xy.data_ptr()), C.c_void_p(ro.data_ptr()), C.c_void_p(ir.data_ptr()), C.c_void_p(kfull.data_ptr()), 9, None, None, B, bench.H, bench.W, C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()), C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()), C.c_void_p(st.cuda_stream)), "poly")
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    print(f"nv={nv} jitter={jit}: {e0.elapsed_time(e1)/100*1e3:.1f} us  n_masked mean {float(f.aux[0][:,2].mean()):.0f}", flush=True)

"""Per-call time of small batches (the reference's per-image use: a handful of instances per call), u8 planes through both engines
vs run-length and polygon input (instance engine only).  usage: python profiles/r03/exp_small_batches.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from labelany3d_amd import InstanceFitter, pack_polygons
from labelany3d_amd._lib import check, lib

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for B in ([int(b) for b in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 4, 8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 320)):
    depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
    f = InstanceFitter(B, bench.H, bench.W, dev)
    kfull = K[None].expand(B, 3, 3).contiguous()
    os.environ.pop("LA3D_ENGINE", None)
    t_split = timed(lambda: f.run(depth, masks, K))
    os.environ["LA3D_ENGINE"] = "instance"
    t_inst = timed(lambda: f.run(depth, masks, K))
    os.environ.pop("LA3D_ENGINE", None)
    rc_np, ro_np = bench.rect_rle(rects)
    rle_c, rle_o = torch.as_tensor(rc_np, device=dev), torch.as_tensor(ro_np, device=dev)

    def run_rle():
        check(lib.la3d_fit_instances_rle(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(rle_c.data_ptr()),
                                         C.c_void_p(rle_o.data_ptr()), C.c_void_p(kfull.data_ptr()), 9, None, None, B, bench.H, bench.W,
                                         C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()),
                                         C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()), C.c_void_p(st.cuda_stream)), "rle")
    os.environ["LA3D_ENGINE"] = "instance"
    t_rle = timed(run_rle)
    os.environ["LA3D_ENGINE"] = "split"
    t_rle_s = timed(run_rle)
    os.environ.pop("LA3D_ENGINE", None)
    r0, c0, hh, ww = rects
    segs = [[[int(b), int(a), int(b + w - 1), int(a), int(b + w - 1), int(a + h - 1), int(b), int(a + h - 1)]] for a, b, h, w in zip(r0, c0, hh, ww)]
    xy, ro, ir, _, _ = pack_polygons(segs, bench.H, bench.W)
    xy, ro, ir = (torch.as_tensor(x, device=dev) for x in (xy, ro, ir))

    def run_poly():
        check(lib.la3d_fit_instances_poly(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(xy.data_ptr()),
                                          C.c_void_p(ro.data_ptr()), C.c_void_p(ir.data_ptr()), C.c_void_p(kfull.data_ptr()), 9, None, None, B,
                                          bench.H, bench.W, C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()),
                                          C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()), C.c_void_p(st.cuda_stream)), "poly")
    os.environ["LA3D_ENGINE"] = "instance"
    t_poly = timed(run_poly)
    os.environ["LA3D_ENGINE"] = "split"
    t_poly_s = timed(run_poly)
    os.environ.pop("LA3D_ENGINE", None)
    big = int(np.max(np.asarray(hh) * np.asarray(ww)))
    print(f"B={B:4d} (largest mask {big:6d} px): u8 split {t_split:6.1f} us | u8 instance engine {t_inst:6.1f} | run lengths: instance {t_rle:6.1f} split {t_rle_s:6.1f} | polygons: instance {t_poly:6.1f} split {t_poly_s:6.1f}")

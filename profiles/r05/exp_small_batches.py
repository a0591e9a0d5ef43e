"""Round 5: per-call time of small and medium batches by engine, mask format and camera kind (un-grounded = the separable single
pass of the instance engine; grounded = its two-pass form), through raw la3d_fit_instances_ex calls (per-call opt_engine).
usage: python profiles/r05/exp_small_batches.py [B,B,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from labelany3d_amd import InstanceFitter, pack_polygons
from labelany3d_amd._lib import FitArgs, check, lib

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ENG = {"default": 0, "instance": 1, "split": 2, "band": 3, "rows": 4}


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


def block(f, depth, K, B, engine, masks=None, rle=None, poly=None, ground=None):
    a = FitArgs()
    a.struct_size = C.sizeof(FitArgs)
    a.B, a.H, a.W = B, bench.H, bench.W
    a.depth, a.depth_plane_stride = depth.data_ptr(), bench.H * bench.W
    if masks is not None:
        a.mask = masks.data_ptr()
    elif rle is not None:
        a.rle_counts, a.rle_offsets = rle[0].data_ptr(), rle[1].data_ptr()
    else:
        a.poly_xy, a.ring_offsets, a.inst_rings = (t.data_ptr() for t in poly)
    a.K, a.k_stride = K.data_ptr(), 0
    a.filter_boundary = -1
    a.ground = None if ground is None else ground.data_ptr()
    a.out, a.status, a.aux = f.boxes[0].data_ptr(), f.status[0].data_ptr(), f.aux[0].data_ptr()
    a.workspace, a.stream = f.workspace[0].data_ptr(), st.cuda_stream
    a.opt_engine = ENG[engine]
    return a


for B in ([int(b) for b in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 4, 16, 32, 64, 128, 192, 256, 288, 320, 384, 512)):
    depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
    f = InstanceFitter(B, bench.H, bench.W, dev)
    rc_np, ro_np = bench.rect_rle(rects)
    rle = (torch.as_tensor(rc_np, device=dev), torch.as_tensor(ro_np, device=dev))
    r0, c0, hh, ww = rects
    segs = [[[int(b), int(a), int(b + w - 1), int(a), int(b + w - 1), int(a + h - 1), int(b), int(a + h - 1)]] for a, b, h, w in zip(r0, c0, hh, ww)]
    xy, ro, ir, _, _ = pack_polygons(segs, bench.H, bench.W)
    poly = tuple(torch.as_tensor(x, device=dev) for x in (xy, ro, ir))
    rs = np.random.RandomState(1)
    ground = torch.as_tensor(np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.02 * rs.randn(B, 4), device=dev)
    row = [f"B={B:4d}"]
    for gname, g in (("no ground", None), ("ground", ground)):
        for fmt, kw in (("u8", dict(masks=masks)), ("rle", dict(rle=rle)), ("poly", dict(poly=poly))):
            cells = []
            for eng in ("default", "instance", "split", "band", "rows"):
                if eng == "band" and fmt != "u8":
                    continue
                if eng == "rows" and (fmt != "u8" or g is not None or B > 512):   # (the row engine: u8 planes without a ground array)
                    continue
                a = block(f, depth, K, B, eng, ground=g, **kw)
                t = timed(lambda: check(lib.la3d_fit_instances_ex(C.byref(a)), "fit"))
                cells.append(f"{eng[:4]} {t:5.1f}")
            row.append(f"{gname}/{fmt}: " + " ".join(cells))
    print(" | ".join(row), flush=True)

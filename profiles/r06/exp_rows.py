"""Round 6: per-call time of un-grounded u8 batches by engine - instance (one workgroup per instance), rows (one launch: the band
that arrives last merges its instance), rows2 (round 5: a merge launch behind the band launch) - through raw
la3d_fit_instances_ex calls (per-call opt_engine).  LA3D_ROWS_WGS (read once per process) caps the workgroups of the row engine.
usage: python profiles/r06/exp_rows.py [B,B,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from labelany3d_amd import InstanceFitter
from labelany3d_amd._lib import FitArgs, check, lib

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ENG = {"default": 0, "instance": 1, "rows": 4, "rows2": 5}


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


def block(f, depth, K, B, engine, masks):
    a = FitArgs()
    a.struct_size = C.sizeof(FitArgs)
    a.B, a.H, a.W = B, bench.H, bench.W
    a.depth, a.depth_plane_stride = depth.data_ptr(), bench.H * bench.W
    a.mask = masks.data_ptr()
    a.K, a.k_stride = K.data_ptr(), 0
    a.filter_boundary = -1
    a.out, a.status, a.aux = f.boxes[0].data_ptr(), f.status[0].data_ptr(), f.aux[0].data_ptr()
    a.workspace, a.stream = f.workspace[0].data_ptr(), st.cuda_stream
    a.opt_engine = ENG[engine]
    return a


print(f"LA3D_ROWS_WGS={os.environ.get('LA3D_ROWS_WGS', '(default 1024)')}", flush=True)
for B in ([int(b) for b in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 16, 64, 128, 192, 256, 320, 384, 448, 512)):
    depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
    f = InstanceFitter(B, bench.H, bench.W, dev)
    cells = []
    ref = None
    for eng in ("instance", "rows2", "rows", "default"):
        a = block(f, depth, K, B, eng, masks)
        t = timed(lambda: check(lib.la3d_fit_instances_ex(C.byref(a)), "fit"))
        torch.cuda.synchronize()
        rec = f.boxes[0].clone()
        assert int((f.status[0] != 0).sum()) == 0
        if eng == "rows2":
            ref = rec
        same = "" if eng != "rows" else (" ==rows2" if torch.equal(rec, ref) else " DIFFERS")
        cells.append(f"{eng} {t:5.1f}{same}")
    print(f"B={B:4d}: " + " | ".join(cells), flush=True)

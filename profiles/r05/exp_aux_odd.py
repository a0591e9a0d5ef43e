"""The mask helpers (filter statistics, decoders to planes) on a frame of odd width against 640x480: 1024 rectangles, us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
dev = torch.device("cuda", 0)
B = 1024
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for H, W in ((480, 640), (640, 427), (375, 500)):
    rs = np.random.RandomState(1)
    segs, rles = [], []
    for i in range(B):
        h, w = rs.randint(8, int(0.62 * H)), rs.randint(8, int(0.52 * W))
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        segs.append([[c0, r0, c0 + w - 1, r0, c0 + w - 1, r0 + h - 1, c0, r0 + h - 1]])
        rles.append({"size": [H, W], "counts": [c0 * H + r0] + [h, H - h] * (w - 1) + [h, (W - c0 - w) * H + (H - r0 - h)]})
    prle = la.pack_rle(rles); drle = (torch.as_tensor(prle[0], device=dev), torch.as_tensor(prle[1], device=dev), H, W)
    pp = la.pack_polygons(segs, H, W); dpp = tuple(torch.as_tensor(x, device=dev) for x in pp[:3]) + (H, W)
    print(f"{H}x{W}: mask_stats_rle {timed(lambda: la.mask_stats_rle(drle)):7.1f} | mask_stats_poly {timed(lambda: la.mask_stats_poly(dpp)):7.1f} | "
          f"rle_decode {timed(lambda: la.rle_decode(drle)):7.1f} | poly_decode {timed(lambda: la.poly_decode(dpp)):7.1f} us", flush=True)

"""Per-call time of 1024 instances on the frame sizes COCO images come in (the tiled / single-pass forms need W % 32 == 0; other
widths take the row-linear form): u8 planes, run lengths, polygons; rectangles of the config-2 size mix scaled to the frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
from oracle import la3d_oracle as O   # (rle_encode only: input preparation)

dev = torch.device("cuda", 0)
B = 1024


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, W in ((480, 640), (427, 640), (640, 480), (640, 427), (375, 500), (500, 375), (333, 500), (612, 612)):
    rs = np.random.RandomState(1)
    depth = torch.rand((B, H, W), device=dev) * 9.5 + 0.5
    K = torch.tensor([[500.0, 0, W / 2], [0, 500.0, H / 2], [0, 0, 1]], dtype=torch.float64, device=dev)
    m = np.zeros((B, H, W), np.uint8)
    segs, rles = [], []
    for i in range(B):
        h, w = rs.randint(8, int(0.62 * H)), rs.randint(8, int(0.52 * W))
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        m[i, r0:r0 + h, c0:c0 + w] = 1
        segs.append([[c0, r0, c0 + w - 1, r0, c0 + w - 1, r0 + h - 1, c0, r0 + h - 1]])
        rles.append({"size": [H, W], "counts": [c0 * H + r0] + [h, H - h] * (w - 1) + [h, (W - c0 - w) * H + (H - r0 - h)]})
    masks = torch.as_tensor(m, device=dev)
    f = la.InstanceFitter(B, H, W, dev)
    st = torch.cuda.current_stream()
    t_u8 = timed(lambda: f.run(depth, masks, K, stream=st))
    prle = la.pack_rle(rles)
    drle = (torch.as_tensor(prle[0], device=dev), torch.as_tensor(prle[1], device=dev), H, W)
    pp = la.pack_polygons(segs, H, W)
    dpp = tuple(torch.as_tensor(x, device=dev) for x in pp[:3]) + (H, W)
    # rows padded ONCE to the next multiple of 32 (what a caller with resident planes does; the wrappers pad per call otherwise)
    dpad, _ = la.pad_depth_rows(depth)
    fr = la.InstanceFitter(B, H, dpad.shape[-1], dev)
    t_rle = timed(lambda: la.fit_instances_ex(dpad, K, rles=drle, frame_width=W, _fitter=fr))
    t_poly = timed(lambda: la.fit_instances_ex(dpad, K, polys=dpp, frame_width=W, _fitter=fr))
    t_pad = timed(lambda: la.pad_depth_rows(depth)) if W % 32 else 0.0
    px = float(m.sum()) / B
    print(f"{H}x{W} (W % 32 = {W % 32:2d}, H % 8 = {H % 8}): u8 {t_u8:7.1f} us | run lengths {t_rle:7.1f} | polygons {t_poly:7.1f} | "
          f"padding {B} private planes {t_pad:6.1f} | mean mask {px:7.0f} px, plane {H * W} px", flush=True)

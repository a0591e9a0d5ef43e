"""One-off stress run (not part of the test suite), round-5 form: random frame sizes - ANY width, not only multiples of 32 - / mask
shapes / formats / batch sizes through the default dispatch, the row engine, the instance engine (single pass and two-pass build),
the split and band engines, and the host-pointer annotation entry; every record against the NumPy oracle.
    python profiles/r05/stress_differential.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
from oracle import la3d_oracle as O
from oracle import poly_oracle as P

np_ = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
worst = 0.0
for case in range(ncase):
    W = int(rs.choice([32 * rs.randint(1, 22), rs.randint(20, 700), rs.choice([427, 500, 375, 333, 612, 426])]))
    H = int(rs.choice([8 * rs.randint(1, 75), rs.randint(8, 600)]))
    B = int(rs.choice([1, 2, 3, 7, 16, 33, 64, 130, 200, 300, 520, 1030]))
    if H * W * B > 60e6:
        B = max(1, int(60e6 // (H * W)))
    shared = rs.rand() < 0.5
    Pn = max(1, B // 3) if shared else B
    depth = rs.uniform(0.5, 10, (Pn, H, W)).astype(np.float32)
    img = np.sort(rs.randint(0, Pn, B)).astype(np.int32) if shared else None
    K = np.array([[0.8 * W, 0, W / 2 + 0.5], [0, 0.9 * W, H / 2 - 0.25], [0, 0, 1]])
    segs, masks = [], np.zeros((B, H, W), bool)
    for i in range(B):
        kind = rs.randint(0, 4)
        if kind == 0:
            h, w = rs.randint(1, H + 1), rs.randint(1, W + 1); r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            seg = [[c0, r0, c0 + w - 1, r0, c0 + w - 1, r0 + h - 1, c0, r0 + h - 1]]
        elif kind == 1:    # star, may leave the frame on every side
            n = rs.randint(3, 40); ang = np.sort(rs.uniform(0, 2 * np.pi, n)); rad = rs.uniform(0.3, 1.0, n)
            cx, cy = rs.uniform(-0.1 * W, 1.1 * W), rs.uniform(-0.1 * H, 1.1 * H)
            seg = [np.stack([cx + 0.4 * W * rad * np.cos(ang), cy + 0.4 * H * rad * np.sin(ang)], 1).round().ravel().tolist()]
        elif kind == 2:
            seg = []
            for _ in range(2):
                h, w = rs.randint(1, H // 2 + 2), rs.randint(1, W // 2 + 2); r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
                seg.append([c0, r0, c0 + w - 1, r0, c0 + w - 1, r0 + h - 1, c0, r0 + h - 1])
        else:
            x0, y0 = rs.randint(0, W), rs.randint(0, H)
            seg = [[x0, y0, min(W - 1, x0 + rs.randint(0, 40)), min(H - 1, y0 + rs.randint(0, 3)), x0, min(H - 1, y0 + rs.randint(0, 3))]]
        segs.append(seg)
        masks[i] = np.logical_or.reduce([P.create_boolean_mask_from_polygon((W, H), [part])[0] for part in seg])
    ground = None
    if rs.rand() < 0.4:
        ground = np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.05 * rs.randn(B, 4)
    dfull = depth if img is None else depth[img]
    ref, rst, _, _ = O.fit_instances(dfull, masks, K[None].repeat(B, 0), ground=ground)
    polys = la.pack_polygons(segs, H, W)
    rles = [O.rle_encode(m) for m in masks]
    res = {}
    for eng, build in ((None, None), ("rows", None), ("instance", None), ("instance", "plain"), ("split", None), ("band", None)):
        with la.scheduling(engine=eng, build=build):
            res[(eng, build, "u8")] = la.fit_instances(depth, masks, K, ground=ground, image_index=img)
            if eng in (None, "instance"):
                res[(eng, build, "rle")] = la.fit_instances_rle(depth, rles, K, ground=ground, image_index=img)
                res[(eng, build, "poly")] = la.fit_instances_poly(depth, polys, K, ground=ground, image_index=img)
    # the host-pointer annotation entry (no filter: thresholds that keep everything it can)
    anns = [{"iscrowd": 0, "bbox": [0, 0, 1, 1], "category_id": 1, "segmentation": (rles[i] if i % 2 else segs[i])} for i in range(B)]
    if B <= 300:
        dd = torch.as_tensor(depth, device="cuda")
        bb, kept, cats, hb, hs = la.fit_annotations(anns, (W, H), dd, K, ground=ground, image_index=(img if img is not None else np.arange(B, dtype=np.int32)),
                                                    to_host=True, boundary_threshold=0, scale_threshold=0)
        full = np.full((B, 39), np.nan); st = np.full(B, 6, np.int32)
        full[kept] = hb; st[kept] = hs
        drop = st == 6                                  # (the keep rule still drops flat masks: height / H <= 1/16)
        res[("host", None, "annotations")] = (np.where(drop[:, None], np.where(rst[:, None] == 0, ref, np.nan), full), np.where(drop, rst, st), None)
    ok = rst == 0
    for key, (b, s, a) in res.items():
        assert np.array_equal(np_(s), rst), (case, key, H, W, B, np_(s)[:8], rst[:8])
        if a is not None:
            assert np.array_equal(np_(a)[:, 2], masks.reshape(B, -1).sum(1)), (case, key)
        if ok.any():
            scale = np.maximum(1, np.abs(ref[ok][:, :6]).max(1))[:, None]
            err = np.abs(np_(b)[ok][:, :6] - ref[ok][:, :6]) / scale
            worst = max(worst, float(err.max()))
            assert (err <= 1e-9).all(), (case, key, H, W, B, err.max())
    if case % 15 == 0:
        print(f"case {case}: {H}x{W} B={B} shared={shared} ground={ground is not None} ok={int(ok.sum())}/{B} worst so far {worst:.2e}", flush=True)
print(f"{ncase} cases passed; worst |center/dims - oracle| / scale = {worst:.2e}")

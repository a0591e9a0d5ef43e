"""One-off stress run (not part of the test suite), round-4 form: random frame sizes / mask shapes / formats / BATCH SIZES UP TO 1500 (ordered,
self-estimating, staggered launches) through the split engine, both builds of the instance engine, the band engine and the default dispatch;
every record against the NumPy oracle and the engines against each other.  Scheduling per call (la.scheduling), not through the environment.
    python profiles/r04/stress_differential.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import labelany3d_amd as la
from oracle import la3d_oracle as O
from oracle import poly_oracle as P

np_ = lambda t: t.detach().cpu().numpy()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rs = np.random.RandomState(2026)
worst = 0.0
for case in range(ncase):
    W = 32 * rs.randint(1, 24)
    H = rs.randint(8, 600)
    B = int(rs.choice([1, 2, 5, 9, 17, 40, 130, 300, 520, 1030, 1500]))
    if H * W * B > 90e6:
        B = max(1, int(90e6 // (H * W)))
    shared = rs.rand() < 0.5
    Pn = max(1, B // 3) if shared else B
    depth = rs.uniform(0.5, 10, (Pn, H, W)).astype(np.float32)
    img = np.sort(rs.randint(0, Pn, B)).astype(np.int32) if shared else None
    K = np.array([[0.8 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]])
    segs, masks = [], np.zeros((B, H, W), bool)
    for i in range(B):
        kind = rs.randint(0, 4)
        if kind == 0:      # rectangle
            h, w = rs.randint(1, H + 1), rs.randint(1, W + 1); r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            seg = [[c0, r0, c0 + w - 1, r0, c0 + w - 1, r0 + h - 1, c0, r0 + h - 1]]
        elif kind == 1:    # star
            n = rs.randint(3, 40); ang = np.sort(rs.uniform(0, 2 * np.pi, n)); rad = rs.uniform(0.3, 1.0, n)
            cx, cy = rs.uniform(0, W), rs.uniform(0, H)
            seg = [np.stack([cx + 0.4 * W * rad * np.cos(ang), cy + 0.4 * H * rad * np.sin(ang)], 1).round().ravel().tolist()]
        elif kind == 2:    # two parts
            seg = []
            for _ in range(2):
                h, w = rs.randint(1, H // 2 + 2), rs.randint(1, W // 2 + 2); r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
                seg.append([c0, r0, c0 + w - 1, r0, c0 + w - 1, r0 + h - 1, c0, r0 + h - 1])
        else:              # thin sliver / tiny
            x0, y0 = rs.randint(0, W), rs.randint(0, H)
            seg = [[x0, y0, min(W - 1, x0 + rs.randint(0, 40)), min(H - 1, y0 + rs.randint(0, 3)), x0, min(H - 1, y0 + rs.randint(0, 3))]]
        segs.append(seg)
        masks[i] = P.create_boolean_mask_from_polygon((W, H), seg)[0]
    ground = None
    if rs.rand() < 0.5:
        ground = np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.05 * rs.randn(B, 4)
    dfull = depth if img is None else depth[img]
    ref, rst, _, _ = O.fit_instances(dfull, masks, K[None].repeat(B, 0), ground=ground)
    polys = la.pack_polygons(segs, H, W)
    rles = [O.rle_encode(m) for m in masks]
    res = {}
    for eng, build in (("split", None), ("instance", "plain"), ("instance", "retaining"), ("band", None), (None, None)):
        with la.scheduling(engine=eng, build=build):
            res[(eng, build, "u8")] = la.fit_instances(depth, masks, K, ground=ground, image_index=img)
            if eng != "band":   # (the band engine takes u8 planes only: the others fall back to the default dispatch there)
                res[(eng, build, "rle")] = la.fit_instances_rle(depth, rles, K, ground=ground, image_index=img)
                res[(eng, build, "poly")] = la.fit_instances_poly(depth, polys, K, ground=ground, image_index=img)
    ok = rst == 0
    for key, (b, s, a) in res.items():
        assert np.array_equal(np_(s), rst), (case, key, H, W, B)
        assert np.array_equal(np_(a)[:, 2], masks.reshape(B, -1).sum(1)), (case, key)
        if ok.any():
            scale = np.maximum(1, np.abs(ref[ok][:, :6]).max(1))[:, None]
            gap = np.maximum(np_(a)[ok][:, 3], 1e-300)[:, None]
            err = np.abs(np_(b)[ok][:, :6] - ref[ok][:, :6]) / scale          # center + dims: no conditioning issue
            worst = max(worst, float(err.max()))
            assert (err <= 1e-9).all(), (case, key, H, W, B, err.max())
    for eng, ret in (("split", None), ("instance", "plain"), ("instance", "retaining")):   # formats agree bit for bit within an engine / build
        assert np.array_equal(np_(res[(eng, ret, "u8")][0]), np_(res[(eng, ret, "rle")][0]), equal_nan=True), (case, eng, ret, "rle")
        assert np.array_equal(np_(res[(eng, ret, "u8")][0]), np_(res[(eng, ret, "poly")][0]), equal_nan=True), (case, eng, ret, "poly")
    if case % 20 == 0:
        print(f"case {case}: {H}x{W} B={B} shared={shared} ground={ground is not None} ok={int(ok.sum())}/{B} worst so far {worst:.2e}", flush=True)
print(f"{ncase} cases passed; worst |center/dims - oracle| / scale = {worst:.2e}")

"""Generate tests/golden/g16_fillpoly.npz from OpenCV ITSELF - the pin the polygon rasteriser still lacks.

    python tests/golden/make_golden_fillpoly.py        (needs `import cv2`; the reference pins opencv-python==4.10.0.84)

The build container has no cv2 and no network, so this script has never run there: oracle/poly_oracle.py is a restatement of
OpenCV 4.x drawing.cpp and says "PARITY UNPINNED".  On any machine with cv2 this writes seeded polygons - convex, star-shaped,
self-intersecting, tiny, with vertices on the frame border (x == W-1, y == H-1), at x == W / y == H, and far outside the frame on
every side, negative coordinates included - together with cv2.fillPoly's masks (packed bits).  tests/test_oracle_poly.py and
tests/test_gpu_poly.py pick the file up when it exists and hold the restatement and the HIP rasteriser to it bit for bit,
exactly the way create_boolean_mask_from_polygon calls it (/root/reference/src/util.py:386-400: one fillPoly per part, color 1).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def polygons(rs, W, H):
    out = []
    for k in range(240):
        kind = k % 6
        n = int(rs.randint(3, 24))
        if kind == 0:      # convex-ish inside the frame
            c = rs.uniform([10, 10], [W - 10, H - 10])
            ang = np.sort(rs.uniform(0, 2 * np.pi, n))
            r = rs.uniform(2, min(W, H) / 2)
            p = c + r * np.stack([np.cos(ang), np.sin(ang)], 1)
        elif kind == 1:    # star / non-convex
            c = rs.uniform([0, 0], [W, H])
            ang = np.sort(rs.uniform(0, 2 * np.pi, n))
            r = rs.uniform(0.2, 1.0, n) * rs.uniform(5, max(W, H))
            p = c + r[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1)
        elif kind == 2:    # random (self-intersecting) points, many outside
            p = rs.uniform([-W, -H], [2 * W, 2 * H], (n, 2))
        elif kind == 3:    # vertices exactly on / just beyond the border
            choices_x = np.array([-1, 0, 1, W - 2, W - 1, W, W + 1])
            choices_y = np.array([-1, 0, 1, H - 2, H - 1, H, H + 1])
            p = np.stack([rs.choice(choices_x, n), rs.choice(choices_y, n)], 1).astype(float)
            inside = rs.rand(n) < 0.4                     # mix in a few interior vertices
            p[inside] = rs.uniform([0, 0], [W, H], (int(inside.sum()), 2))
        elif kind == 4:    # thin slivers and near-horizontal / near-vertical sides
            a = rs.uniform([0, 0], [W, H])
            p = a + rs.uniform(-1, 1, (n, 2)) * np.array([W, 3.0]) if k % 12 < 6 else a + rs.uniform(-1, 1, (n, 2)) * np.array([3.0, H])
        else:              # tiny
            p = rs.uniform([0, 0], [W, H]) + rs.uniform(-2.5, 2.5, (n, 2))
        out.append(np.asarray(p).astype(np.int32))       # the reference truncates to int32 (src/util.py:398)
    return out


def main():
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        print(f"cv2 is not importable here ({e}); nothing generated - the polygon row stays 'parity unpinned'")
        return 0
    rs = np.random.RandomState(16)
    sets = {}
    for tag, (W, H) in {"a": (64, 48), "b": (640, 480), "c": (100, 37)}.items():
        polys = polygons(rs, W, H)
        masks = np.zeros((len(polys), H, W), np.uint8)
        for i, p in enumerate(polys):
            cv2.fillPoly(masks[i], [p], color=1)
        sets[tag + "_size"] = np.array([W, H])
        sets[tag + "_xy"] = np.concatenate(polys).astype(np.int32)
        sets[tag + "_off"] = np.concatenate([[0], np.cumsum([len(p) for p in polys])]).astype(np.int64)
        sets[tag + "_bits"] = np.packbits(masks.reshape(len(polys), -1), axis=1)
    sets["cv2_version"] = np.array([cv2.__version__])
    np.savez_compressed(os.path.join(HERE, "g16_fillpoly.npz"), **sets)
    print("wrote g16_fillpoly.npz with cv2", cv2.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Polygon rasterisation oracle (oracle/poly_oracle.py): the restatement of cv2.fillPoly the reference uses at
/root/reference/src/util.py:386-400.  PARITY UNPINNED against OpenCV itself (cv2 is not installed here, no network):
what these tests pin is everything that can be pinned without it — exact results where the rule is unambiguous
(axis-aligned rectangles, single pixels, lines), geometric bounds for convex polygons, internal consistency
(iterator vs closed form, start-vertex / orientation invariance) and agreement with an independent rasteriser
(Pillow) away from the boundary."""
import numpy as np
import pytest

from oracle import poly_oracle as P


def test_rectangles_are_inclusive():
    for (x0, y0, x1, y1) in [(3, 2, 10, 7), (0, 0, 63, 47), (5, 5, 5, 9), (7, 3, 20, 3), (4, 4, 4, 4)]:
        m = np.zeros((48, 64), np.uint8)
        P.fill_poly(m, [(x0, y0), (x1, y0), (x1, y1), (x0, y1)])
        ref = np.zeros_like(m)
        ref[y0:y1 + 1, x0:x1 + 1] = 1
        assert (m == ref).all(), (x0, y0, x1, y1)
        # orientation and start vertex do not matter
        m2 = np.zeros_like(m)
        P.fill_poly(m2, [(x1, y1), (x1, y0), (x0, y0), (x0, y1)])
        assert (m2 == ref).all()


def test_degenerate_parts():
    m = np.zeros((20, 30), np.uint8)
    P.fill_poly(m, [(4, 5)])                      # one point: a dot
    assert m.sum() == 1 and m[5, 4] == 1
    m[:] = 0
    P.fill_poly(m, [(2, 3), (12, 8)])             # two points: the line, drawn twice
    assert set(zip(*np.nonzero(m)[::-1])) == set(P.line_pixels(30, 20, (2, 3), (12, 8)))
    m[:] = 0
    P.fill_poly(m, [])                            # nothing
    assert m.sum() == 0
    m[:] = 0
    P.fill_poly(m, [(1, 1), (5, 5), (9, 9)])      # collinear: only the outline
    assert m.sum() == 9 and all(m[i, i] for i in range(1, 10))


def test_bresenham_closed_form_equals_iterator():
    rs = np.random.RandomState(0)
    for _ in range(4000):
        w, h = rs.randint(4, 40), rs.randint(4, 40)
        p1 = (int(rs.randint(-10, w + 10)), int(rs.randint(-10, h + 10)))
        p2 = (int(rs.randint(-10, w + 10)), int(rs.randint(-10, h + 10)))
        a = P.line_pixels(w, h, p1, p2)
        b = P.line_pixels_closed_form(w, h, p1, p2)
        assert sorted(a) == sorted(b), (w, h, p1, p2)
        assert all(0 <= x < w and 0 <= y < h for x, y in a)
        if a and 0 <= p1[0] < w and 0 <= p2[0] < w and 0 <= p1[1] < h and 0 <= p2[1] < h:
            assert p1 in a and p2 in a and len(a) == max(abs(p1[0] - p2[0]), abs(p1[1] - p2[1])) + 1


def _convex(rs, w, h, n):
    ang = np.sort(rs.uniform(0, 2 * np.pi, n))
    cx, cy = rs.uniform(0.3 * w, 0.7 * w), rs.uniform(0.3 * h, 0.7 * h)
    rx, ry = rs.uniform(2, 0.45 * w), rs.uniform(2, 0.45 * h)
    pts = np.stack([cx + rx * np.cos(ang), cy + ry * np.sin(ang)], 1).astype(np.int32)
    pts = np.unique(np.clip(pts, 0, [w - 1, h - 1]), axis=0)   # in-frame: clipped sides follow a different line (own test)
    if len(pts) < 3:
        return pts
    try:   # truncation to integers can dent the polygon: take the hull of the integer points
        from scipy.spatial import ConvexHull
        return pts[ConvexHull(pts).vertices]
    except Exception:  # noqa: BLE001 - collinear points
        return pts


def _signed_dist_outside(pts, xx, yy):
    """max over edges of the signed distance to the edge line (positive = outside a CCW convex polygon)."""
    p = pts.astype(np.float64)
    area = 0.5 * np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1])
    if area < 0:
        p = p[::-1]
    d = np.full(xx.shape, -np.inf)
    for a, b in zip(p, np.roll(p, -1, 0)):
        e = b - a
        n = np.hypot(*e)
        if n == 0:
            continue
        d = np.maximum(d, ((xx - a[0]) * e[1] - (yy - a[1]) * e[0]) / n)
    return d, abs(area)


def test_convex_polygons_bounds_and_invariance():
    rs = np.random.RandomState(1)
    w, h = 96, 72
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for it in range(150):
        pts = _convex(rs, w, h, rs.randint(3, 9))
        m = P.fill_poly(np.zeros((h, w), np.uint8), pts)
        d, area = _signed_dist_outside(pts, xx, yy)
        if area < 1:
            continue
        assert m[d < -1e-9].all(), it                 # every pixel centre strictly inside is set
        assert not m[d > 1.0].any(), it               # nothing farther than one pixel outside
        k = rs.randint(len(pts))
        assert (P.fill_poly(np.zeros((h, w), np.uint8), np.roll(pts, k, 0)) == m).all()
        assert (P.fill_poly(np.zeros((h, w), np.uint8), pts[::-1]) == m).all()


def test_against_pillow_interior():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageDraw

    rs = np.random.RandomState(2)
    w, h = 96, 72
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for it in range(60):
        pts = _convex(rs, w, h, rs.randint(3, 8))
        m = P.fill_poly(np.zeros((h, w), np.uint8), pts)
        im = Image.new("L", (w, h), 0)
        ImageDraw.Draw(im).polygon([tuple(map(int, p)) for p in pts], fill=1, outline=1)
        q = np.asarray(im)
        d, area = _signed_dist_outside(pts, xx, yy)
        if area < 1:
            continue
        inner = d < -1.0
        assert (m[inner] == q[inner]).all() and m[inner].all(), it
        assert not (m != q)[np.abs(d) > 1.5].any(), it   # the two rasterisers differ only on the boundary band


def test_even_odd_inside_one_ring_and_union_across_parts():
    # a self-overlapping ring (bow-tie) and a ring wound twice: even-odd inside ONE fillPoly call
    m = P.fill_poly(np.zeros((40, 40), np.uint8), [(5, 5), (30, 5), (30, 30), (5, 30), (5, 5), (30, 5), (30, 30), (5, 30)])
    assert m[6:30, 6:30].sum() == 0 and m[5, 5:31].all() and m[30, 5:31].all()   # interior cancels, outline stays
    # the reference fills every part on its own: parts are OR-ed
    mask, height = P.create_boolean_mask_from_polygon((40, 40), [[5, 5, 20, 5, 20, 20, 5, 20], [10.9, 10.2, 30.7, 10, 30, 30.99, 10, 30]])
    assert mask[5:21, 5:21].all() and mask[10:31, 10:31].all() and mask.sum() == 16 * 16 + 21 * 21 - 11 * 11
    assert height == 26


def test_out_of_frame_vertices_are_clipped():
    w, h = 32, 24
    m = P.fill_poly(np.zeros((h, w), np.uint8), [(-5, -5), (40, -5), (40, 30), (-5, 30)])
    assert m.all()
    m = P.fill_poly(np.zeros((h, w), np.uint8), [(10, -8), (40, 12), (10, 30), (-20, 12)])
    assert m[12, :].all() and m[0, 0] == 1 and m[0, w - 1] == 0 and m[h - 1, w - 1] == 0 and 0 < m.sum() < w * h
    m = P.fill_poly(np.zeros((h, w), np.uint8), [(40, 2), (50, 2), (50, 20), (40, 20)])   # entirely outside
    assert m.sum() == 0


def test_reference_truncation_of_float_vertices():
    # np.array(polygon).reshape(-1, 2).astype(np.int32) truncates toward zero (src/util.py:398)
    mask, height = P.create_boolean_mask_from_polygon((64, 48), [[3.9, 2.9, 10.99, 2.2, 10.5, 7.7, 3.1, 7.999]])
    ref = np.zeros((48, 64), bool)
    ref[2:8, 3:11] = True
    assert (mask == ref).all() and height == 6
    with pytest.raises(ValueError):
        P.create_boolean_mask_from_polygon((64, 48), [[1, 2, 3]])   # odd number of coordinates: reshape fails


def _converter_cases():
    """tests/golden/g13_polygons.json.gz: masks + the polygons the reference's own converter (binary_mask_to_polygon,
    src/download_coconut.py:178-199, run with scikit-image by make_golden_polygons.py) writes for them"""
    import gzip
    import json
    import os

    g = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_polygons.json.gz"), "rt"))
    for c in g["cases"]:
        H, W = c["size"]
        flat = np.zeros(H * W, bool)
        pos = 0
        for i, n in enumerate(c["counts"]):
            if i & 1:
                flat[pos:pos + n] = True
            pos += n
        yield H, W, flat.reshape(W, H).T, c["polygons"]


def test_round_trip_through_the_reference_converter():
    """mask -> the reference converter's polygons -> fillPoly restatement: the result stays within 3 px of the (hole-filled:
    every part is filled with 1, so holes close, as in the reference) mask on both sides, and overlaps it like a tolerance-2
    simplification should.  A wrong fill rule (shifted by a pixel, missing outline, inverted pairing) breaks these bounds."""
    from scipy import ndimage

    n = 0
    for H, W, mask, polys in _converter_cases():
        got, _ = P.create_boolean_mask_from_polygon((W, H), polys)
        filled = ndimage.binary_fill_holes(mask)
        core = ndimage.binary_erosion(filled, iterations=3)
        halo = ndimage.binary_dilation(filled, iterations=3)
        assert not (got & ~halo).any()
        assert (core & ~got).sum() <= 0.002 * filled.sum()          # (parts of < 3 vertices are dropped by the converter)
        iou = (got & filled).sum() / (got | filled).sum()
        assert iou > (0.98 if len(polys) == 1 and filled.sum() > 10000 else 0.8), (H, W, len(polys), iou)   # thin / ragged shapes lose more
        n += 1
    assert n == 20


def _g16():
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_fillpoly.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/g16_fillpoly.npz absent: generate it with tests/golden/make_golden_fillpoly.py on a machine with cv2 "
                    "(the build container has none) - until then the polygon row is 'parity unpinned'")
    return np.load(path, allow_pickle=False)


def test_g16_fillpoly_against_opencv_itself():
    """cv2.fillPoly's own masks (tests/golden/make_golden_fillpoly.py) against the restatement, bit for bit - runs only where the
    fixture exists."""
    g = _g16()
    for tag in ("a", "b", "c"):
        W, H = (int(v) for v in g[tag + "_size"])
        off, xy = g[tag + "_off"], g[tag + "_xy"]
        want = np.unpackbits(g[tag + "_bits"], axis=1)[:, : H * W].reshape(-1, H, W)
        for i in range(len(off) - 1):
            m = np.zeros((H, W), np.uint8)
            P.fill_poly(m, xy[off[i]:off[i + 1]])
            assert np.array_equal(m, want[i]), (tag, i, int((m != want[i]).sum()))

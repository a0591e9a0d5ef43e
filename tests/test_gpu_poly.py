"""GPU tests of polygon mask ingestion (the branch every kept COCONut instance takes in the reference:
create_boolean_mask_from_polygon -> cv2.fillPoly, /root/reference/src/util.py:386-400).

PARITY UNPINNED against OpenCV: cv2 is not installed in the build container, so the checker is oracle/poly_oracle.py, a
restatement of OpenCV 4.x drawing.cpp (see its header and tests/test_oracle_poly.py for what pins it).  The HIP
rasteriser must agree with that restatement bit for bit on every polygon below."""
import numpy as np
import pytest

from .conftest import SCHED

from oracle import la3d_oracle as O
from oracle import poly_oracle as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


def np_(t):
    return t.detach().cpu().numpy()


class _engine:
    """Pin one engine for the calls inside (labelany3d_amd.options): up to 288 instances of polygon / run-length input and up to 272 of u8 planes
    take the split engine, whose partial sums are grouped differently from the instance engine's - comparisons "bit for bit" are
    between calls of the SAME engine."""

    def __init__(self, name="instance"):
        self.name = name

    def __enter__(self):
        self.prev = SCHED().engine
        SCHED().engine = self.name

    def __exit__(self, *a):
        SCHED().engine = self.prev


_instance_engine = _engine


def _star(rs, w, h, n, jitter=0.6, margin=0.0):
    """A simple (mostly) star-shaped ring with n vertices; margin > 0 pushes vertices out of the frame."""
    ang = np.sort(rs.uniform(0, 2 * np.pi, n))
    cx, cy = rs.uniform(0.2 * w, 0.8 * w), rs.uniform(0.2 * h, 0.8 * h)
    rad = rs.uniform(1 - jitter, 1.0, n)
    rx, ry = rs.uniform(2, (0.5 + margin) * w), rs.uniform(2, (0.5 + margin) * h)
    return np.stack([cx + rx * rad * np.cos(ang), cy + ry * rad * np.sin(ang)], 1)


def _random_segmentation(rs, w, h, kind):
    if kind == 0:     # one in-frame star ring, float coordinates (the reference truncates them)
        q = np.clip(_star(rs, w, h, rs.randint(3, 40)), 0, [w - 1, h - 1])
        return [q.ravel().tolist()]
    if kind == 1:     # vertices outside the frame: clipLine paths
        return [_star(rs, w, h, rs.randint(3, 25), margin=0.5).ravel().tolist()]
    if kind == 2:     # several parts, overlapping or not
        return [np.clip(_star(rs, w, h, rs.randint(3, 15)), 0, [w - 1, h - 1]).ravel().tolist() for _ in range(rs.randint(2, 5))]
    if kind == 3:     # self-intersecting: random vertex order (even-odd inside one fillPoly call)
        return [rs.uniform(0, [w - 1, h - 1], (rs.randint(4, 12), 2)).ravel().tolist()]
    if kind == 4:     # degenerate parts: a dot, a segment, repeated vertices, collinear points
        x, y = rs.randint(0, w), rs.randint(0, h)
        return [[x, y], [x, y, (x + 7) % w, (y + 3) % h], [3, 3, 3, 3, 9, 9, 9, 9], [1, 1, 5, 5, 9, 9]]
    if kind == 5:     # a comb: many crossings per scanline (more than the 8 a scanline keeps per sweep)
        teeth = rs.randint(6, 14)
        xs = np.linspace(2, w - 3, 2 * teeth + 1)
        top, bot = rs.randint(1, h // 3), rs.randint(2 * h // 3, h - 1)
        pts = [(xs[0], bot)]
        for k in range(2 * teeth):
            pts.append((xs[k + 1], top if k % 2 == 0 else bot - 2))
        pts.append((xs[-1], bot))
        return [np.asarray(pts).ravel().tolist()]
    if kind == 6:     # axis-aligned rectangle, often flush with a frame border (spans that run to the last column / row)
        x0, x1 = sorted(rs.randint(0, w, 2))
        y0, y1 = sorted(rs.randint(0, h, 2))
        t = rs.randint(0, 6)
        if t == 0:
            x1 = w - 1
        elif t == 1:
            x0 = 0
        elif t == 2:
            y1 = h - 1
        elif t == 3:
            x0, y0, x1, y1 = 0, 0, w - 1, h - 1
        elif t == 4:
            x1 = w + 5          # beyond the right border: clipped side
        return [[x0, y0, x1, y0, x1, y1, x0, y1]]
    # very many vertices (more than one stage of 32 sides, long sweeps)
    return [np.clip(_star(rs, w, h, rs.randint(100, 400), jitter=0.3), 0, [w - 1, h - 1]).ravel().tolist()]


@pytest.mark.parametrize("H,W,seed", [(48, 64, 0), (37, 53, 1), (480, 640, 2), (120, 100, 3)])
def test_poly_decode_equals_fillpoly_oracle(la, H, W, seed):
    rs = np.random.RandomState(seed)
    n = 40 if H * W > 100000 else 90          # 260 polygons over the four frames
    segs = [_random_segmentation(rs, W, H, k % 8) for k in range(n)]
    # deterministic border cases: parts flush with / beyond every frame border, few and many sides (both rasteriser forms)
    ring = lambda x0, y0, x1, y1, k: [np.concatenate([np.stack([np.linspace(x0, x1, k), np.full(k, y0)], 1),     # noqa: E731
                                                      np.stack([np.full(k, x1), np.linspace(y0, y1, k)], 1),
                                                      np.stack([np.linspace(x1, x0, k), np.full(k, y1)], 1),
                                                      np.stack([np.full(k, x0), np.linspace(y1, y0, k)], 1)]).ravel().tolist()]
    for k in (2, 12):
        segs += [ring(W // 3, 2, W - 1, H // 2, k), ring(0, H // 3, W // 2, H - 1, k), ring(0, 0, W - 1, H - 1, k),
                 ring(W // 2, -7, W + 9, H + 4, k), ring(-5, -5, 3, 3, k)]
    segs.append([])                            # an instance without parts: empty mask
    polys = la.pack_polygons(segs, H, W)
    got = np_(la.poly_decode(polys))
    assert got.shape == (len(segs), H, W) and got.dtype == np.bool_
    stats = np_(la.mask_stats_poly(polys))
    for i, seg in enumerate(segs):
        want, height = P.create_boolean_mask_from_polygon((W, H), seg)
        bad = np.argwhere(got[i] != want)
        assert bad.size == 0, (i, i % 8, bad[:5], seg if len(str(seg)) < 400 else "...")
        ws = O.mask_stats(want)
        assert tuple(stats[i]) == ws, (i, stats[i], ws)
        assert ws[2] == height                 # get_maximum_height
    assert got[-1].sum() == 0


def test_fit_from_polygons_equals_fit_from_planes(la):
    """The composed path fed with polygon parts gives the records of the same path fed with the rasterised planes
    (bit for bit), and those agree with the oracle fit on the oracle's masks."""
    rs = np.random.RandomState(5)
    B, H, W = 40, 480, 640
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    segs = [_random_segmentation(rs, W, H, k % 8) for k in range(B - 1)] + [[]]
    ground = np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.02 * rs.randn(B, 4)
    polys = la.pack_polygons(segs, H, W)
    masks = la.poly_decode(polys)
    for eng in ("split", "instance"):      # the split engine's decode front end (scan_bits_kernel) and the fused instance engine
        with _engine(eng):
            b1, s1, a1 = la.fit_instances_poly(depth, polys, K, ground=ground)
            b2, s2, a2 = la.fit_instances(depth, masks, K, ground=ground)
            b3, s3, a3 = la.fit_instances_rle(depth, [O.rle_encode(m) for m in np_(masks).astype(bool)], K, ground=ground)
        np.testing.assert_array_equal(np_(s1), np_(s2))
        np.testing.assert_array_equal(np_(b1), np_(b2))
        np.testing.assert_array_equal(np_(a1), np_(a2))
        np.testing.assert_array_equal(np_(s3), np_(s2))
        np.testing.assert_array_equal(np_(b3), np_(b2))
    # the library's own choice at 40 instances: the split engine for polygon parts, the band engine for u8 planes (round 4) - two
    # engines, two groupings of the fp64 partial sums: equal to rounding, not bit for bit
    b1, s1, a1 = la.fit_instances_poly(depth, polys, K, ground=ground)
    b2 = np_(la.fit_instances(depth, masks, K, ground=ground)[0])
    ok = np_(s1) == 0
    np.testing.assert_allclose(np_(b1)[ok][:, :15], b2[ok][:, :15], rtol=1e-10, atol=1e-10)
    assert np_(s1)[-1] == 1                    # the empty instance: "No valid points"
    for i in range(0, B - 1, 5):
        want, _ = P.create_boolean_mask_from_polygon((W, H), segs[i])
        ref, rst, _, _ = O.fit_instances(depth[i:i + 1], want[None], K[None], ground=ground[i:i + 1])
        assert np_(s1)[i] == rst[0]
        if rst[0] == 0:
            np.testing.assert_allclose(np_(b1)[i, :15], ref[0, :15], rtol=0, atol=1e-9)
            np.testing.assert_allclose(np_(b1)[i, 15:], ref[0, 15:], rtol=0, atol=2e-2)   # fp16-quantised corners


def test_fit_from_polygons_large_batch_shared_depth(la):
    """More instances than one resident round, one depth plane per image (the COCO layout), launch order on."""
    rs = np.random.RandomState(6)
    P_img, H, W = 60, 480, 640
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    depth = rs.uniform(0.5, 10, (P_img, H, W)).astype(np.float32)
    img = np.repeat(np.arange(P_img), rs.randint(3, 12, P_img)).astype(np.int32)
    segs = [_random_segmentation(rs, W, H, [0, 2, 6, 7][k % 4]) for k in range(len(img))]
    polys = la.pack_polygons(segs, H, W)
    b1, s1, _ = la.fit_instances_poly(depth, polys, K, image_index=img)
    with _instance_engine():
        b2, s2, _ = la.fit_instances(depth, la.poly_decode(polys), K, image_index=img)
    np.testing.assert_array_equal(np_(s1), np_(s2))
    np.testing.assert_array_equal(np_(b1), np_(b2))


def test_filter_annotations_polygon_and_rle_branches(la):
    """read_bounding_boxes_segmentations (src/util.py:336-383): crowd skip, RLE height = rows with a pixel, polygon
    height = last - first + 1, keep rule :375 — against the oracle's restatement of the same rule."""
    rs = np.random.RandomState(9)
    H, W = 240, 320
    annos, want_keep = [], []
    for i in range(60):
        kind = i % 6
        a = {"iscrowd": 0, "bbox": [float(i), 1.0, 2.0, 3.0], "category_id": 1 + i % 80}
        if kind == 0:      # crowd: skipped before anything is looked at
            a["iscrowd"] = 1
            a["segmentation"] = O.rle_encode(rs.rand(H, W) < 0.3)
            annos.append(a)
            continue
        if kind == 1:      # RLE (uncompressed or compressed string)
            m = np.zeros((H, W), bool)
            h, w = rs.randint(2, 120), rs.randint(2, 160)
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            m[r0:r0 + h, c0:c0 + w] = True
            m[r0 + h // 2, :] &= rs.rand(W) < 0.5       # a row with holes, maybe an empty row: rows != span
            rle = O.rle_encode(m)
            if i % 12 == 1:
                rle["counts"] = O.rle_to_string(rle["counts"])
            a["segmentation"] = rle
            st, from_rle = O.mask_stats(m), True
        else:              # polygon parts
            seg = _random_segmentation(rs, W, H, [0, 2, 6, 0, 1][kind - 1] if kind > 1 else 0)
            a["segmentation"] = seg
            m, _ = P.create_boolean_mask_from_polygon((W, H), seg)
            st, from_rle = O.mask_stats(m), False
        annos.append(a)
        want_keep.append((len(annos) - 1, O.keep_instance(st, H, from_rle)))
    bboxes, segs, kept, cats = la.filter_annotations(annos, (W, H))
    want = [i for i, k in want_keep if k]
    assert list(kept) == want and 0 < len(want) < len(want_keep)
    assert bboxes == [annos[i]["bbox"] for i in want] and cats == [annos[i]["category_id"] for i in want]
    masks = np_(la.segmentations_to_masks(segs, H, W))
    for j, i in enumerate(want):
        seg = annos[i]["segmentation"]
        if isinstance(seg, dict):
            c = seg["counts"]
            ref = O.rle_decode(O.rle_from_string(c) if isinstance(c, (str, bytes)) else c, H, W)
        else:
            ref, _ = P.create_boolean_mask_from_polygon((W, H), seg)
        np.testing.assert_array_equal(masks[j], ref.astype(bool))


def test_polygon_argument_errors(la):
    with pytest.raises(ValueError):
        la.pack_polygons([[[1, 2, 3]]], 48, 64)          # odd number of coordinates: reshape(-1, 2) fails like the reference
    with pytest.raises(ValueError):
        la.pack_polygons([[[1, 2, 3, 4]]])               # frame size is required
    polys = la.pack_polygons([], 48, 64)
    assert tuple(la.poly_decode(polys).shape) == (0, 48, 64)


def test_the_reference_converters_own_polygons(la, monkeypatch):
    """The polygon shapes of the reference's data: binary_mask_to_polygon (src/download_coconut.py:178-199) run on 20 masks
    (tests/golden/make_golden_polygons.py): half-pixel float vertices, up to 596 parts / 5093 points per instance.  The HIP
    rasteriser (decode, filter statistics, and inside the fit kernel) agrees with the fillPoly restatement bit for bit."""
    from tests.test_oracle_poly import _converter_cases

    by_size = {}
    for H, W, mask, polys in _converter_cases():
        by_size.setdefault((H, W), []).append(polys)
    assert len(by_size) == 4
    rs = np.random.RandomState(1)
    for (H, W), segs in by_size.items():
        packed = la.pack_polygons(segs, H, W)
        got = np_(la.poly_decode(packed))
        stats = np_(la.mask_stats_poly(packed))
        want = []
        for i, seg in enumerate(segs):
            w, height = P.create_boolean_mask_from_polygon((W, H), seg)
            want.append(w)
            assert np.array_equal(got[i], w), (H, W, i, np.argwhere(got[i] != w)[:5])
            assert tuple(stats[i]) == O.mask_stats(w) and stats[i][2] == height
        depth = rs.uniform(0.5, 10, (len(segs), H, W)).astype(np.float32)
        K = np.array([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
        with _instance_engine():
            b1, s1, a1 = la.fit_instances_poly(depth, packed, K)
        # polygon input takes the plain build; pin it for the planes too: the 107 k-px blob has more active tiles (> 912) than the
        # plain build's list holds and is walked densely there, which groups the partial sums differently from the no-cull
        # build's longer list (last-ulp differences; checked to rounding below)
        # (round 5: both inputs take the default build - the separable single pass where the camera allows - and must agree bit for
        # bit; the two-pass plain build agrees to rounding)
        with _instance_engine():
            b2, s2, a2 = la.fit_instances(depth, np.stack(want), K)
        monkeypatch.setattr(SCHED(), "build", "plain")
        with _instance_engine():
            b3, s3, _ = la.fit_instances(depth, np.stack(want), K)
        monkeypatch.setattr(SCHED(), "build", None)
        np.testing.assert_array_equal(np_(s1), np_(s2))
        np.testing.assert_array_equal(np_(b1), np_(b2))
        np.testing.assert_array_equal(np_(a1)[:, 2], [w.sum() for w in want])
        np.testing.assert_array_equal(np_(s1), np_(s3))
        with _engine("split"):          # the same parts through the split engine's decode front end vs its u8 scan: bit for bit
            b4, s4, _ = la.fit_instances_poly(depth, packed, K)
            b5, s5, _ = la.fit_instances(depth, np.stack(want), K)
        np.testing.assert_array_equal(np_(s4), np_(s5))
        np.testing.assert_array_equal(np_(b4), np_(b5))
        np.testing.assert_array_equal(np_(s4), np_(s1))
        np.testing.assert_allclose(np_(b4)[:, :15], np_(b1)[:, :15], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(np_(b3)[:, :15], np_(b1)[:, :15], rtol=1e-12, atol=1e-12)


def test_fused_instance_filter_polygons_and_run_lengths(la):
    """*_filtered entry points: the reference's keep rule (src/util.py:375) evaluated inside the fit launch on the bit image it has
    just built.  The statistics equal mask_stats_poly / mask_stats_rle, the decisions equal the oracle's keep_instance (polygon
    branch: span; RLE branch: rows), kept instances carry exactly the records of the unfiltered call, dropped ones status 6."""
    rs = np.random.RandomState(21)
    H, W = 480, 640
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    segs = [_random_segmentation(rs, W, H, k % 8) for k in range(60)]
    segs += [[[5, 5, 300, 5, 300, 200, 5, 200]],                   # touches the 10-px border strips: truncated
             [[100, 100, 104, 100, 104, 400, 100, 400]],             # 5 px wide, tall: kept (area 1505)
             [[100, 100, 400, 100, 400, 120, 100, 120]],             # 21 rows: 21 / 480 < 0.0625 -> dropped
             [[100, 100, 400, 100, 400, 130, 100, 130]],             # 31 rows: kept
             [[200, 200, 205, 200, 205, 205, 200, 205]],             # 36 px: below the area threshold
             []]
    B = len(segs)
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    polys = la.pack_polygons(segs, H, W)
    with _instance_engine():        # the fused filter lives in the instance engine: compare with that engine's unfiltered records
        b0, s0, a0 = la.fit_instances_poly(depth, polys, K)
    b1, s1, a1, st = la.fit_instances_poly(depth, polys, K, filter=True)
    want_stats = np_(la.mask_stats_poly(polys))
    np.testing.assert_array_equal(np_(st), want_stats)
    keep = np.array([O.keep_instance(tuple(r), H, False) for r in want_stats])
    assert keep.sum() > 10 and (~keep).sum() > 10
    assert keep[-5] and not keep[-4] and keep[-3] and not keep[-2] and not keep[-1] and not keep[-6]
    np.testing.assert_array_equal(np_(s1)[keep], np_(s0)[keep])
    np.testing.assert_array_equal(np_(b1)[keep], np_(b0)[keep])
    np.testing.assert_array_equal(np_(a1)[keep], np_(a0)[keep])
    assert (np_(s1)[~keep] == 6).all() and np.isnan(np_(b1)[~keep]).all()
    np.testing.assert_array_equal(np_(a1)[~keep, 2], want_stats[~keep, 0])
    # other thresholds
    _, s2, _, _ = la.fit_instances_poly(depth, polys, K, filter={"scale_threshold": 2000, "truncation_pixels": 1, "boundary_threshold": 30})
    st30 = np_(la.mask_stats_poly(polys, 30))
    keep2 = (16 * st30[:, 2] > H) & (st30[:, 3] < 1) & (st30[:, 0] >= 2000)
    np.testing.assert_array_equal(np_(s2) != 6, keep2)
    with pytest.raises(ValueError, match="unknown filter keys"):
        la.fit_instances_poly(depth, polys, K, filter={"area": 3})
    # run lengths: same masks, RLE branch of the rule (rows holding a pixel)
    masks = np_(la.poly_decode(polys))
    rles = [O.rle_encode(m) for m in masks]
    with _instance_engine():
        r0 = la.fit_instances_rle(depth, rles, K)
    r1 = la.fit_instances_rle(depth, rles, K, filter=True)
    want_r = np_(la.mask_stats_rle(rles))
    np.testing.assert_array_equal(np_(r1[3]), want_r)
    keep_r = np.array([O.keep_instance(tuple(r), H, True) for r in want_r])
    np.testing.assert_array_equal(np_(r1[1]) != 6, keep_r)
    np.testing.assert_array_equal(np_(r1[0])[keep_r], np_(r0[0])[keep_r])
    # a frame whose rows are not word aligned
    Hs, Ws = 120, 200
    segs2 = [_random_segmentation(rs, Ws, Hs, k % 8) for k in range(24)]
    p2 = la.pack_polygons(segs2, Hs, Ws)
    d2 = rs.uniform(0.5, 10, (24, Hs, Ws)).astype(np.float32)
    K2 = np.array([[150.0, 0, 100], [0, 150.0, 60], [0, 0, 1]])
    _, s3, _, st3 = la.fit_instances_poly(d2, p2, K2, filter=True)
    w3 = np_(la.mask_stats_poly(p2))
    np.testing.assert_array_equal(np_(st3), w3)
    np.testing.assert_array_equal(np_(s3) != 6, [O.keep_instance(tuple(r), Hs, False) for r in w3])


def test_fit_annotations_one_pass(la):
    """fit_annotations = filter_annotations + fit of the kept segmentations, with one decode per annotation (the filter is
    evaluated inside the fit launch): same kept set, boxes, categories and records as the two-step route."""
    rs = np.random.RandomState(33)
    H, W = 240, 320
    anns = []
    for i in range(40):
        seg = _random_segmentation(rs, W, H, i % 8)
        a = {"iscrowd": int(i % 11 == 0), "bbox": [float(i), 1.0, 2.0, 3.0], "category_id": 1 + i % 5, "segmentation": seg,
             "area": float(rs.randint(10, 50000))}     # (COCO carries the mask area; here arbitrary: a hint only orders the work)
        if i % 3 == 0:      # every third annotation as an RLE of the same shape
            m, _ = P.create_boolean_mask_from_polygon((W, H), seg)
            a["segmentation"] = O.rle_encode(m)
        anns.append(a)
    anns.append({"iscrowd": 0, "bbox": [0, 0, 1, 1], "category_id": 9})     # no segmentation: skipped
    depth = rs.uniform(0.5, 10, (H, W)).astype(np.float32)
    K = np.array([[250.0, 0, 160], [0, 250.0, 120], [0, 0, 1]])
    ground = np.array([[0.02, -0.97, 0.1, 1.0]] * len(anns)) + 0.02 * rs.randn(len(anns), 4)
    bb, kept, cats, boxes, status = la.fit_annotations(anns, (W, H), depth, K, ground=ground)
    bb0, segs0, kept0, cats0 = la.filter_annotations(anns, (W, H))
    np.testing.assert_array_equal(kept, kept0)
    assert bb == bb0 and cats == cats0 and len(kept) > 5
    masks = la.segmentations_to_masks(segs0, H, W)
    with _instance_engine():
        b0, s0, _ = la.fit_instances(depth, masks, K, ground=ground[kept0])
    np.testing.assert_array_equal(np_(status), np_(s0))
    np.testing.assert_array_equal(np_(boxes), np_(b0))
    out = la.fit_annotations([], (W, H), depth, K)
    assert out[0] == [] and out[3].shape == (0, 39)


def test_fit_annotations_host_entry_equals_the_device_route(la):
    """la3d_fit_annotations_host (``fit_annotations(to_host=True)`` with the depth planes resident and every other argument on the
    host): ONE foreign call per segmentation kind - inputs through the library's pinned block, records back through it - must
    return exactly what the tensor route returns: kept set, boxes, categories, records, status; with and without ground planes,
    one plane or several planes + image_index, per-plane intrinsics."""
    import ctypes as C

    import torch
    from labelany3d_amd._lib import FitArgs, lib

    rs = np.random.RandomState(71)
    H, W, P_ = 240, 320, 3
    anns = []
    for i in range(36):
        seg = _random_segmentation(rs, W, H, i % 8)
        a = {"iscrowd": int(i % 13 == 0), "bbox": [float(i), 1.0, 2.0, 3.0], "category_id": 1 + i % 5, "segmentation": seg}
        if i % 2 == 0:
            a["area"] = float(rs.randint(10, 50000))      # mixed: the hint is only used when every annotation of a kind has one
        if i % 3 == 0:
            m, _ = P.create_boolean_mask_from_polygon((W, H), seg)
            a["segmentation"] = O.rle_encode(m)
        anns.append(a)
    anns.append({"iscrowd": 0, "bbox": [0, 0, 1, 1], "category_id": 9})
    depth = rs.uniform(0.5, 10, (P_, H, W)).astype(np.float32)
    depth_d = torch.as_tensor(depth, device="cuda")
    K = np.array([[250.0, 0, 160], [0, 250.0, 120], [0, 0, 1]])
    Ks = np.stack([K, K * np.array([[1.1], [0.9], [1.0]]), K])
    ground = np.array([[0.02, -0.97, 0.1, 1.0]] * len(anns)) + 0.02 * rs.randn(len(anns), 4)
    img = rs.randint(0, P_, len(anns)).astype(np.int32)
    cases = [dict(depth=0, K=K, ground=None, image_index=None), dict(depth=0, K=K, ground=ground, image_index=None),
             dict(depth=None, K=K, ground=ground, image_index=img), dict(depth=None, K=Ks, ground=None, image_index=img)]
    for c in cases:
        dh = depth[c["depth"]] if c["depth"] is not None else depth
        dd = depth_d[c["depth"]] if c["depth"] is not None else depth_d
        kw = dict(ground=c["ground"], image_index=c["image_index"])
        want = la.fit_annotations(anns, (W, H), dh, c["K"], to_host=True, **kw)            # host depth: the tensor route
        got = la.fit_annotations(anns, (W, H), dd, c["K"], to_host=True, **kw)             # resident depth: the host entry
        assert got[0] == want[0] and got[2] == want[2] and len(got[1]) > 5
        np.testing.assert_array_equal(got[1], want[1])
        np.testing.assert_array_equal(got[4], want[4])
        np.testing.assert_array_equal(got[3], want[3])
        assert isinstance(got[3], np.ndarray) and got[3].dtype == np.float64 and got[4].dtype == np.int32
    # one depth plane per annotation, no image_index: every kind's instances pick their own plane
    dn = rs.uniform(0.5, 10, (len(anns), H, W)).astype(np.float32)
    g3 = la.fit_annotations(anns, (W, H), torch.as_tensor(dn, device="cuda"), K, to_host=True)
    w3 = la.fit_annotations(anns, (W, H), dn, K, to_host=True, image_index=np.arange(len(anns), dtype=np.int32))
    np.testing.assert_array_equal(g3[1], w3[1]); np.testing.assert_array_equal(g3[3], w3[3])
    with pytest.raises(ValueError, match="image_index"):
        la.fit_annotations(anns, (W, H), depth_d[:2], K, to_host=True)
    # other thresholds travel
    w2 = la.fit_annotations(anns, (W, H), depth[0], K, to_host=True, boundary_threshold=30, scale_threshold=2000)
    g2 = la.fit_annotations(anns, (W, H), depth_d[0], K, to_host=True, boundary_threshold=30, scale_threshold=2000)
    np.testing.assert_array_equal(g2[1], w2[1]); np.testing.assert_array_equal(g2[3], w2[3])
    assert len(g2[1]) < len(got[1])
    out = la.fit_annotations([], (W, H), depth_d[0], K, to_host=True)
    assert out[0] == [] and out[3].shape == (0, 39)
    # the C entry itself: aux / stats outputs, error behaviour
    segs = [a["segmentation"] for a in anns if not a.get("iscrowd") and isinstance(a.get("segmentation"), list)]
    xy, ro, ir, Hh, Ww = la.pack_polygons(segs, H, W)
    B = len(segs)
    a = FitArgs(); a.struct_size = C.sizeof(FitArgs)
    a.B, a.H, a.W = B, H, W
    a.depth = depth_d[0].data_ptr()
    a.poly_xy, a.ring_offsets, a.inst_rings = xy.ctypes.data, ro.ctypes.data, ir.ctypes.data
    Kc = np.ascontiguousarray(K.reshape(-1)); a.K = Kc.ctypes.data
    out = np.empty((B, 39)); st = np.empty(B, np.int32); aux = np.empty((B, 4)); stats = np.empty((B, 4), np.int32)
    a.out, a.status, a.aux, a.stats = out.ctypes.data, st.ctypes.data, aux.ctypes.data, stats.ctypes.data
    a.filter_boundary, a.filter_min_area, a.filter_max_edge = 10, 100, 10
    assert lib.la3d_fit_annotations_host(C.byref(a)) == 0
    b1, s1, a1, st1 = la.fit_instances_poly(depth[0], (xy, ro, ir, H, W), K, filter=True)
    np.testing.assert_array_equal(out, np_(b1)); np.testing.assert_array_equal(st, np_(s1))
    np.testing.assert_array_equal(aux, np_(a1)); np.testing.assert_array_equal(stats, np_(st1))
    a.filter_boundary = -1                                # no filter: every instance fitted
    assert lib.la3d_fit_annotations_host(C.byref(a)) == 0
    b0, s0, _ = la.fit_instances_poly(depth[0], (xy, ro, ir, H, W), K)
    np.testing.assert_array_equal(st, np_(s0)); np.testing.assert_array_equal(out, np_(b0))
    a.B = 0
    assert lib.la3d_fit_annotations_host(C.byref(a)) == 0
    a.B = B; a.depth = None
    assert lib.la3d_fit_annotations_host(C.byref(a)) != 0 and b"la3d_fit_annotations_host" in lib.la3d_last_error()
    a.depth = depth_d[0].data_ptr(); a.rle_counts = xy.ctypes.data          # both kinds at once
    assert lib.la3d_fit_annotations_host(C.byref(a)) != 0


def test_fit_instances_ex_projection_in_the_epilogue(la, monkeypatch):
    """la3d_fit_instances_ex: the records' 2-D boxes (bbox2D_proj | bbox2D_trunc, reference src/tools/combine_results.py:105-108,
    :238-252) written by the epilogue that writes the record must equal la3d_project_boxes on the finished records - for u8 planes
    (instance engine, and the split engine with its follow-up launch), run lengths, polygons with the fused filter - and the records
    must be those of the plain entry points."""
    rs = np.random.RandomState(8)
    B, H, W = 48, 480, 640
    P_img = 5
    depth = rs.uniform(0.5, 10, (P_img, H, W)).astype(np.float32)
    img = np.sort(rs.randint(0, P_img, B)).astype(np.int32)
    Ks = np.stack([np.array([[500.0 + 20 * i, 0, 320 + i], [0, 480.0 + 10 * i, 240 - i], [0, 0, 1]]) for i in range(P_img)])
    segs = [_random_segmentation(rs, W, H, k % 8) for k in range(B - 2)] + [[], [[10, 10, 14, 10, 14, 14, 10, 14]]]
    polys = la.pack_polygons(segs, H, W)
    masks = np_(la.poly_decode(polys))
    ground = np.array([[0.03, -0.97, 0.1, 1.1]] * B) + 0.02 * rs.randn(B, 4)
    ground[5] = [0.0, -1.0, 0.0, 0.0]                               # degenerate ground: rejected -> 8 NaNs
    size = (W, H)

    def check(res, ref_boxes, ref_status):
        np.testing.assert_array_equal(np_(res["status"]), np_(ref_status))
        np.testing.assert_array_equal(np_(res["boxes"]), np_(ref_boxes))
        want = np_(la.project_boxes(res["boxes"], Ks, size, image_index=img))
        np.testing.assert_array_equal(np_(res["boxes2d"]), want)
        bad = np_(res["status"]) != 0
        assert bad.any() and np.isnan(np_(res["boxes2d"])[bad]).all() and np.isfinite(np_(res["boxes2d"])[~bad]).all()

    monkeypatch.setattr(SCHED(), "engine", "instance")
    b0, s0, _ = la.fit_instances(depth, masks, Ks, ground=ground, image_index=img)
    check(la.fit_instances_ex(depth, Ks, masks=masks, ground=ground, image_index=img, image_size=size), b0, s0)
    monkeypatch.setattr(SCHED(), "engine", "split")
    b1, s1, _ = la.fit_instances(depth, masks, Ks, ground=ground, image_index=img)
    check(la.fit_instances_ex(depth, Ks, masks=masks, ground=ground, image_index=img, image_size=size), b1, s1)
    monkeypatch.setattr(SCHED(), "engine", None)
    rles = [O.rle_encode(m) for m in masks]
    b2, s2, _ = la.fit_instances_rle(depth, rles, Ks, ground=ground, image_index=img)
    check(la.fit_instances_ex(depth, Ks, rles=rles, ground=ground, image_index=img, image_size=size), b2, s2)
    b3, s3, _, st3 = la.fit_instances_poly(depth, polys, Ks, ground=ground, image_index=img, filter=True)
    res = la.fit_instances_ex(depth, Ks, polys=polys, ground=ground, image_index=img, image_size=size, filter=True)
    check(res, b3, s3)
    np.testing.assert_array_equal(np_(res["stats"]), np_(st3))
    assert (np_(res["status"]) == 6).any()
    with pytest.raises(ValueError, match="exactly one"):
        la.fit_instances_ex(depth, Ks, masks=masks, rles=rles)
    with pytest.raises(ValueError, match="fused filter"):
        la.fit_instances_ex(depth, Ks, masks=masks, filter=True)


def test_parts_with_disjoint_bounding_boxes_share_one_pass(la):
    """Instances of 2..8 parts whose vertex bounding boxes are pairwise disjoint take ONE fast pass (poly_to_bits): side-by-side and
    stacked parts, boxes that touch (adjacent columns / rows), parts partly outside the frame, a degenerate 2-point part among
    them, and the cases that must NOT join: 9 parts, overlapping boxes, a hole contour inside its outer contour.  All against the
    fillPoly restatement, decode and fit."""
    rs = np.random.RandomState(77)
    H, W = 240, 320

    def blob(x0, y0, x1, y1, n):
        ang = np.sort(rs.uniform(0, 2 * np.pi, n))
        rad = rs.uniform(0.55, 1.0, n)
        cx, cy, rx, ry = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2, (y1 - y0) / 2
        return np.clip(np.stack([cx + rx * rad * np.cos(ang), cy + ry * rad * np.sin(ang)], 1), [x0, y0], [x1, y1]).ravel().tolist()

    segs = []
    for nparts in range(2, 10):                                   # side by side, touching boxes (x1 of one = x0 - 1 of the next)
        xs = np.linspace(5, W - 5, nparts + 1).astype(int)
        segs.append([blob(xs[i], 20 + 7 * i, xs[i + 1] - 1, 200 - 5 * i, rs.randint(3, 40)) for i in range(nparts)])
    for nparts in (2, 3, 5, 8):                                   # stacked in y, touching rows
        ys = np.linspace(3, H - 3, nparts + 1).astype(int)
        segs.append([blob(30 + 5 * i, ys[i], 290 - 9 * i, ys[i + 1] - 1, rs.randint(3, 70)) for i in range(nparts)])
    segs.append([blob(-40, -30, 100, 90, 25), blob(150, 100, W + 60, H + 50, 31), [200, 10, 230, 40]])   # outside the frame + a 2-point part
    segs.append([blob(10, 10, 150, 200, 20), blob(140, 50, 300, 220, 20)])                               # overlapping boxes: not joint
    segs.append([blob(20, 20, 300, 220, 40), [100, 80, 220, 80, 220, 160, 100, 160]])                    # a "hole" contour: filled, not joint
    segs.append([[10, 10, 60, 10, 60, 60, 10, 60], [61, 10, 120, 10, 120, 60, 61, 60], [10, 61, 60, 61, 60, 120, 10, 120]])   # rectangles that touch
    polys = la.pack_polygons(segs, H, W)
    got = np_(la.poly_decode(polys))
    for i, seg in enumerate(segs):
        want, _ = P.create_boolean_mask_from_polygon((W, H), seg)
        assert np.array_equal(got[i], want), (i, len(seg), np.argwhere(got[i] != want)[:5])
    depth = rs.uniform(0.5, 10, (len(segs), H, W)).astype(np.float32)
    K = np.array([[250.0, 0, 160], [0, 250.0, 120], [0, 0, 1]])
    for eng in ("split", "instance"):
        with _engine(eng):
            b1, s1, a1 = la.fit_instances_poly(depth, polys, K)
            b2, s2, a2 = la.fit_instances(depth, got, K)
        np.testing.assert_array_equal(np_(s1), np_(s2))
        np.testing.assert_array_equal(np_(b1), np_(b2))
    np.testing.assert_array_equal(np_(la.mask_stats_poly(polys))[:, 0], got.reshape(len(segs), -1).sum(1))


def test_area_hint_orders_the_launch_without_the_estimate_pass(la, monkeypatch):
    """la3d_fit_args::area_hint: the launch order is built from the caller's areas instead of the estimate kernel.  Records are those of
    the unhinted call bit for bit - with the true areas, with useless ones (all equal, random, negative) and for every mask kind -
    because the order never shows in a record."""
    import torch

    import bench

    dev = torch.device("cuda", 0)
    monkeypatch.setattr(SCHED(), "engine", "instance")
    B = 1024
    depth, masks, K, _, rects = bench.make_inputs(B, dev, 3)
    ref = la.fit_instances_ex(depth, K, masks=masks)
    areas = masks.reshape(B, -1).sum(1, dtype=torch.int32)
    rs = np.random.RandomState(0)
    for hint in (areas, torch.zeros_like(areas), torch.as_tensor(rs.randint(-5, 400000, B).astype(np.int32), device=dev)):
        got = la.fit_instances_ex(depth, K, masks=masks, area_hint=hint)
        assert torch.equal(got["boxes"], ref["boxes"]) and torch.equal(got["status"], ref["status"]) and torch.equal(got["aux"], ref["aux"])
    rc, ro = bench.rect_rle(rects)
    rles = (rc, ro, bench.H, bench.W)
    r0 = la.fit_instances_ex(depth, K, rles=rles)
    r1 = la.fit_instances_ex(depth, K, rles=rles, area_hint=areas)
    assert torch.equal(r0["boxes"], r1["boxes"]) and torch.equal(r0["boxes"], ref["boxes"])
    with pytest.raises(ValueError, match="one entry per instance"):
        la.fit_instances_ex(depth, K, masks=masks, area_hint=areas[:5])


def test_g16_fillpoly_against_opencv_itself_on_gpu():
    """The HIP rasteriser against cv2.fillPoly's own masks (tests/golden/g16_fillpoly.npz, generated where cv2 exists): decoder and
    filter statistics.  Skipped while the fixture is absent."""
    import os

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_fillpoly.npz")
    if not os.path.exists(path):
        pytest.skip("g16_fillpoly.npz absent (needs cv2: tests/golden/make_golden_fillpoly.py)")
    from labelany3d_amd import poly_decode

    g = np.load(path, allow_pickle=False)
    for tag in ("a", "b", "c"):
        W, H = (int(v) for v in g[tag + "_size"])
        off, xy = g[tag + "_off"], g[tag + "_xy"]
        n = len(off) - 1
        want = np.unpackbits(g[tag + "_bits"], axis=1)[:, : H * W].reshape(n, H, W).astype(bool)
        polys = (xy.reshape(-1, 2).astype(np.int32), off.astype(np.int64), np.arange(n + 1, dtype=np.int64), H, W)
        got = poly_decode(polys).cpu().numpy()
        assert np.array_equal(got, want), (tag, int((got != want).sum()))

"""GPU tests of the mask-ingestion row: run-length decode, instance filters, and the box fit
fed with run lengths — against the oracle and the reference's own encoder / filter outputs (g8_masks.npz)."""
import numpy as np
import pytest

from .conftest import SCHED

from oracle import la3d_oracle as O

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


def _g8_rles(g):
    offs = np.concatenate([[0], np.cumsum(g["lens"])])
    return [{"size": [48, 64], "counts": g["counts"][offs[i]:offs[i + 1]].tolist()} for i in range(len(g["lens"]))]


def test_rle_decode_and_filters_vs_reference(la, golden):
    g = golden("g8_masks.npz")
    rles = _g8_rles(g)
    masks = la.rle_decode(rles)
    assert masks.dtype.is_floating_point is False and tuple(masks.shape) == g["masks"].shape
    np.testing.assert_array_equal(np_(masks), g["masks"].astype(bool))       # inverts the reference's encoder
    st = np_(la.mask_stats(masks))
    ref = g["ref_stats"]
    np.testing.assert_array_equal(st[:, :3], ref[:, :3])                       # area, rows, get_maximum_height
    np.testing.assert_array_equal(st[:, 3] >= 10, ref[:, 3].astype(bool))      # analyze_mask: is_truncated
    np.testing.assert_array_equal(st[:, 0] >= 100, ref[:, 4].astype(bool))     # analyze_mask: is_scaleable
    for from_rle in (True, False):
        keep = np_(la.keep_instances(la.mask_stats(masks), 48, from_rle))
        want = [O.keep_instance(O.mask_stats(m), 48, from_rle) for m in g["masks"]]
        np.testing.assert_array_equal(keep, want)
    # same stats straight from uint8 planes with arbitrary non-zero values
    st2 = np_(la.mask_stats((g["masks"] * 255).astype(np.uint8)))
    np.testing.assert_array_equal(st2, st)


@pytest.mark.parametrize("H,W", [(480, 640), (37, 53), (64, 96), (9, 4096), (270, 960), (1000, 1024)])
def test_rle_decode_random_masks(la, H, W):
    rs = np.random.RandomState(H + W)
    masks = np.zeros((12, H, W), bool)
    # runs that continue through many columns (toggle form: row 0 of every crossed column), ending mid-column / at a column end
    masks[9, :, 3:W - 5] = True
    masks[9, :H // 3, 3] = False
    masks[9, H // 2:, W - 6] = False
    masks[10, :, 1:2 + W // 3] = True
    masks[10, H - 1, 1] = False               # a one-pixel gap at the bottom of a column: two runs meet across the column end
    masks[11, H - 1, :] = True                # one-pixel runs at the bottom of every column
    masks[11, 0, ::2] = True                  # ... meeting one-pixel runs at the top of every other next column
    for i in range(5):
        h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = True
    masks[5] = rs.rand(H, W) < 0.5          # ~H*W/2 runs: many scan steps
    masks[6] = True                          # one run covering the frame
    masks[7] = False
    masks[8, :, W // 2] = True
    rles = [O.rle_encode(m) for m in masks]
    rles[3]["counts"] = O.rle_to_string(rles[3]["counts"])      # compressed-string form
    out = la.rle_decode(rles)
    np.testing.assert_array_equal(np_(out), masks)
    want = np.array([O.mask_stats(m) for m in masks])
    np.testing.assert_array_equal(np_(la.mask_stats(out)), want)


def test_fit_from_rle_equals_fit_from_planes(la):
    """The composed path fed with run lengths gives the same boxes as fed with the decoded u8 planes."""
    rs = np.random.RandomState(77)
    B, H, W = 24, 480, 640
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = np.zeros((B, H, W), bool)
    for i in range(B - 3):
        h, w = rs.randint(8, 301), rs.randint(8, 331)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = True
    vv, uu = np.mgrid[0:H, 0:W]
    masks[B - 3] = ((uu - 300) ** 2 / 9000.0 + (vv - 200) ** 2 / 4000.0) < 1.0
    masks[B - 2] = rs.rand(H, W) < 0.02
    # masks[B-1] stays empty
    ground = np.array([[0.02, -0.98, 0.1, 1.5]] * B) + 0.03 * rs.randn(B, 4)
    rles = [O.rle_encode(m) for m in masks]
    b_r, s_r, a_r = la.fit_instances_rle(depth, rles, K, ground=ground)
    ref = [O.fit_instance(depth[i], masks[i], K, ground[i]) for i in range(B)]
    assert np_(s_r).tolist() == [r[1] for r in ref]
    got = np_(b_r)
    for i, (rec, st, aux) in enumerate(ref):
        if st:
            assert np.isnan(got[i]).all()
            continue
        np.testing.assert_allclose(got[i, :15], rec[:15], rtol=0, atol=1e-9 * max(1, np.abs(rec[:6]).max()))
        np.testing.assert_allclose(got[i, 15:], rec[15:], rtol=0, atol=max(np.abs(rec[15:]).max(), 1) * 2.0 ** -10)
    np.testing.assert_array_equal(np_(a_r)[:, 2], masks.reshape(B, -1).sum(1))
    # bit-identical to the u8-plane entry point of the same engine
    import os
    SCHED().engine = "instance"
    try:
        b_p, s_p, a_p = la.fit_instances(depth, masks, K, ground=ground)
    finally:
        SCHED().engine = None
    assert np.array_equal(np_(s_p), np_(s_r))
    ok = np_(s_r) == 0      # same kernel after phase 0: equal (to the last bits today; asserted to rounding)
    np.testing.assert_allclose(np_(b_p)[ok][:, :15], got[ok][:, :15], rtol=1e-12, atol=1e-12)
    # reference-subsample mode through the RLE entry point
    counts = masks.reshape(B, -1).sum(1)
    np.random.seed(3)
    idx = la.draw_sample_idx(counts)
    b_s, s_s, _ = la.fit_instances_rle(depth, rles, K, ground=ground, sample_idx=idx)
    ref_s, st_s, _, _ = O.fit_instances(depth, masks, K[None].repeat(B, 0), ground=ground, sample_idx=idx)
    assert np_(s_s).tolist() == st_s.tolist()
    ok = st_s == 0
    np.testing.assert_allclose(np_(b_s)[ok][:, :15], ref_s[ok][:, :15], rtol=0, atol=1e-8)


def test_run_lengths_through_both_engines_ragged_frames(la, monkeypatch):
    """Run-length input on the split engine's decode front end (scan_bits_kernel) and on the instance engine, frames whose height
    is not a multiple of the 8-row tiles (rows past the frame must read as zeros in the last band), shared depth planes, a
    degenerate ground and an empty mask: each engine against the oracle, and bit for bit against ITS u8-plane entry."""
    rs = np.random.RandomState(123)
    for (H, W, B) in ((61, 64, 9), (100, 96, 17), (477, 640, 6), (8, 32, 3)):
        K = np.array([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
        P_img = max(1, B // 3)
        depth = rs.uniform(0.5, 10, (P_img, H, W)).astype(np.float32)
        img = rs.randint(0, P_img, B).astype(np.int32)
        masks = np.zeros((B, H, W), bool)
        for i in range(B - 1):
            h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            masks[i, r0:r0 + h, c0:c0 + w] = rs.rand(h, w) < (0.5 if i % 3 else 1.0)
        masks[0, H - 1, :] = True            # the last (ragged) tile row is in use
        ground = np.array([[0.02, -0.98, 0.1, 1.5]] * B) + 0.03 * rs.randn(B, 4)
        ground[1] = [0.0, -1.0, 0.0, 0.0]
        rles = [O.rle_encode(m) for m in masks]
        ref, rst, _, _ = O.fit_instances(depth[img], masks, K[None].repeat(B, 0), ground=ground)
        got = {}
        for eng in ("split", "instance"):
            monkeypatch.setattr(SCHED(), "engine", eng)
            b_r, s_r, a_r = la.fit_instances_rle(depth, rles, K, ground=ground, image_index=img)
            b_p, s_p, a_p = la.fit_instances(depth, masks, K, ground=ground, image_index=img)
            np.testing.assert_array_equal(np_(s_r), rst)
            np.testing.assert_array_equal(np_(s_r), np_(s_p))
            np.testing.assert_array_equal(np_(b_r), np_(b_p))
            np.testing.assert_array_equal(np_(a_r)[:, 2], masks.reshape(B, -1).sum(1))
            ok = rst == 0
            scale = np.maximum(1, np.abs(ref[ok][:, :6]).max(1))[:, None]
            assert (np.abs(np_(b_r)[ok][:, :15] - ref[ok][:, :15]) <= 1e-9 * scale).all(), (H, W, eng)
            got[eng] = np_(b_r)
        monkeypatch.setattr(SCHED(), "engine", None)
        ok = rst == 0
        np.testing.assert_allclose(got["split"][ok][:, :15], got["instance"][ok][:, :15], rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("H,W", [(1024, 1024), (720, 1280)])
def test_run_lengths_and_polygons_large_frames_both_engines(la, monkeypatch, H, W):
    """Frames whose bit image needs more than the 64 KiB default of dynamic LDS (1024x1024: 128 KiB image + decode scratch in the
    split engine's decode front end; the limit of run-length / polygon input is H*W <= 1048576): run lengths and polygon parts
    through both engines against the oracle and against the u8-plane entry of the same engine."""
    from oracle import poly_oracle as P
    rs = np.random.RandomState(H + W)
    B = 4
    K = np.array([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    segs = []
    for i in range(B):
        n = 5 + 3 * i
        ang = np.sort(rs.uniform(0, 2 * np.pi, n))
        cx, cy = rs.uniform(0.3 * W, 0.7 * W), rs.uniform(0.3 * H, 0.7 * H)
        rad = rs.uniform(0.4, 1.0, n)
        segs.append([np.stack([cx + 0.3 * W * rad * np.cos(ang), cy + 0.3 * H * rad * np.sin(ang)], 1).round().ravel().tolist()])
    segs[1].append([5, H - 40, 300, H - 40, 300, H - 1, 5, H - 1])           # a second part in the last rows of the frame
    polys = la.pack_polygons(segs, H, W)
    masks = np.stack([P.create_boolean_mask_from_polygon((W, H), sg)[0] for sg in segs])
    np.testing.assert_array_equal(np_(la.poly_decode(polys)).astype(bool), masks)
    rles = [O.rle_encode(m) for m in masks]
    ref, rst, _, _ = O.fit_instances(depth, masks, K[None].repeat(B, 0))
    assert (rst == 0).all()
    for eng in ("split", "instance"):
        monkeypatch.setattr(SCHED(), "engine", eng)
        b_u, s_u, _ = la.fit_instances(depth, masks, K)
        b_r, s_r, a_r = la.fit_instances_rle(depth, rles, K)
        b_p, s_p, a_p = la.fit_instances_poly(depth, polys, K)
        for b, st in ((b_r, s_r), (b_p, s_p)):
            np.testing.assert_array_equal(np_(st), rst)
            np.testing.assert_array_equal(np_(b), np_(b_u))
            scale = np.maximum(1, np.abs(ref[:, :6]).max(1))[:, None]
            assert (np.abs(np_(b)[:, :15] - ref[:, :15]) <= 1e-9 * scale).all(), (H, W, eng)
        np.testing.assert_array_equal(np_(a_r)[:, 2], masks.reshape(B, -1).sum(1))
        np.testing.assert_array_equal(np_(a_p)[:, 2], masks.reshape(B, -1).sum(1))
    monkeypatch.setattr(SCHED(), "engine", None)


def test_box_consumers_vs_reference(la, golden):
    """la3d_project_boxes / la3d_iou_matrix against the reference's project_to_2d, iou2D and hungarian_matching."""
    g = golden("g9_consumers.npz")
    out = np_(la.project_boxes(g["records"], g["K"], tuple(g["image_size"])))
    np.testing.assert_allclose(out, g["boxes2d"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np_(la.iou2d_matrix(g["iou_a"], g["iou_b"])), g["iou"], rtol=1e-12, atol=1e-14)
    got = la.hungarian_matching(g["iou_a"], g["iou_b"])
    want = g["matches"]
    assert [(i, j) for i, j, _ in got] == [(int(i), int(j)) for i, j, _ in want]
    np.testing.assert_allclose([v for _, _, v in got], want[:, 2], rtol=1e-12)
    # per-image K through image_index, and a failed box (NaN record) gives NaNs
    rec = g["records"].copy()
    rec[3] = np.nan
    Ks = np.stack([g["K"], g["K"] * [[1.1], [0.9], [1.0]]])
    idx = (np.arange(len(rec)) % 2).astype(np.int32)
    out = np_(la.project_boxes(rec, Ks, (640, 480), image_index=idx))
    ref = O.project_boxes(rec, Ks[idx], (640, 480))
    assert np.isnan(out[3]).all()
    ok = np.arange(len(rec)) != 3
    np.testing.assert_allclose(out[ok], ref[ok], rtol=1e-12, atol=1e-12)


def test_masked_ratio_median_vs_numpy(la):
    """the float32 arithmetic of reference src/util.py:480-486, restated with the same NumPy calls."""
    rs = np.random.RandomState(12)
    B, H, W = 9, 96, 128
    depth_map = rs.uniform(0.5, 10, (H, W)).astype(np.float32)
    render = (depth_map[None] * rs.uniform(0.3, 3.0, (B, 1, 1)) * (1 + 0.05 * rs.randn(B, H, W))).astype(np.float32)
    mask = rs.rand(B, H, W) < 0.3
    rmask = rs.rand(B, H, W) < 0.7
    mask[1] = False                                   # empty overlap -> the reference returns the identity transform
    mask[2] = False; mask[2, 5, 7] = True; rmask[2, 5, 7] = True           # one pixel
    mask[3] = False; mask[3, 10, 10:12] = True; rmask[3] = True            # two pixels: mean of both, in float32
    render[4, mask[4] & rmask[4]] *= -1               # negative ratios
    render[5, 20, 20] = 0.0; mask[5, 20, 20] = True; rmask[5, 20, 20] = True   # +inf ratio sorts last
    render[6, 30, 30] = np.nan; mask[6, 30, 30] = True; rmask[6, 30, 30] = True  # NaN -> NaN
    render[7][:] = render[7][0, 0]                    # heavy ties
    med, cnt = la.masked_ratio_median(depth_map, render, mask, rmask)
    med, cnt = np_(med), np_(cnt)
    for i in range(B):
        overlap = mask[i] & rmask[i]
        assert cnt[i] == overlap.sum()
        if not overlap.any():
            assert np.isnan(med[i])
            continue
        with np.errstate(all="ignore"):
            ratios = depth_map[overlap] / render[i][overlap]      # src/util.py:480-485
            want = np.median(ratios)                              # :486
        assert want.dtype == np.float32
        if np.isnan(want):
            assert np.isnan(med[i]), i
        else:
            assert med[i] == want, (i, med[i], want)              # exact: selection, not approximation
    # per-instance depth planes through image_index, no second mask
    dm = rs.uniform(1, 5, (3, H, W)).astype(np.float32)
    idx = rs.randint(0, 3, B).astype(np.int32)
    med2, _ = la.masked_ratio_median(dm, np.abs(render) + 0.1, mask, None, image_index=idx)
    for i in range(B):
        if mask[i].any():
            assert np_(med2)[i] == np.median(dm[idx[i]][mask[i]] / (np.abs(render[i]) + 0.1)[mask[i]]) or np.isnan(np_(med2)[i])


def test_unproject_matches_vs_numpy(la):
    """reference src/matching/matcher.py:70-91 restated with the same NumPy expressions."""
    rs = np.random.RandomState(8)
    depth = rs.uniform(1, 4, (512, 512)).astype(np.float32)
    depth[rs.rand(512, 512) < 0.2] = -1                       # the renderer's "no surface" value
    m1 = rs.uniform(0, 511.9, (300, 2))
    R = np.linalg.qr(rs.randn(3, 3))[0]
    T = rs.randn(3)
    pts, valid = la.unproject_matches(depth, m1, R=R, T=T)
    d_of = depth[m1[:, 1].astype(int), m1[:, 0].astype(int)]           # :71
    ok = d_of != -1                                                      # :72
    assert np.array_equal(np_(valid), ok)
    fx, fy, cx, cy = 560.44, 560.44, 256, 256                            # :78
    u = 512 - m1[ok][:, 0]; v = 512 - m1[ok][:, 1]                       # :79-80
    x = (u - cx) * d_of[ok] / fx; y = (v - cy) * d_of[ok] / fy; z = d_of[ok]   # :82-84
    p3 = np.stack((x, y, z), axis=-1)
    world = np.matmul(R, (p3.T - T.reshape(3, 1))).T                      # :88-90
    np.testing.assert_allclose(np_(pts)[ok], world, rtol=1e-13, atol=1e-13)
    assert np.isnan(np_(pts)[~ok]).all()
    pts2, _ = la.unproject_matches(depth, m1, flip=None)                  # plain pinhole, camera frame
    np.testing.assert_allclose(np_(pts2)[ok][:, 2], d_of[ok])


def test_correspondences_to_world_vs_reference(la):
    """SURVEY §8f-4 pinned: tests/golden/g11_matcher.npz holds what the reference's ImageMatcher.get_correspondences
    (src/matching/matcher.py:12-91) returned with its network / OpenCV calls replaced by stand-ins (make_golden_matcher.py): the
    border filter, crop offset, depth filter, flipped unprojection and world transform are the reference's own code."""
    import os
    import sys

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    try:
        from make_golden_matcher import make_depth
    finally:
        sys.path.remove(here)
    g = np.load(os.path.join(here, "g11_matcher.npz"))
    for case in ("a", "b", "none"):
        depth = make_depth(*g[f"{case}_depth_par"])
        pw, um0 = la.correspondences_to_world(g[f"{case}_m0"], g[f"{case}_m1"], g[f"{case}_shape0"], g[f"{case}_shape1"], depth,
                                              g[f"{case}_T"], g[f"{case}_R"])
        assert pw.shape == g[f"{case}_points_world"].shape and pw.dtype == np.float64
        np.testing.assert_array_equal(um0, g[f"{case}_matches0"])
        np.testing.assert_allclose(pw, g[f"{case}_points_world"], rtol=1e-13, atol=1e-13)
    assert len(g["a_points_world"]) > 100 and len(g["none_points_world"]) == 0


# ------------------------------------------------------------------------------------------
# the instance filter without a mask plane: stats from the run lengths, and the wide-load plane kernel
# ------------------------------------------------------------------------------------------
def _random_masks(rs, B, H, W):
    m = np.zeros((B, H, W), bool)
    for i in range(B):
        kind = i % 6
        if kind == 0:    # rectangle anywhere (often touching a border)
            h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            m[i, r0:r0 + h, c0:c0 + w] = True
        elif kind == 1:  # salt noise
            m[i] = rs.rand(H, W) < 0.02
        elif kind == 2:  # ellipse
            yy, xx = np.mgrid[0:H, 0:W]
            m[i] = ((yy - rs.uniform(0, H)) / rs.uniform(2, H / 2)) ** 2 + ((xx - rs.uniform(0, W)) / rs.uniform(2, W / 2)) ** 2 < 1
        elif kind == 3:  # full columns / full frame: runs longer than H
            m[i, :, rs.randint(0, W // 2):rs.randint(W // 2, W + 1)] = True
        elif kind == 4:  # two blobs with a gap of empty rows between them
            m[i, 1:3, 2:9] = True
            m[i, H - 5:H - 2, W - 7:W - 1] = True
        # kind 5: empty
    return m


@pytest.mark.parametrize("H,W,boundary", [(48, 64, 10), (480, 640, 10), (37, 53, 10), (30, 48, 20), (16, 16, 10), (8, 128, 3)])
def test_mask_stats_rle_and_planes_vs_oracle(la, H, W, boundary):
    import torch

    rs = np.random.RandomState(H * 1000 + W)
    B = 24
    m = _random_masks(rs, B, H, W)
    want = np.array([O.mask_stats(x, boundary) for x in m])
    # (a) straight from the runs
    rles = [O.rle_encode(x) for x in m]
    got = np_(la.mask_stats_rle(rles, boundary))
    np.testing.assert_array_equal(got, want)
    # (b) from u8 planes: 16-byte path when W % 16 == 0, byte path otherwise; any non-zero byte counts
    planes = torch.as_tensor(m.astype(np.uint8) * rs.randint(1, 256, (B, 1, 1)).astype(np.uint8), device="cuda")
    np.testing.assert_array_equal(np_(la.mask_stats(planes, boundary)), want)
    # (c) the keep rule on both
    for from_rle in (True, False):
        k1 = np_(la.keep_instances(la.mask_stats_rle(rles, boundary), H, from_rle))
        assert k1.tolist() == [O.keep_instance(tuple(s), H, from_rle) for s in want]


def test_mask_stats_rle_golden_and_compressed_strings(la, golden):
    g = golden("g8_masks.npz")
    rles = _g8_rles(g)
    want = np.array([O.mask_stats(m) for m in g["masks"]])
    np.testing.assert_array_equal(np_(la.mask_stats_rle(rles)), want)
    # compressed counts strings (what COCONut annotations carry) go through the host codec first
    strs = [{"size": r["size"], "counts": O.rle_to_string(r["counts"])} for r in rles]
    np.testing.assert_array_equal(np_(la.mask_stats_rle(strs)), want)
    # malformed input: runs beyond the frame are clipped exactly like the decoder clips them
    over = [{"size": [48, 64], "counts": [10, 48 * 64 + 500]}]
    dec = np_(la.rle_decode(over))[0]
    np.testing.assert_array_equal(np_(la.mask_stats_rle(over))[0], np.array(O.mask_stats(dec)))


def test_filter_annotations_matches_the_reference_rule(la):
    """filter_annotations == the RLE branch of read_bounding_boxes_segmentations (src/util.py:336-383): crowds skipped,
    height = rows holding a pixel, truncation and area rules — checked against the oracle's decode + stats + keep."""
    rs = np.random.RandomState(3)
    H, W = 120, 160
    m = _random_masks(rs, 30, H, W)
    annos = []
    for i, x in enumerate(m):
        rle = O.rle_encode(x)
        if i % 3 == 0:
            rle = {"size": rle["size"], "counts": O.rle_to_string(rle["counts"])}   # compressed string form
        annos.append({"iscrowd": int(i % 7 == 0), "bbox": [i, i, 10, 10], "category_id": 1 + i % 5, "segmentation": rle})
    bboxes, rles, kept, cats = la.filter_annotations(annos, (W, H))
    want = [i for i, x in enumerate(m) if not annos[i]["iscrowd"] and O.keep_instance(O.mask_stats(x), H, True)]
    assert kept.tolist() == want and len(want) > 0
    assert bboxes == [annos[i]["bbox"] for i in want] and cats == [annos[i]["category_id"] for i in want]
    np.testing.assert_array_equal(np_(la.rle_decode(rles)), m[want])
    # a polygon annotation takes the polygon branch (tests/test_gpu_poly.py); this small triangle fails the height / area rules
    out = la.filter_annotations([{"iscrowd": 0, "bbox": [0, 0, 1, 1], "category_id": 1, "segmentation": [[0, 0, 5, 0, 5, 5]]}], (W, H))
    assert out[0] == [] and out[1] == [] and out[2].tolist() == [] and out[3] == []


@pytest.mark.parametrize("H,W", [(48, 64), (30, 96), (37, 53), (480, 640)])
def test_rle_decode_fuzzed_run_lengths(la, H, W):
    """Run-length lists that no encoder would write: zero-length runs anywhere (ones-runs that touch, empty ones-runs), totals short of
    the frame (the rest is background) or beyond it (clipped, later runs ignored), very long runs crossing many columns, a single
    count, thousands of one-pixel runs - against pycocotools' rleDecode as restated in the oracle.  Word-aligned widths take the
    toggle decode, the others the per-pixel painter; the fit kernel's decode (same function) is checked through its pixel count."""
    rs = np.random.RandomState(H * 7 + W)
    cases = []
    for k in range(48 if H * W < 10000 else 12):
        n = int(rs.randint(1, 60 if k % 3 else 1500))
        kind = k % 6
        if kind == 0:
            c = rs.randint(0, 4, n)                                   # many zero-length runs
        elif kind == 1:
            c = rs.randint(0, 2 * H, n)                               # runs around one column long
        elif kind == 2:
            c = rs.randint(0, 5 * H, n)                               # runs over several columns
        elif kind == 3:
            c = np.where(rs.rand(n) < 0.3, 0, rs.randint(1, H * W // max(n, 1) + 2, n))
        elif kind == 4:
            c = rs.randint(0, H * W // 2, min(n, 6))                  # overshoots the frame quickly
        else:
            c = np.ones(n, np.int64)
        cases.append({"size": [H, W], "counts": [int(v) for v in c]})
    cases.append({"size": [H, W], "counts": [0, H * W]})
    cases.append({"size": [H, W], "counts": [H * W]})
    cases.append({"size": [H, W], "counts": [3, 2 * H * W]})
    cases.append({"size": [H, W], "counts": [H - 1, 1, 0, 1, H - 1, H + 1]})      # runs meeting across a column end
    want = np.stack([O.rle_decode(c["counts"], H, W) for c in cases])
    got = np_(la.rle_decode(cases))
    bad = [i for i in range(len(cases)) if not np.array_equal(got[i], want[i])]
    assert not bad, (bad[:5], cases[bad[0]]["counts"][:20])
    depth = np.full((H, W), 2.0, np.float32)
    K = np.array([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
    _, _, aux = la.fit_instances_rle(depth, cases, K)
    np.testing.assert_array_equal(np_(aux)[:, 2], want.reshape(len(cases), -1).sum(1))
    np.testing.assert_array_equal(np_(la.mask_stats_rle(cases))[:, 0], want.reshape(len(cases), -1).sum(1))

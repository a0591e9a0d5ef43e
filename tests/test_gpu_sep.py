"""The separable single pass of the instance engine (round 5, la3d_walks.hpp: sweep_sep).  For a camera without ground rotation and
without skew the x ray depends on the column only, the y ray on the row only and z is the depth itself, so the moments factor into
per-column sums and all six extents follow from per-column depth ranges + a per-pixel y: ONE walk over the depth instead of two
passes and a cull plan.  Checked here: parity with the CPU oracle (reference src/util_3dbox.py:106-178 composed with
src/util.py:52-75), agreement with the two-pass plain build to rounding, and every way OUT of the single pass (non-finite or
negative depth under the mask, skewed K, a ground vector, masks too large for the column arrays) giving exactly the two-pass
records."""
import numpy as np
import pytest

from oracle import la3d_oracle as O

from .conftest import SCHED
from .test_gpu_parity import K640, assert_records, np_, rect_masks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


def _blobs(rs, B, H, W):
    """irregular masks: unions of ellipses, rings, single pixels, thin lines, one empty"""
    vv, uu = np.mgrid[0:H, 0:W]
    m = np.zeros((B, H, W), bool)
    for i in range(B):
        for _ in range(rs.randint(1, 4)):
            cy, cx = rs.uniform(0, H), rs.uniform(0, W)
            ry, rx = rs.uniform(2, H / 3), rs.uniform(2, W / 3)
            e = ((vv - cy) / ry) ** 2 + ((uu - cx) / rx) ** 2
            m[i] |= (e < 1.0) & ~((e < 0.4) & (rs.rand() < 0.5))
    m[0] = False
    m[1] = False; m[1, H // 2, W // 2] = True
    m[2] = False; m[2, 5, :] = True                 # one row
    m[3] = False; m[3, :, W - 1] = True             # one column: the last
    m[4] = False; m[4, ::7, ::13] = True            # scattered single pixels
    return m


def _both(la, monkeypatch, *args, **kw):
    monkeypatch.setattr(SCHED(), "engine", "instance")
    monkeypatch.setattr(SCHED(), "build", None)
    one = tuple(np_(t) for t in la.fit_instances(*args, **kw))
    monkeypatch.setattr(SCHED(), "build", "plain")
    two = tuple(np_(t) for t in la.fit_instances(*args, **kw))
    monkeypatch.setattr(SCHED(), "build", None)
    return one, two


def _close(one, two, tag):
    b1, s1, a1 = one
    b2, s2, a2 = two
    assert s1.tolist() == s2.tolist(), tag
    ok = s1 == 0
    assert np.isnan(b1[~ok]).all() and np.isnan(b2[~ok]).all()
    np.testing.assert_array_equal(a1[:, 1:3], a2[:, 1:3])      # n_valid, n_masked
    for i in np.nonzero(ok)[0]:
        # the axis is conditioned like 1 / eigen-gap; both paths round differently at the 1e-16 level of the moments
        tol = 1e-11 * max(1.0, 1e-3 / max(a2[i, 3], 1e-12))
        np.testing.assert_allclose(b1[i, :15], b2[i, :15], rtol=tol, atol=tol * max(1.0, np.abs(b2[i, :6]).max()), err_msg=f"{tag}[{i}]")
        np.testing.assert_allclose(b1[i, 15:], b2[i, 15:], rtol=0, atol=max(np.abs(b2[i, 15:]).max(), 1.0) * 2.0 ** -10, err_msg=f"{tag}[{i}] vertices")


@pytest.mark.parametrize("H,W,B", [(480, 640, 40), (96, 128, 24), (60, 96, 12), (720, 1280, 6), (8, 32, 3)])
def test_single_pass_vs_oracle_and_two_pass(la, monkeypatch, H, W, B):
    rs = np.random.RandomState(H + W + B)
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = _blobs(rs, B, H, W) if B >= 6 else rect_masks(rs, B, H, W, hmax=H, wmax=W)
    # non-square pixels, principal point off the centre (still skew-free: the single pass applies)
    K = np.array([[0.8 * W, 0, 0.47 * W], [0, 0.9 * W, 0.55 * H], [0, 0, 1]])
    one, two = _both(la, monkeypatch, depth, masks, K)
    _close(one, two, f"{H}x{W}")
    ref, rst, _, nval = O.fit_instances(depth, masks, np.broadcast_to(K, (B, 3, 3)))
    assert one[1].tolist() == list(rst)
    # exact ties aside (single pixels / lines have gap 0 and are the documented don't-care), the oracle is matched at 1e-9
    ok = (one[1] == 0) & (one[2][:, 3] > 1e-6)
    assert_records(one[0][ok], ref[ok], f"single pass {H}x{W}", gap=one[2][ok, 3])
    np.testing.assert_array_equal(one[2][one[1] == 0, 1], nval[one[1] == 0])


def test_single_pass_shared_planes_smooth_depth_and_per_image_K(la, monkeypatch):
    """shared depth planes (configs 3 / 4), smooth depth (strong cancellation in the moments), a different skew-free K per image"""
    rs = np.random.RandomState(11)
    P, B, H, W = 4, 60, 480, 640
    vv, uu = np.mgrid[0:H, 0:W]
    depth = np.stack([(3.0 + 0.003 * (p + 1) * uu + 0.005 * vv + 0.02 * rs.randn(H, W)) for p in range(P)]).astype(np.float32)
    Ks = np.array([[[480.0 + 9 * p, 0, 318 + p], [0, 505.0 - 4 * p, 242 - p], [0, 0, 1]] for p in range(P)])
    img = rs.randint(0, P, B).astype(np.int32)
    masks = rect_masks(rs, B, H, W, 200, 260)
    one, two = _both(la, monkeypatch, depth, masks, Ks, image_index=img)
    _close(one, two, "shared")
    ref, rst, _, _ = O.fit_instances(depth, masks, Ks, depth_index=img)
    assert one[1].tolist() == list(rst)
    assert_records(one[0], ref, "single pass / shared planes", gap=one[2][:, 3])


def test_every_way_out_of_the_single_pass_gives_the_two_pass_records(la, monkeypatch):
    """NaN / inf / NEGATIVE depth under the mask (the per-column ranges rely on non-negative floats ordering like unsigned integers),
    a skewed K, a ground vector, a mask with more active tiles than leave room for the column arrays: the workgroup runs the general
    two-pass path and the records are the plain build's, bit for bit."""
    rs = np.random.RandomState(21)
    B, H, W = 16, 480, 640
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = rect_masks(rs, B, H, W)
    for i, bad in enumerate((np.nan, np.inf, -np.inf, -1.5, -0.0)):
        r, c = np.argwhere(masks[i])[rs.randint(int(masks[i].sum()))]
        depth[i, r, c] = bad
    masks[5] = True                      # full frame: 1200 active tiles
    masks[6] = True; masks[6, :40] = False
    one, two = _both(la, monkeypatch, depth, masks, K640)
    for i in (0, 1, 2, 3, 5):
        np.testing.assert_array_equal(one[0][i], two[0][i], err_msg=str(i))
    _close(one, two, "ways out")
    ref, rst, _, _ = O.fit_instances(depth, masks, np.broadcast_to(K640, (B, 3, 3)))
    assert one[1].tolist() == list(rst)
    assert_records(one[0], ref, "ways out vs oracle", gap=one[2][:, 3])
    # skewed K / ground vector: not separable - identical to the pinned two-pass build
    Ks = K640.copy(); Ks[0, 1] = 0.7
    one, two = _both(la, monkeypatch, depth[8:], masks[8:], Ks)
    np.testing.assert_array_equal(one[0], two[0])
    ground = np.array([[0.05, -0.97, 0.1, 1.2]] * 8) + 0.02 * rs.randn(8, 4)
    ground[::3, 0] = np.nan              # every third: no ground -> the single pass, next to grounded instances in one launch
    one, two = _both(la, monkeypatch, depth[8:], masks[8:], K640, ground=ground)
    grounded = ~np.isnan(ground[:, 0])
    np.testing.assert_array_equal(one[0][grounded], two[0][grounded])
    _close(one, two, "mixed ground")
    gl = [None if np.isnan(g[0]) else g for g in ground]
    ref = np.array([O.fit_instance(depth[8 + i], masks[8 + i], K640, gl[i])[0] for i in range(8)])
    assert_records(one[0], ref, "mixed ground vs oracle", gap=one[2][:, 3])


def test_single_pass_run_lengths_and_polygons_equal_planes(la, monkeypatch):
    """the three mask formats decode to the same bit image and take the same single pass: bit-identical records"""
    from labelany3d_amd import fit_instances_poly, fit_instances_rle, pack_polygons, poly_decode

    rs = np.random.RandomState(31)
    B, H, W = 300, 480, 640       # (above the split engine's range: the instance engine takes all three)
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    segs = []
    for i in range(B):
        n = rs.randint(3, 9)
        cx, cy, r = rs.uniform(60, W - 60), rs.uniform(60, H - 60), rs.uniform(10, 120)
        ang = np.sort(rs.uniform(0, 2 * np.pi, n))
        segs.append([np.stack([cx + r * np.cos(ang), cy + 0.7 * r * np.sin(ang)], 1).reshape(-1).tolist()])
    packed = pack_polygons(segs, H, W)
    planes = np_(poly_decode(packed)).astype(bool)
    monkeypatch.setattr(SCHED(), "engine", "instance")
    b0, s0, _ = la.fit_instances(depth, planes, K640)
    b1, s1, _ = fit_instances_poly(depth, packed, K640)
    b2, s2, _ = fit_instances_rle(depth, [O.rle_encode(m) for m in planes], K640)
    assert np_(s0).tolist() == np_(s1).tolist() == np_(s2).tolist()
    np.testing.assert_array_equal(np_(b0), np_(b1))
    np.testing.assert_array_equal(np_(b0), np_(b2))
    ref, rst, _, _ = O.fit_instances(depth[:40], planes[:40], np.broadcast_to(K640, (40, 3, 3)))
    assert np_(s0)[:40].tolist() == list(rst)
    assert_records(np_(b0)[:40], ref, "formats")

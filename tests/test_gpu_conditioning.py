"""Ill-conditioned moments (round 6; found by profiles/r06/fuzz_engines.py).  The kernels form the footprint's covariance from RAW
sums.  The reference (scikit-learn PCA(2), src/util_3dbox.py:181-186) centres the points and takes an SVD for fewer than 20 points
- exact however far the cloud is - and works from raw sums itself from 20 points on ('covariance_eigh').  For a cloud whose spread
in the x'z' plane is tiny against its distance from the camera (a one-pixel column, a sliver a millimetre wide 48 m away) raw sums
lose the axis to cancellation: axis_from_sums detects it (kappa = raw second moment / variance along the axis > 2^17) and the
engines run the moments a second time about the mean of the first pass - instance engine (tiled, row-linear and subsample forms),
band, row and split engines, the point-cloud kernels.  Clouds with no spread at all (the reference's own axis is
rounding noise) report gap = 0 everywhere.  Checked against the CPU oracle, which follows the reference's two solvers: at the
stated 1e-9 for n < 20; for n >= 20 the REFERENCE's own rounding noise (test_gpu_parity.reference_axis_noise) is allowed on top -
and the kernels' axis is held to 1e-9 of a long-double evaluation of the centred moments, i.e. they are the exact ones."""
import numpy as np
import pytest

from oracle import la3d_oracle as O

from .conftest import SCHED
from .test_gpu_parity import assert_records, np_, reference_axis_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


def thin_far_case(rs, B, H, W, skew=3.0, ripple=0.0):
    """Instances whose footprint is a sliver: one-pixel columns (x varies with the row only through the skew, or not at all),
    2 x 3 blobs, short rows - on a plane tens of metres away."""
    depth = np.full((H, W), 48.0, np.float32)
    if ripple:
        yy, xx = np.mgrid[:H, :W]
        depth = (48.0 + ripple * np.sin(yy / 7.0) * np.cos(xx / 5.0)).astype(np.float32)
    masks = np.zeros((B, H, W), bool)
    for n in range(B):
        kind = n % 4
        c = rs.randint(2, W - 4)
        r0 = rs.randint(0, H // 2)
        if kind in (0, 1):
            masks[n, r0:r0 + rs.randint(2, H // 2), c] = True            # one column
        elif kind == 2:
            masks[n, r0:r0 + 2, c:c + 3] = True                           # 2 x 3 blob
        else:
            masks[n, r0, c:c + rs.randint(2, 4)] = True                   # two or three pixels of a row
    K = np.array([[900.0, skew, W / 2 + 3.3], [0, 880.0, H / 2 - 1.7], [0, 0, 1]])
    return depth, masks, K


def run_engines(la, monkeypatch, engines, depth, masks, K, ground=None, sample_idx=None):
    out = {}
    for eng in engines:
        monkeypatch.setattr(SCHED(), "engine", eng)
        b, s, a = la.fit_instances(depth, masks, K, ground=ground, sample_idx=sample_idx)
        out[eng] = (np_(b), np_(s), np_(a))
    monkeypatch.setattr(SCHED(), "engine", None)
    return out


def check(got, ref, rst, kappa, tag, need_resolved):
    b, s, a = got
    assert s.tolist() == list(rst), tag
    ok = (s == 0) & (a[:, 3] >= 1e-9)
    assert int(ok.sum()) >= need_resolved, f"{tag}: only {int(ok.sum())} records resolved (gap > 0)"
    assert_records(b[ok], ref[ok], tag, gap=a[ok, 3], noise=reference_axis_noise(kappa[ok], a[ok, 1], a[ok, 3]))
    return ok


def exact_yaw(depth, mask, K, ground=None):
    """The first principal axis of the instance's (x', z') footprint from CENTRED moments in long double: what both of the
    reference's solvers approximate."""
    pts = O.depth_to_points(np.asarray(depth)[None], np.asarray(K, float))[mask]
    rot = np.dot(pts, O.ground_rotation(ground))
    rot = rot[np.isfinite(rot).all(1)]
    x, z = rot[:, 0].astype(np.longdouble), rot[:, 2].astype(np.longdouble)
    x, z = x - x.mean(), z - z.mean()
    return O.yaw_from_cov(float((x * x).sum()), float((x * z).sum()), float((z * z).sum()), len(x))


@pytest.mark.parametrize("grounded", [False, True])
def test_thin_far_instances_follow_the_reference_axis(la, monkeypatch, grounded):
    rs = np.random.RandomState(7)
    B, H, W = 40, 96, 160
    depth, masks, K = thin_far_case(rs, B, H, W, skew=3.0, ripple=2e-3)
    ground = (np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.02 * rs.randn(B, 4)) if grounded else None
    ref, rst, _, _, kap = O.fit_instances(depth, masks, K, ground=None if ground is None else list(ground), return_kappa=True)
    assert (kap > 2.0 ** 17).sum() >= B - 2   # (the case is ill-conditioned for raw sums: above the kernels' threshold)
    engines = [None, "instance", "band", "split"] + ([] if grounded else ["rows", "rows2"])
    got = run_engines(la, monkeypatch, engines, depth, masks, K, ground)
    yex = np.array([exact_yaw(depth, masks[n], K, None if ground is None else ground[n]) for n in range(B)])
    for eng in engines:
        ok = check(got[eng], ref, rst, kap, f"thin/{eng}/ground={grounded}", need_resolved=B - 2)
        a = got[eng][2]
        dy = np.abs((a[ok, 0] - yex[ok] + np.pi / 2) % np.pi - np.pi / 2)   # (an axis: modulo pi)
        assert (dy <= 1e-9 / np.minimum(a[ok, 3], 1.0)).all(), f"{eng}: yaw off the exact axis by {dy.max():.2e}"


def test_no_spread_at_all_reports_gap_zero(la, monkeypatch):
    """One-pixel columns on a constant plane through a skew-free camera: every point of an instance has the same (x, z) - the
    reference's axis is the SVD of rounding noise.  Status, counts, centre and dims still agree; the gap says 'unresolved'."""
    rs = np.random.RandomState(3)
    B, H, W = 16, 64, 96
    depth, masks, K = thin_far_case(rs, B, H, W, skew=0.0)
    masks[:] = False
    for n in range(B):
        masks[n, 3 + n:40 + n, 5 * n + 4] = True
    ref, rst, _, _ = O.fit_instances(depth, masks, K)
    for eng, g in run_engines(la, monkeypatch, [None, "instance", "band", "rows", "split"], depth, masks, K).items():
        b, s, a = g
        assert s.tolist() == list(rst), eng
        assert (a[:, 3] == 0).all(), f"{eng}: gaps {a[:, 3]}"
        scale = np.maximum(np.abs(ref[:, :6]).max(1), 1.0)
        # dims: the y extent is exact; the x / z extents are zero up to the rounding of an arbitrary rotation of identical points
        np.testing.assert_allclose(b[:, 4], ref[:, 4], rtol=0, atol=1e-9 * scale.max())
        assert np.abs(b[:, [3, 5]]).max() < 1e-9 * scale.max() and np.abs(ref[:, [3, 5]]).max() < 1e-9 * scale.max()


def test_thin_far_row_linear_and_subsample_forms(la, monkeypatch):
    """The instance engine's other forms: a frame of odd width in one call of one instance (row-linear walk) and the
    reference-subsample mode (more than 500 mask pixels: the 500 drawn points are the cloud)."""
    rs = np.random.RandomState(11)
    # row-linear: W % 32 != 0 and B = 1
    H, W = 120, 100
    depth = (48.0 + 1e-3 * rs.rand(H, W)).astype(np.float32)
    K = np.array([[900.0, 2.0, 51.0], [0, 880.0, 60.0], [0, 0, 1]])
    for c in (7, 50, 93):
        m = np.zeros((1, H, W), bool)
        m[0, 10:90, c] = True
        ref, rst, _, _, kap = O.fit_instances(depth, m, K, return_kappa=True)
        b, s, a = la.fit_instances(depth, m, K)
        check((np_(b), np_(s), np_(a)), ref, rst, kap, f"row-linear c={c}", need_resolved=1)
    # subsample mode: two columns x 300 rows = 600 mask pixels
    H, W, B = 304, 64, 6
    depth = (48.0 + 1e-3 * rs.rand(H, W)).astype(np.float32)
    K = np.array([[900.0, 2.0, 30.0], [0, 880.0, 150.0], [0, 0, 1]])
    masks = np.zeros((B, H, W), bool)
    for n in range(B):
        masks[n, 2:302, 8 * n + 3:8 * n + 5] = True
    sidx = np.stack([rs.randint(0, 600, 500) for _ in range(B)]).astype(np.int32)
    ref, rst, _, _, kap = O.fit_instances(depth, masks, K, sample_idx=sidx, return_kappa=True)
    got = run_engines(la, monkeypatch, [None, "instance"], depth, masks, K, sample_idx=sidx)
    for eng, g in got.items():
        check(g, ref, rst, kap, f"subsample/{eng}", need_resolved=B)


def test_well_conditioned_records_are_untouched(la, monkeypatch):
    """The detection threshold (kappa = 2^17: spread below ~1/360 of the distance) is far from ordinary instances: a config-2 style
    batch reports the gaps it always reported (none zero) - the second pass is not taken."""
    rs = np.random.RandomState(5)
    B, H, W = 32, 480, 640
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = np.zeros((B, H, W), bool)
    for n in range(B):
        h, w = rs.randint(8, 300), rs.randint(8, 330)
        r, c = rs.randint(0, H - h), rs.randint(0, W - w)
        masks[n, r:r + h, c:c + w] = True
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    b, s, a = (np_(t) for t in la.fit_instances(depth, masks, K))
    assert (s == 0).all() and (a[:, 3] > 1e-3).all()


def test_thin_far_point_clouds(la):
    """The point-cloud entry (la3d_fit_points, every launch form) and the scalar drop-in (la3d_estimate_bbox_host): clouds a few
    millimetres long tens of metres from the origin of their frame take the second moments pass too."""
    import contextlib
    import io

    from labelany3d_amd import util_3dbox as U

    rs = np.random.RandomState(2)
    clouds = []
    for n in (2, 5, 19, 20, 64, 500, 700, 3000):
        c = rs.randn(n, 3) * [10 ** rs.uniform(-3, -2), 0.3, 10 ** rs.uniform(-4, -3)]   # (millimetres long, a tenth of that wide)
        clouds.append(c @ O.rotate_y(rs.uniform(-3, 3)).T + [rs.uniform(-30, 30), 0, rs.uniform(30, 80)])
    ground = np.array([[0.05, -0.97, 0.1, 1.2]] * len(clouds)) + 0.02 * rs.randn(len(clouds), 4)
    for g in (None, ground):
        refs = [O.fit_points(c, None if g is None else g[i], False, "pca") for i, c in enumerate(clouds)]
        rec = np.stack([r[0] for r in refs]); rst = [r[1] for r in refs]
        kap = np.array([r[2]["kappa"] for r in refs])
        assert (kap > 2.0 ** 17).all()
        for kw in (dict(), dict(small_clouds=True), dict(small_clouds=False)):
            b, s, a = (np_(t) for t in la.fit_points(clouds, g, None, "pca", **kw))
            check((b, s, a), rec, rst, kap, f"points {kw} ground={g is not None}", need_resolved=len(clouds))
        for i, c in enumerate(clouds[:6]):   # the scalar drop-in: one host-pointer call per cloud (no subsampling: <= 500 rows)
            with contextlib.redirect_stdout(io.StringIO()):
                r1, a1 = U._fit_one(c, None if g is None else g[i], "pca", subsample=False)
            check((r1[None], np.zeros(1, np.int32), a1[None]), rec[i:i + 1], [0], kap[i:i + 1], f"scalar drop-in cloud {i}", need_resolved=1)

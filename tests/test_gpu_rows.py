"""The row engine (round 5, la3d_rows.hip: fit_rows_kernel + merge_rows_kernel): small batches of u8 planes without a ground array are
fitted by up to sixteen workgroups per instance, one per band of rows - the separable single pass split by rows, the bands' partial
sums / extents / per-column depth ranges merged by a second short launch.  Checked here: parity with the CPU oracle (reference
src/util_3dbox.py:106-178 composed with src/util.py:52-75), agreement with the instance engine to rounding, every status, the ways
out (non-finite / negative depth, skewed K: the merge workgroup's generic walk), shared depth planes, the records' 2-D boxes, and
which calls take it by default."""
import numpy as np
import pytest

from oracle import la3d_oracle as O

from .conftest import SCHED
from .test_gpu_parity import K640, assert_records, np_, rect_masks
from .test_gpu_sep import _blobs, _close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


def _engines(la, monkeypatch, *args, **kw):
    monkeypatch.setattr(SCHED(), "build", None)
    monkeypatch.setattr(SCHED(), "engine", "rows")
    rows = tuple(np_(t) for t in la.fit_instances(*args, **kw))
    monkeypatch.setattr(SCHED(), "engine", "instance")
    inst = tuple(np_(t) for t in la.fit_instances(*args, **kw))
    monkeypatch.setattr(SCHED(), "engine", None)
    return rows, inst


@pytest.mark.parametrize("H,W,B", [(480, 640, 1), (480, 640, 40), (480, 640, 300), (96, 128, 24), (720, 1280, 6), (16, 32, 7), (64, 96, 512),
                                   (427, 640, 20), (426, 640, 3), (101, 96, 9), (17, 64, 5)])   # (heights that are not a multiple of 8: a partial last tile row)
def test_row_engine_vs_oracle_and_instance_engine(la, monkeypatch, H, W, B):
    rs = np.random.RandomState(H + W + B)
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = _blobs(rs, B, H, W) if B >= 6 else rect_masks(rs, B, H, W, hmax=H, wmax=W)
    K = np.array([[0.8 * W, 0, 0.47 * W], [0, 0.9 * W, 0.55 * H], [0, 0, 1]])
    rows, inst = _engines(la, monkeypatch, depth, masks, K)
    _close(rows, inst, f"rows {H}x{W} B={B}")
    n = min(B, 48)                                             # (the oracle takes its time)
    ref, rst, _, nval = O.fit_instances(depth[:n], masks[:n], np.broadcast_to(K, (n, 3, 3)))
    assert rows[1][:n].tolist() == list(rst)
    ok = (rows[1][:n] == 0) & (rows[2][:n, 3] > 1e-6)
    assert_records(rows[0][:n][ok], ref[ok], f"row engine {H}x{W}", gap=rows[2][:n][ok, 3])
    np.testing.assert_array_equal(rows[2][:n][rows[1][:n] == 0, 1], nval[rows[1][:n] == 0])


def test_row_engine_ways_out_and_masks_of_255(la, monkeypatch):
    """NaN / inf / negative depth under the mask, a skewed K: the band raises its flag, the merge workgroup walks the whole plane
    with the generic two passes - the records of the pinned two-pass build bit for bit; 255-valued masks take the general byte test."""
    rs = np.random.RandomState(5)
    B, H, W = 12, 480, 640
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = rect_masks(rs, B, H, W)
    for i, bad in enumerate((np.nan, np.inf, -np.inf, -1.5, -0.0)):
        r, c = np.argwhere(masks[i])[rs.randint(int(masks[i].sum()))]
        depth[i, r, c] = bad
    masks[5] = True
    rows, inst = _engines(la, monkeypatch, depth, masks, K640)
    _close(rows, inst, "ways out")
    ref, rst, _, _ = O.fit_instances(depth, masks, np.broadcast_to(K640, (B, 3, 3)))
    assert rows[1].tolist() == list(rst)
    assert_records(rows[0], ref, "row engine ways out vs oracle", gap=rows[2][:, 3])
    Ks = K640.copy(); Ks[0, 1] = 0.7                             # skew: no band can take the single pass
    rows, inst = _engines(la, monkeypatch, depth[6:], masks[6:], Ks)
    _close(rows, inst, "skew")
    ref, rst, _, _ = O.fit_instances(depth[6:], masks[6:], np.broadcast_to(Ks, (B - 6, 3, 3)))
    assert_records(rows[0], ref, "row engine skew vs oracle", gap=rows[2][:, 3])
    m255 = masks[6:].astype(np.uint8) * 255
    r255, _ = _engines(la, monkeypatch, depth[6:], m255, K640)
    r1, _ = _engines(la, monkeypatch, depth[6:], masks[6:], K640)
    for a, b in zip(r255, r1):
        np.testing.assert_array_equal(a, b)


def test_row_engine_shared_planes_2d_boxes_and_default_dispatch(la, monkeypatch):
    import torch

    rs = np.random.RandomState(9)
    P, B, H, W = 3, 20, 480, 640
    vv, uu = np.mgrid[0:H, 0:W]
    depth = np.stack([(3.0 + 0.003 * (p + 1) * uu + 0.005 * vv + 0.02 * rs.randn(H, W)) for p in range(P)]).astype(np.float32)
    Ks = np.array([[[480.0 + 9 * p, 0, 318 + p], [0, 505.0 - 4 * p, 242 - p], [0, 0, 1]] for p in range(P)])
    img = rs.randint(0, P, B).astype(np.int32)
    masks = rect_masks(rs, B, H, W, 200, 260)
    rows, inst = _engines(la, monkeypatch, depth, masks, Ks, image_index=img)
    _close(rows, inst, "shared planes")
    ref, rst, _, _ = O.fit_instances(depth, masks, Ks, depth_index=img)
    assert rows[1].tolist() == list(rst)
    assert_records(rows[0], ref, "row engine / shared planes", gap=rows[2][:, 3])
    # the records' 2-D boxes from the merge launch's epilogue = la3d_project_boxes on the records
    monkeypatch.setattr(SCHED(), "engine", "rows")
    res = la.fit_instances_ex(depth, Ks, masks=masks, image_index=img, image_size=(W, H))
    monkeypatch.setattr(SCHED(), "engine", None)
    want = la.project_boxes(res["boxes"], Ks, (W, H), image_index=img)
    np.testing.assert_array_equal(np_(res["boxes2d"]), np_(want))
    np.testing.assert_array_equal(np_(res["boxes"]), rows[0])
    # default dispatch: an un-grounded small batch of u8 planes takes the row engine (records equal the pinned engine's bit for bit);
    # with a ground array it does not
    dflt = tuple(np_(t) for t in la.fit_instances(depth, masks, Ks, image_index=img))
    for a, b in zip(dflt, rows):
        np.testing.assert_array_equal(a, b)
    ground = np.array([[0.05, -0.97, 0.1, 1.2]] * B)
    g_default = np_(la.fit_instances(depth, masks, Ks, image_index=img, ground=ground)[0])
    monkeypatch.setattr(SCHED(), "engine", "rows")               # pinned, but not applicable: falls back, same records
    g_rows = np_(la.fit_instances(depth, masks, Ks, image_index=img, ground=ground)[0])
    monkeypatch.setattr(SCHED(), "engine", None)
    np.testing.assert_array_equal(g_default, g_rows)
    # two calls on two streams, each with its own workspace (InstanceFitter): no shared state
    f1, f2 = la.InstanceFitter(B, H, W, torch.device("cuda", 0)), la.InstanceFitter(B, H, W, torch.device("cuda", 0))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d_t, m_t, k_t, i_t = (torch.as_tensor(x, device="cuda") for x in (depth, masks.view(np.uint8), Ks, img))
    torch.cuda.synchronize()
    for _ in range(10):
        f1.run(d_t, m_t, k_t, image_index=i_t, stream=s1)
        f2.run(d_t, m_t, k_t, image_index=i_t, stream=s2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(np_(f1.boxes[0]), rows[0])
    np.testing.assert_array_equal(np_(f2.boxes[0]), rows[0])


def test_row_engine_odd_band_splits(la, monkeypatch):
    """Frames whose tile rows do not divide evenly among the bands (a shorter last band, bands of two tile rows, one tile column):
    the row engine against the instance engine (to rounding) on every shape, against the oracle on the first instances of each."""
    rs = np.random.RandomState(77)
    for ty, tx, B in ((3, 1, 5), (5, 2, 9), (7, 3, 33), (9, 5, 3), (11, 8, 2), (13, 2, 70), (17, 1, 12), (23, 3, 4), (31, 2, 1), (60, 20, 2)):
        H, W = ty * 8, tx * 32
        depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
        masks = rs.rand(B, H, W) < rs.uniform(0.02, 0.9, (B, 1, 1))
        for i in range(0, B, 3):                                   # every third: a rectangle that may straddle the band boundaries
            h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            masks[i] = False; masks[i, r0:r0 + h, c0:c0 + w] = True
        K = np.array([[0.9 * W, 0, 0.5 * W + 1.5], [0, 0.8 * W, 0.45 * H], [0, 0, 1]])
        rows, inst = _engines(la, monkeypatch, depth, masks, K)
        _close(rows, inst, f"{H}x{W} B={B}")
        n = min(B, 6)
        ref, rst, _, nval = O.fit_instances(depth[:n], masks[:n], np.broadcast_to(K, (n, 3, 3)))
        assert rows[1][:n].tolist() == list(rst)
        ok = (rows[1][:n] == 0) & (rows[2][:n, 3] > 1e-6)
        assert_records(rows[0][:n][ok], ref[ok], f"row engine {H}x{W}", gap=rows[2][:n][ok, 3])


@pytest.mark.parametrize("B", [1, 7, 40, 256, 384, 512])
def test_row_engine_one_launch_equals_two_launches(la, monkeypatch, B):
    """Round 6: the row engine is ONE launch - the band workgroup that arrives last at its instance's counter merges the instance
    (agent-scope stores / loads through the workspace, tagged arrival words nobody clears).  The merge is the same code on the same
    operands in the same order as the merge launch of the two-launch form ("rows2", what a graph-captured call takes): the records
    must be equal bit for bit, call after call on the same workspace, and - B = 256 / 384 / 512, the batches it newly takes over -
    agree with the instance engine to rounding and with the oracle."""
    import torch

    rs = np.random.RandomState(1000 + B)
    H, W = 480, 640
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    masks = rect_masks(rs, B, H, W)
    if B >= 7:
        masks[3] = False                      # an empty instance
        r, c = np.argwhere(masks[5])[0]
        depth[5, r, c] = np.nan               # a band that raises its flag: the merging workgroup walks the plane itself
    d_t, m_t, k_t = (torch.as_tensor(x, device="cuda") for x in (depth, masks.view(np.uint8), K640))
    f = la.InstanceFitter(B, H, W, torch.device("cuda", 0))
    out = {}
    for eng in ("rows2", "rows", "instance"):
        monkeypatch.setattr(SCHED(), "engine", eng)
        for _ in range(3):                    # the same workspace call after call: every call carries its own tag
            f.run(d_t, m_t, k_t)
        torch.cuda.synchronize()
        out[eng] = (np_(f.boxes[0]).copy(), np_(f.status[0]).copy(), np_(f.aux[0]).copy())
    monkeypatch.setattr(SCHED(), "engine", None)
    for a, b in zip(out["rows"], out["rows2"]):
        np.testing.assert_array_equal(a, b)
    _close(out["rows"], out["instance"], f"one-launch rows vs instance, B={B}")
    n = min(B, 24)
    ref, rst, _, nval = O.fit_instances(depth[:n], masks[:n], np.broadcast_to(K640, (n, 3, 3)))
    assert out["rows"][1][:n].tolist() == list(rst)
    ok = (out["rows"][1][:n] == 0) & (out["rows"][2][:n, 3] > 1e-6)
    assert_records(out["rows"][0][:n][ok], ref[ok], f"one-launch rows B={B}", gap=out["rows"][2][:n][ok, 3])

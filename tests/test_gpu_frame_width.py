"""Frames whose width is not a multiple of 32 (COCO: 427, 500, 375, 333 ...) with run-length / polygon masks: the depth rows are
padded to the next multiple of 32 and the fit runs its tiled / single-pass forms on the padded frame, ``la3d_fit_args::frame_width``
saying where the image ends (polygon sides are clipped to the IMAGE, the filter's right border is the image's).  Checked against
the CPU oracle on the unpadded frame (reference src/util_3dbox.py:106-178 composed with src/util.py:52-75, masks decoded by the
reference's rules: src/util.py:364-367, :386-400)."""
import ctypes as C

import numpy as np
import pytest

from oracle import la3d_oracle as O
from oracle import poly_oracle as P

from .test_gpu_parity import assert_records, np_
from .test_gpu_sep import _blobs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


def _polys(rs, B, H, W):
    """irregular polygons, some with vertices beyond the right / bottom / left border (clipLine at the IMAGE width)"""
    segs = []
    for i in range(B):
        n = rs.randint(3, 40)
        ang = np.sort(rs.uniform(0, 2 * np.pi, n))
        cx, cy = rs.uniform(0.2 * W, 1.05 * W), rs.uniform(0.2 * H, 0.9 * H)
        rad = rs.uniform(0.3, 1.0, n)
        xy = np.stack([cx + 0.45 * W * rad * np.cos(ang), cy + 0.4 * H * rad * np.sin(ang)], 1)
        parts = [np.trunc(xy).astype(int).ravel().tolist()]
        if i % 4 == 0:   # a second, small part at the right border
            parts.append([W - 9, 10, W + 40, 10, W + 40, 60, W - 9, 60])
        segs.append(parts)
    return segs


@pytest.mark.parametrize("H,W", [(640, 427), (375, 500), (500, 333), (117, 75)])
def test_odd_widths_run_lengths_and_polygons_vs_oracle(la, H, W):
    rs = np.random.RandomState(H * 7 + W)
    B = 26
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    K = np.array([[0.9 * W, 0, 0.48 * W], [0, 0.95 * W, 0.52 * H], [0, 0, 1]])
    ground = np.array([[0.03, -0.97, 0.1, 1.1]] * B) + 0.02 * rs.randn(B, 4)
    ground[::2, 0] = np.nan                                      # every other instance un-grounded (the single pass)
    # run lengths
    masks = _blobs(rs, B, H, W)
    rles = [O.rle_encode(m) for m in masks]
    gl = [None if np.isnan(g[0]) else g for g in ground]
    for gr in (None, ground):
        b, s, a = (np_(t) for t in la.fit_instances_rle(depth, rles, K, ground=gr))
        ref = [O.fit_instance(depth[i], masks[i], K, None if gr is None else gl[i]) for i in range(B)]
        assert s.tolist() == [r[1] for r in ref]
        ok = (s == 0) & (a[:, 3] > 1e-6)
        assert_records(b[ok], np.array([r[0] for r in ref])[ok], f"rle {H}x{W}", gap=a[ok, 3])
    # polygons, the reference's rasteriser on the UNPADDED frame
    segs = _polys(rs, B, H, W)
    pm = np.stack([np.logical_or.reduce([P.create_boolean_mask_from_polygon((W, H), [part])[0] for part in seg]) for seg in segs])
    polys = la.pack_polygons(segs, H, W)
    np.testing.assert_array_equal(np_(la.poly_decode(polys)).astype(bool), pm)     # (the plane decoder, unpadded, for reference)
    b, s, a = (np_(t) for t in la.fit_instances_poly(depth, polys, K))
    ref = [O.fit_instance(depth[i], pm[i], K, None) for i in range(B)]
    assert s.tolist() == [r[1] for r in ref]
    np.testing.assert_array_equal(a[s == 0, 2], pm.reshape(B, -1).sum(1)[s == 0])    # mask pixels: no padding column counted
    ok = (s == 0) & (a[:, 3] > 1e-6)
    assert_records(b[ok], np.array([r[0] for r in ref])[ok], f"poly {H}x{W}", gap=a[ok, 3])
    # the fused filter: statistics and keep rule of the unpadded frame
    b2, s2, a2, st = (np_(t) for t in la.fit_instances_poly(depth, polys, K, filter=True))
    want = np_(la.mask_stats_poly(polys))
    np.testing.assert_array_equal(st, want)
    keep = np.array([O.keep_instance(tuple(r), H, False) for r in want])
    np.testing.assert_array_equal(s2 != 6, keep)
    np.testing.assert_array_equal(b2[keep], b[keep])
    r4 = [np_(t) for t in la.fit_instances_rle(depth, rles, K, filter=True)]
    np.testing.assert_array_equal(r4[3], np_(la.mask_stats_rle(rles)))


def test_prepadded_depth_and_fit_annotations(la):
    """A caller that pads once (pad_depth_rows) and passes frame_width gets the records of the automatic route; fit_annotations on an
    odd-width image - tensor route and host-pointer route - equals the two-step route."""
    import torch

    rs = np.random.RandomState(3)
    H, W, B = 375, 500, 14
    depth = rs.uniform(0.5, 10, (H, W)).astype(np.float32)
    K = np.array([[400.0, 0, 250], [0, 400.0, 187], [0, 0, 1]])
    segs = _polys(rs, B, H, W)
    polys = la.pack_polygons(segs, H, W)
    auto = la.fit_instances_ex(depth, K, polys=polys, image_size=(W, H))
    dp, w0 = la.pad_depth_rows(depth)
    assert w0 == W and dp.shape == (H, 512) and la.padded_width(W) == 512
    pre = la.fit_instances_ex(dp, K, polys=polys, image_size=(W, H), frame_width=W)
    for k in ("boxes", "status", "aux", "boxes2d"):
        np.testing.assert_array_equal(np_(pre[k]), np_(auto[k]))
    with pytest.raises(ValueError, match="frame_width"):
        la.fit_instances_ex(dp, K, polys=polys, frame_width=W - 1)
    anns = [{"iscrowd": 0, "bbox": [1.0, 2.0, 3.0, 4.0], "category_id": 1 + i % 3, "segmentation": segs[i], "area": 5000.0} for i in range(B)]
    for i in range(0, B, 3):
        m = np.logical_or.reduce([P.create_boolean_mask_from_polygon((W, H), [part])[0] for part in segs[i]])
        anns[i]["segmentation"] = O.rle_encode(m)
    ground = np.array([[0.02, -0.97, 0.1, 1.0]] * B) + 0.02 * rs.randn(B, 4)
    t = la.fit_annotations(anns, (W, H), depth, K, ground=ground, to_host=True)                                  # tensor route
    h = la.fit_annotations(anns, (W, H), torch.as_tensor(depth, device="cuda"), K, ground=ground, to_host=True)    # host-pointer route
    assert t[0] == h[0] and t[2] == h[2] and len(t[1]) >= 3
    np.testing.assert_array_equal(t[1], h[1]); np.testing.assert_array_equal(t[3], h[3]); np.testing.assert_array_equal(t[4], h[4])
    ball, sall = (np_(x) for x in la.fit_annotations_all(anns, (W, H), depth, K, ground=ground, filter=True))   # (the sharded path's local step)
    np.testing.assert_array_equal(np.nonzero(sall != 6)[0], t[1])
    np.testing.assert_array_equal(ball[t[1]], t[3]); np.testing.assert_array_equal(sall[t[1]], t[4])
    bb0, segs0, kept0, cats0 = la.filter_annotations(anns, (W, H))
    np.testing.assert_array_equal(t[1], kept0)
    masks = np_(la.segmentations_to_masks(segs0, H, W)).astype(bool)
    ref = [O.fit_instance(depth, masks[j], K, ground[kept0[j]]) for j in range(len(kept0))]
    assert t[4].tolist() == [r[1] for r in ref]
    assert_records(t[3], np.array([r[0] for r in ref]), "fit_annotations 375x500")


def test_frame_width_argument_checks(la):
    import torch
    from labelany3d_amd._lib import FitArgs, lib

    dev = torch.device("cuda", 0)
    H, W = 64, 96
    depth = torch.rand((1, H, W), device=dev) + 1.0
    K = torch.tensor([[80.0, 0, 40], [0, 80.0, 32], [0, 0, 1]], dtype=torch.float64, device=dev)
    f = la.InstanceFitter(1, H, W, dev)
    mask = torch.ones((1, H, W), dtype=torch.uint8, device=dev)
    counts = torch.tensor([0, H * 80], dtype=torch.int32, device=dev)
    offs = torch.tensor([0, 2], dtype=torch.int64, device=dev)

    def call(frame_width, use_mask=False, Wc=W):
        a = FitArgs(); a.struct_size = C.sizeof(FitArgs)
        a.B, a.H, a.W = 1, H, Wc
        a.depth, a.K = depth.data_ptr(), K.data_ptr()
        if use_mask:
            a.mask = mask.data_ptr()
        else:
            a.rle_counts, a.rle_offsets = counts.data_ptr(), offs.data_ptr()
        a.filter_boundary = -1
        a.out, a.status, a.aux = f.boxes[0].data_ptr(), f.status[0].data_ptr(), f.aux[0].data_ptr()
        a.workspace, a.stream = f.workspace[0].data_ptr(), torch.cuda.current_stream().cuda_stream
        a.frame_width = frame_width
        return lib.la3d_fit_instances_ex(C.byref(a))

    assert call(0) == 0 and call(W) == 0 and call(80) == 0
    torch.cuda.synchronize()
    assert int(f.status[0][0]) == 0 and float(f.aux[0][0, 2]) == H * 80
    assert call(W + 1) != 0 and call(-3) != 0 and b"frame_width" in lib.la3d_last_error()
    assert call(80, use_mask=True) != 0                      # u8 planes: the caller pads the planes with zeros instead
    assert call(70, Wc=80) != 0                              # rows must be word aligned when a frame width is given


def test_small_batches_of_u8_planes_on_odd_widths(la):
    """fit_instances pads a SMALL batch of u8 planes on a frame of odd width by itself (mask planes and depth rows, zeros on the right):
    the records are the oracle's; image_index is range-checked on the host for host arrays and once per tensor on the device."""
    import torch

    rs = np.random.RandomState(4)
    H, W, B, P_ = 375, 500, 12, 3
    depth = rs.uniform(0.5, 10, (P_, H, W)).astype(np.float32)
    masks = _blobs(rs, B, H, W)
    K = np.array([[420.0, 0, 251], [0, 415.0, 188], [0, 0, 1]])
    img = rs.randint(0, P_, B).astype(np.int32)
    b, s, a = (np_(t) for t in la.fit_instances(depth, masks, K, image_index=img))
    ref, rst, _, nval = O.fit_instances(depth, masks, np.broadcast_to(K, (P_, 3, 3)), depth_index=img)
    assert s.tolist() == list(rst)
    ok = (s == 0) & (a[:, 3] > 1e-6)
    assert_records(b[ok], ref[ok], "u8 planes 375x500", gap=a[ok, 3])
    np.testing.assert_array_equal(a[s == 0, 1], nval[s == 0])
    with pytest.raises(ValueError, match="out of range"):
        la.fit_instances(depth, masks, K, image_index=np.full(B, P_, np.int32))
    it = torch.as_tensor(img, device="cuda")
    for _ in range(3):                                           # (the device-side check runs once for this tensor)
        b2 = np_(la.fit_instances(depth, masks, K, image_index=it)[0])
    np.testing.assert_array_equal(b2, b)
    it2 = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="out of range"):
        la.fit_instances(depth, masks, K, image_index=it2)
    it.fill_(P_)                                                 # modified in place: a new version is checked again
    with pytest.raises(ValueError, match="out of range"):
        la.fit_instances(depth, masks, K, image_index=it)


def test_pad_rows_entry(la):
    """la3d_pad_rows: [rows][W] f32 -> [rows][Wp] f32 with zeros on the right, for any W; argument checks."""
    import torch
    from labelany3d_amd._lib import lib

    dev = torch.device("cuda", 0)
    for rows, W, Wp in ((7, 427, 448), (1, 1, 4), (33, 500, 512), (5, 64, 64), (3, 333, 352), (0, 10, 32)):
        src = torch.rand((max(rows, 1), W), device=dev)[:rows].contiguous()
        dst = torch.full((max(rows, 1), Wp), -1.0, device=dev)[:rows]
        rc = lib.la3d_pad_rows(src.data_ptr() if rows else None, rows, W, Wp, dst.data_ptr() if rows else None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.la3d_last_error()
        torch.cuda.synchronize()
        assert torch.equal(dst, torch.nn.functional.pad(src, (0, Wp - W)))
    a = torch.zeros(64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.la3d_pad_rows(a.data_ptr(), 2, 8, 4, a.data_ptr(), s) != 0          # Wp < W
    assert lib.la3d_pad_rows(a.data_ptr(), 2, 8, 10, a.data_ptr(), s) != 0         # Wp % 4 != 0
    assert lib.la3d_pad_rows(None, 2, 8, 8, a.data_ptr(), s) != 0 and b"la3d_pad_rows" in lib.la3d_last_error()
    d, w = la.pad_depth_rows(np.ones((2, 5, 33), np.float32))
    assert w == 33 and d.shape == (2, 5, 64) and float(d.sum()) == 2 * 5 * 33 and la.pad_depth_rows(d)[0] is d

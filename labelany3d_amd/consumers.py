"""Box consumers on the device: what the reference does with the fitted boxes right after the
path (src/tools/combine_results.py) — project the 8 corners with K, take the 2-D box and its clamp to the frame,
and build the IoU matrix the Hungarian matching consumes.  The assignment itself stays scipy
(`linear_sum_assignment`, as in the reference :138): it is a tiny sequential problem per image."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, lib
from .batched import _as_dev, _dev, _ptr, _stream


def project_boxes(boxes, K, image_size, image_index=None, stream=None) -> torch.Tensor:
    """boxes (B,39) records -> (B,8) = bbox2D_proj [min_x,min_y,max_x,max_y] + bbox2D_trunc (clamped to
    [0,W]x[0,H]); image_size = (W, H) as in the reference (:233-252).  K (3,3) or (P,3,3) with image_index."""
    dev = boxes.device if isinstance(boxes, torch.Tensor) and boxes.is_cuda else _dev()
    b = _as_dev(boxes, torch.float64, dev)
    k = _as_dev(K, torch.float64, dev, cache=True)
    if k.dim() == 2:
        k = k[None]
    B = b.shape[0]
    ii = None if image_index is None else _as_dev(image_index, torch.int32, dev)
    if ii is None and k.shape[0] not in (1, B):
        raise ValueError("K must be (3,3), (B,3,3), or (P,3,3) with image_index")
    out = torch.empty((B, 8), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_project_boxes(_ptr(b), _ptr(k), 9 if k.shape[0] > 1 else 0, _ptr(ii), B, float(image_size[0]),
                                     float(image_size[1]), _ptr(out), _stream(stream)), "la3d_project_boxes")
    return out


def iou2d_matrix(boxes0, boxes1, stream=None) -> torch.Tensor:
    """(n0,4) x (n1,4) xyxy boxes -> (n0,n1) IoU (reference iou2D, :111-124)."""
    dev = boxes0.device if isinstance(boxes0, torch.Tensor) and boxes0.is_cuda else _dev()
    a, b = _as_dev(boxes0, torch.float64, dev).reshape(-1, 4), _as_dev(boxes1, torch.float64, dev).reshape(-1, 4)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_iou_matrix(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream(stream)), "la3d_iou_matrix")
    return out


def unproject_matches(depth, uv, fx=560.44, fy=560.44, cx=256.0, cy=256.0, flip=512.0, R=None, T=None, stream=None):
    """Sparse pinhole unprojection at 2-D match points (reference src/matching/matcher.py:70-91; the defaults are
    its hard-coded intrinsics and 512-flip): returns (points (N,3) float64, valid (N,) bool) on the GPU; matches
    whose depth is -1 are flagged invalid (the reference drops them, :72-75).  With R (3,3) and T (3,):
    world = R (p - T).  ``flip=None`` disables the 512-u / 512-v flip."""
    import ctypes as C

    dev = depth.device if isinstance(depth, torch.Tensor) and depth.is_cuda else _dev()
    d = _as_dev(depth, torch.float32, dev)
    m = _as_dev(uv, torch.float64, dev).reshape(-1, 2)
    N = m.shape[0]
    out = torch.empty((N, 3), dtype=torch.float64, device=dev)
    valid = torch.empty(N, dtype=torch.int32, device=dev)
    R9 = T3 = None
    if R is not None or T is not None:
        R9 = (C.c_double * 9)(*np.asarray(np.eye(3) if R is None else R, dtype=np.float64).ravel())
        T3 = (C.c_double * 3)(*np.asarray(np.zeros(3) if T is None else T, dtype=np.float64).ravel())
    with torch.cuda.device(dev):
        check(lib.la3d_unproject_matches(_ptr(d), d.shape[0], d.shape[1], _ptr(m), N, fx, fy, cx, cy, int(flip is not None),
                                         float(flip or 0.0), R9, T3, _ptr(out), _ptr(valid), _stream(stream)),
              "la3d_unproject_matches")
    return out, valid.bool()


def correspondences_to_world(matches_im0, matches_im1, true_shape0, true_shape1, depth, T, R, image_size: int = 512):
    """The geometry of ``ImageMatcher.get_correspondences`` after its network calls (reference src/matching/matcher.py:37-91):
    keep matches at least 3 px inside both views (:38-55), shift them by the crop offset of the ``image_size`` square render
    (``cy - int(3 * halfw / 4)`` rows, :58-64), drop those whose render depth is -1 (:69-75), unproject with the flipped pinhole
    and the hard-coded intrinsics on the GPU (:77-86) and move them to world coordinates ``R (p - T)`` (:88-90).
    Returns ``(points_world (N,3) float64, unprocessed_matches0 (N,2))`` as NumPy arrays, like the reference."""
    m0, m1 = np.asarray(matches_im0), np.asarray(matches_im1)
    (H0, W0), (H1, W1) = (int(v) for v in true_shape0), (int(v) for v in true_shape1)
    ok = ((m0[:, 0] >= 3) & (m0[:, 0] < W0 - 3) & (m0[:, 1] >= 3) & (m0[:, 1] < H0 - 3) &
          (m1[:, 0] >= 3) & (m1[:, 0] < W1 - 3) & (m1[:, 1] >= 3) & (m1[:, 1] < H1 - 3))
    m0, m1 = m0[ok], m1[ok]
    cx = cy = image_size // 2
    halfw = ((2 * cx) // 16) * 8
    off = np.array([0, cy - int(3 * halfw / 4)])
    u0, u1 = m0 + off, m1 + off
    if len(u1) == 0:
        return np.zeros((0, 3)), u0
    pts, valid = unproject_matches(depth, u1.astype(np.float64), flip=float(image_size), cx=float(cx), cy=float(cy),
                                   R=np.asarray(R, dtype=np.float64).reshape(3, 3), T=np.asarray(T, dtype=np.float64).reshape(3))
    valid = valid.cpu().numpy()
    return pts.cpu().numpy()[valid], u0[valid]


def hungarian_matching(boxes0, boxes1):
    """Reference hungarian_matching (:127-144): IoU matrix on the GPU, assignment with SciPy as in the reference.
    Returns [(index0, index1, iou), ...]."""
    from scipy.optimize import linear_sum_assignment

    iou = iou2d_matrix(boxes0, boxes1).cpu().numpy()
    rows, cols = linear_sum_assignment(-iou)
    return [(int(i), int(j), float(iou[i, j])) for i, j in zip(rows, cols)]

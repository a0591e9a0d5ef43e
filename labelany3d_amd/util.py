"""Drop-in for the hot-path function of the reference's ``util`` module: ``depth_to_points``
(reference src/util.py:52-75).  Runs on the MI355X through ``la3d_unproject`` (include/la3d.h).
Also here: the box overlay of the scene harness, ``project_to_2d`` / ``draw_cube`` (reference :225-289) - host code; the
pixels are OpenCV's (``cv2.circle`` / ``line`` / ``putText``), the geometry and the order of the drawing calls are this module's."""
from __future__ import annotations

import json
import os

import numpy as np

from ._lib import check, lib
from .batched import unproject


def depth_to_points(depth, K=None, R=None, t=None):
    """Pinhole back-projection of a depth batch; returns the points of batch element 0 as
    ``(H, W, 3)`` float64, like the reference (:75).  ``u`` = column, ``v`` = row, no half-pixel
    offset (:62-69).  NumPy in -> NumPy out; a CUDA tensor in -> a CUDA tensor out (no copy back).

    As in the reference, ``K=None`` is not usable (``np.linalg.inv(None)`` raises there, :56).
    """
    if K is None:
        raise TypeError("depth_to_points: K is required (the reference fails in np.linalg.inv(None))")
    try:
        import torch

        is_t = isinstance(depth, torch.Tensor)
    except ImportError:  # pragma: no cover
        is_t = False
    if is_t:
        if depth.dim() != 3:
            raise ValueError("depth must be (B, H, W)")
        return unproject(depth[0], K, R, t)
    d = np.asarray(depth)
    if d.ndim != 3:
        raise ValueError("depth must be (B, H, W)")
    # host frame in, host points out: ONE C call (la3d_unproject_host: upload, kernel, download on the library's private stream)
    d0 = np.ascontiguousarray(d[0], dtype=np.float32)
    k = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3))
    rt = None
    if R is not None or t is not None:
        rt = np.concatenate([(np.eye(3) if R is None else np.asarray(R, dtype=np.float64)).reshape(-1),
                             (np.zeros(3) if t is None else np.asarray(t, dtype=np.float64)).reshape(-1)])
    out = np.empty(d0.shape + (3,), np.float64)
    check(lib.la3d_unproject_host(d0.ctypes.data, k.ctypes.data, None if rt is None else rt.ctypes.data, d0.shape[0], d0.shape[1],
                                  out.ctypes.data, 1), "la3d_unproject_host")
    return out


def project_to_2d(point_3d, camera_matrix):
    """Pinhole projection of one camera-frame point: ``(K @ p)[:2] / (K @ p)[2]`` (reference src/util.py:225-227)."""
    h = np.dot(camera_matrix, point_3d)
    return h[:2] / h[2]


# the twelve sides of a box in the corner order of convert_box_vertices (reference src/util.py:270-272): bottom ring, top ring, pillars
CUBE_EDGES = tuple((i, (i + 1) % 4) for i in range(4)) + tuple((4 + i, 4 + (i + 1) % 4) for i in range(4)) + tuple((i, i + 4) for i in range(4))


def cube_overlay(cube_list, K):
    """What ``draw_cube`` draws for the boxes of one scene, as data: per box ``{"points": (8,2) int pixel positions (np.round of
    the projected corners, :266), "edges": [(start, end), ...] of the twelve sides, "label": category_name, "label_at": (x, y)}``
    with the label ten pixels above the projected corner of smallest y (the first one on ties; truncated, not rounded: :259-264,
    :281-282).  Raises like the reference on a corner with z = 0 only through NumPy's warnings (inf / nan positions)."""
    K = np.asarray(K, dtype=np.float64)
    out = []
    for cube in cube_list:
        uv = np.array([project_to_2d(np.array(v), K) for v in cube["bbox3D_cam"]])
        px = [tuple(np.round(q).astype(int)) for q in uv]
        top = int(np.argmin(uv[:, 1])) if len(uv) and (uv[:, 1] < np.inf).any() else -1
        # (the reference's scan keeps the FIRST strictly smaller y: argmin's tie rule; a corner with y = nan never wins there, and a
        # box whose every y is inf / nan gets no label)
        if top >= 0 and np.isnan(uv[:, 1]).any():
            ys = np.where(np.isnan(uv[:, 1]), np.inf, uv[:, 1])
            top = int(np.argmin(ys)) if (ys < np.inf).any() else -1
        out.append({"points": px, "edges": [(px[a], px[b]) for a, b in CUBE_EDGES], "label": f'{cube["category_name"]}',
                    "label_at": None if top < 0 else (int(uv[top, 0]), int(uv[top, 1]) - 10)})
    return out


def draw_cube(scene_dir, is_ground=False):
    """The scene harness's box overlay (reference src/util.py:232-289, called at src/batch_scripts/whole.py:128): reads
    ``cam_params.json`` (``K``), ``3dbbox_ground.json`` / ``3dbbox.json`` and ``input.png`` of ``scene_dir``, draws every box - a
    filled green circle of radius 3 on each projected corner, the twelve sides in blue (BGR (255, 0, 0)), two pixels thick, the
    category name in red above the topmost corner - and writes ``vis_3dbox.png`` / ``vis_3dbox_no_ground.png``.  The raster work
    is OpenCV's: ``cv2`` (and Pillow for the read) must be importable, as in the reference."""
    import cv2
    from PIL import Image

    with open(os.path.join(scene_dir, "cam_params.json")) as f:
        K = np.array(json.load(f)["K"])
    with open(os.path.join(scene_dir, "3dbbox_ground.json" if is_ground else "3dbbox.json")) as f:
        cubes = json.load(f)
    image = cv2.cvtColor(np.array(Image.open(os.path.join(scene_dir, "input.png"))), cv2.COLOR_RGB2BGR)
    for item in cube_overlay(cubes, K):
        for q in item["points"]:
            cv2.circle(image, q, radius=3, color=(0, 255, 0), thickness=-1)
        for a, b in item["edges"]:
            cv2.line(image, a, b, (255, 0, 0), 2)
        if item["label_at"] is not None:
            cv2.putText(image, item["label"], item["label_at"], cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 0, 255), 1)
    cv2.imwrite(os.path.join(scene_dir, "vis_3dbox.png" if is_ground else "vis_3dbox_no_ground.png"), image)

"""Drop-in for the reference's ``util_3dbox`` module (reference src/util_3dbox.py) — same function
names, arguments, return types, exceptions and side effects; the box fit runs on the MI355X through
libla3d.so (``la3d_fit_points`` / ``la3d_fit_instances``, include/la3d.h).  No CPU fit path exists.

The small 3x3 helpers (normalize / rotate_y / rotation_matrix_from_vectors / convert_box_vertices /
point_to_plane_distance, reference :20-103) are host-side argument utilities — a handful of flops
on 3-vectors that callers use to prepare inputs — and stay NumPy.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os

import numpy as np

from . import _lib
from .batched import fit_points

_CORNER_SIGNS = np.array(
    [(-1, -1, -1), (+1, -1, -1), (+1, +1, -1), (-1, +1, -1), (-1, -1, +1), (+1, -1, +1), (+1, +1, +1), (-1, +1, +1)],
    dtype=np.float64,
)  # corner order of reference :83-92

_METHODS = {"pca": _lib.METHOD_PCA, "convex_hull": _lib.METHOD_CONVEX_HULL}
_MESSAGES = {
    _lib.BOX_EMPTY: "No valid points after removing NaN values",  # reference :143
    _lib.BOX_BAD_GROUND: "No valid points after removing NaN values",  # NaN rotation -> every row NaN -> :143
    _lib.BOX_TOO_FEW: "n_components=2 must be between 0 and min(n_samples, n_features)=1 with svd_solver='full'",
    _lib.BOX_NONFINITE: "Input X contains infinity or a value too large for dtype('float64').",
    _lib.BOX_UNSUPPORTED: "convex_hull on more than 2048 valid points is not supported (the reference subsamples to 500)",
}


# ---- basic geometry (reference :20-64) -------------------------------------------------------
def normalize(v):
    """Unit vector; a zero vector is returned unchanged (reference :20-25)."""
    n = np.linalg.norm(v)
    return v if n == 0 else v / n


def rotate_y(yaw):
    """Rotation about +y (reference :28-34)."""
    c, s = np.cos(yaw), np.sin(yaw)
    out = np.zeros((3, 3))
    out[0, 0], out[0, 2], out[1, 1], out[2, 0], out[2, 2] = c, s, 1, -s, c
    return out


def rotation_matrix_from_vectors(vec1, vec2):
    """Rodrigues rotation taking vec1 onto vec2 (reference :37-55); parallel or antiparallel
    inputs yield an all-NaN matrix exactly as in the reference."""
    a = normalize(np.asarray(vec1, dtype=np.float64))
    b = normalize(np.asarray(vec2, dtype=np.float64))
    ax = np.cross(a, b)
    kx = np.zeros((3, 3))
    kx[0, 1], kx[0, 2], kx[1, 0], kx[1, 2], kx[2, 0], kx[2, 1] = -ax[2], ax[1], ax[2], -ax[0], -ax[1], ax[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.eye(3) + kx + (kx @ kx) * (1 - np.dot(a, b)) / (np.linalg.norm(ax) ** 2)


def point_to_plane_distance(plane, x, y, z):
    """|ax+by+cz+d| / |(a,b,c)| (reference :58-64)."""
    a, b, c, d = np.array(plane)
    return abs(a * x + b * y + c * z + d) / np.sqrt(a**2 + b**2 + c**2)


def convert_box_vertices(center_x, center_y, center_z, l, w, h, yaw):
    """8 corners of a yawed box, fixed order (reference :71-103)."""
    half = np.array([l / 2, w / 2, h / 2])
    rot = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    return np.dot(_CORNER_SIGNS * half, rot.T) + np.array([center_x, center_y, center_z])


# ---- box fit (reference :106-224) --------------------------------------------------------------
def _fit_one(in_pc, ground_equ, method, subsample=True):
    """One cloud through la3d_fit_points, with the reference's RNG side effect for N > 500 (estimate_bbox only: the yaw
    helpers of the reference see the cloud they are given, :181-224)."""
    if method not in ("pca", "convex_hull"):
        raise ValueError(f"Unknown method: {method}. Use 'pca' or 'convex_hull'")  # reference :151
    pc = np.asarray(in_pc)
    pc = pc.reshape(-1, 3) if pc.size else np.zeros((0, 3))
    if subsample and pc.shape[0] > _lib.NSAMPLE:  # reference :123-125 — global stream, with replacement
        # the draw happens here exactly as there (in_pc = in_pc[idx]); only the 500 drawn rows travel to the GPU
        pc = pc[np.random.randint(0, pc.shape[0], _lib.NSAMPLE)]
    ground = None if ground_equ is None else np.asarray(ground_equ, dtype=np.float64).reshape(-1)[:4][None]
    if ground is not None and ground.shape[1] < 4:
        ground = np.concatenate([ground, np.zeros((1, 4 - ground.shape[1]))], axis=1)  # only [:3] is used (:129)
    if ground is not None:
        ground = np.ascontiguousarray(ground, dtype=np.float64)
    # ONE C call on host pointers (la3d_estimate_bbox_host, round 5): the library's pinned block is the staging area, the kernel
    # pulls the cloud over the link, writes the record back and raises a flag the call polls - no torch tensor in between
    pts = np.ascontiguousarray(pc, dtype=np.float64)
    host = np.empty(_lib.REC + _lib.AUX, np.float64)
    st_c = C.c_int32(-1)
    _lib.check(_lib.lib.la3d_estimate_bbox_host(pts.ctypes.data, pts.shape[0], None if ground is None else ground.ctypes.data,
                                                _METHODS[method], host.ctypes.data, host[_lib.REC:].ctypes.data, C.byref(st_c)),
               "la3d_estimate_bbox_host")
    rec, aux = host[:_lib.REC], host[_lib.REC:]
    st = int(st_c.value)
    if st != _lib.BOX_OK:
        raise ValueError(_MESSAGES[st])
    if method == "convex_hull" and aux[3] >= 0:  # no 2-D hull: the kernel took the reference's PCA fallback (:222-224)
        print("ConvexHull failed: degenerate footprint (fewer than 3 hull vertices), falling back to PCA")
    return rec, aux


def estimate_bbox(in_pc, cat_name=None, ground_equ=None, method="pca"):
    """Oriented box of a point cloud (reference :106-178).

    Returns ``(vertices (8,3) f64 [fp16-quantised, :165], center_cam (3,), dimension [dz, dy, dx]
    (list of np.float64, :175), R_cam (3,3))``.  Raises ``ValueError`` where the reference does; prints
    the reference's per-box line (:162); draws from ``np.random`` when N > 500 (:124).
    """
    rec, _ = _fit_one(in_pc, ground_equ, method)
    dz, dy, dx = (np.float64(x) for x in rec[3:6])
    print(f"[{method}] dx={dx:.3f}, dy={dy:.3f}, dz={dz:.3f}")
    return rec[15:39].reshape(8, 3).copy(), rec[0:3].copy(), [dz, dy, dx], rec[6:15].reshape(3, 3).copy()


def _estimate_yaw_pca(rotated_pc):
    """Yaw of the first principal axis of the (x,z) footprint (reference :181-186)."""
    _, aux = _fit_one(rotated_pc, None, "pca", subsample=False)   # no subsampling, no RNG draw: as the reference helper
    return np.float64(aux[0])


def _estimate_yaw_convex_hull(rotated_pc):
    """Yaw of the minimum-area enclosing rectangle over hull edges (reference :189-224)."""
    # PCA fallback (:222-224) happens inside the kernel; no subsampling (clouds above 2048 valid points are not supported by
    # the hull kernel - the reference only ever feeds this helper the <= 500 points estimate_bbox kept)
    _, aux = _fit_one(rotated_pc, None, "convex_hull", subsample=False)
    return np.float64(aux[0])


# ---- scene harness (reference :231-294) ----------------------------------------------------------
def save_3d_with_ground_alignment_bbox(scene_dir, bbox_method="pca"):
    """Fit a ground-aligned box for every ``reconstruction/*.glb`` of a scene and write
    ``3dbbox_ground.json`` with the reference's six keys (reference :231-294).  Mesh loading and the
    500-point surface sampling need ``trimesh`` exactly as in the reference; all clouds of the scene
    are fitted in ONE la3d_fit_points launch."""
    import trimesh  # same hard dependency as the reference (:9)

    recons_dir = os.path.join(scene_dir, "reconstruction")
    names = [f for f in os.listdir(recons_dir) if f.endswith(".glb") and f not in ("full_scene.glb", "background.ply")]
    todo = []
    for obj in names:
        obj_id, rest = obj.split("_", 1)
        category = rest.split(".", 1)[0]
        mesh = trimesh.load(os.path.join(recons_dir, obj))
        upright = np.load(os.path.join(recons_dir, f"{obj.split('.', 1)[0]}_canonical_upright.npy"))
        if isinstance(mesh, trimesh.Scene):
            mesh = mesh.dump()[0]
        if mesh.is_empty or mesh.area == 0 or len(mesh.faces) == 0:
            print(f"Invalid mesh at {os.path.join(recons_dir, obj)}, skipping.")
            continue
        cloud = np.array(trimesh.points.PointCloud(mesh.sample(500)).vertices)
        todo.append((obj, obj_id, category, cloud, np.asarray(upright, dtype=np.float64).reshape(-1)[:4]))
    bbox_list = []
    if todo:
        if bbox_method not in ("pca", "convex_hull"):
            for obj, *_ in todo:  # the reference raises per object and skips it (:279-281)
                print(f"Error estimating bbox for {obj}: Unknown method: {bbox_method}. Use 'pca' or 'convex_hull'")
            todo = []
    if todo:
        grounds = np.zeros((len(todo), 4))
        for i, t in enumerate(todo):
            grounds[i, : len(t[4])] = t[4]
        idx = None
        if any(len(t[3]) > _lib.NSAMPLE for t in todo):  # not reached with mesh.sample(500); kept for parity
            idx = np.zeros((len(todo), _lib.NSAMPLE), np.int32)
            for i, t in enumerate(todo):
                if len(t[3]) > _lib.NSAMPLE:
                    idx[i] = np.random.randint(0, len(t[3]), _lib.NSAMPLE)
        boxes, status, _ = fit_points([t[3] for t in todo], grounds, idx, bbox_method)
        boxes, status = boxes.cpu().numpy(), status.cpu().numpy()
        for (obj, obj_id, category, _, _), rec, st in zip(todo, boxes, status):
            if st != _lib.BOX_OK:
                print(f"Error estimating bbox for {obj}: {_MESSAGES[int(st)]}")
                continue
            print(f"[{bbox_method}] dx={rec[5]:.3f}, dy={rec[4]:.3f}, dz={rec[3]:.3f}")
            bbox_list.append(dict(obj_id=obj_id, category_name=category, center_cam=rec[0:3].tolist(),
                                  R_cam=rec[6:15].reshape(3, 3).tolist(), dimensions=[float(x) for x in rec[3:6]],
                                  bbox3D_cam=rec[15:39].reshape(8, 3).tolist()))
    with open(os.path.join(scene_dir, "3dbbox_ground.json"), "w") as f:
        json.dump(bbox_list, f)
    return bbox_list

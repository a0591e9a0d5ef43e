"""CPU ORACLE — test infrastructure, NOT product code.

A NumPy restatement of the LabelAny3D geometric hot path (depth back-projection ->
oriented 3D box fit).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker /
the timed CPU baseline.  The product path (``labelany3d_amd``) never imports it.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so this restatement is pinned against outputs of the reference itself,
generated in the build container by importing ``/root/reference/src/{util,util_3dbox,
cam_utils}.py`` read-only (``tests/golden/make_golden.py``) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks every function below
against those fixtures.

Every function cites the reference lines it follows (paths relative to /root/reference).
The one third-party piece on the path is scikit-learn's ``PCA(2).fit`` (reference pins
scikit-learn==1.5.0, requirements.txt:23; call site src/util_3dbox.py:181-186).  Its
published algorithm for 2 features is restated in ``yaw_pca_closed_form``:
    n >= 20 -> 'covariance_eigh' solver: C = X^T X - n mu mu^T, C /= n-1, eigh, sort
               descending;  n < 20 -> LAPACK SVD of the centred data;
    both   -> svd_flip(u_based_decision=False): each component is multiplied by the sign of
               its largest-|.| entry (first index on ties).
The first principal axis of a symmetric 2x2 matrix [[a,b],[b,c]] is at angle
theta = atan2(2b, a-c)/2, so no eigen-solver is needed.
"""
from __future__ import annotations

import math

import numpy as np

SUBSAMPLE = 500  # src/util_3dbox.py:123

# status codes shared with the C-ABI (include/la3d.h)
ST_OK, ST_EMPTY, ST_BAD_GROUND, ST_TOO_FEW, ST_NONFINITE = 0, 1, 2, 3, 4


# --------------------------------------------------------------------------------------
# helpers  (src/util_3dbox.py:20-64)
# --------------------------------------------------------------------------------------
def normalize(v):
    """src/util_3dbox.py:20-25 — returns the input unchanged when its norm is 0."""
    norm = np.linalg.norm(v)
    if norm == 0:
        return v
    return v / norm


def rotate_y(yaw):
    """src/util_3dbox.py:28-34."""
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotation_matrix_from_vectors(vec1, vec2):
    """src/util_3dbox.py:37-55 — Rodrigues form I + K + K^2 (1-cos)/|axis|^2.

    Parallel / antiparallel inputs give 0/0 = NaN everywhere (reference behaviour, kept).
    """
    vec1 = normalize(np.asarray(vec1, dtype=np.float64))
    vec2 = normalize(np.asarray(vec2, dtype=np.float64))
    axis = np.cross(vec1, vec2)
    cos_theta = np.dot(vec1, vec2)
    k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.eye(3) + k + np.dot(k, k) * (1 - cos_theta) / (np.linalg.norm(axis) ** 2)


def point_to_plane_distance(plane, x, y, z):
    """src/util_3dbox.py:58-64."""
    a, b, c, d = np.array(plane)
    return abs(a * x + b * y + c * z + d) / np.sqrt(a**2 + b**2 + c**2)


def convert_box_vertices(cx, cy, cz, l, w, h, yaw):
    """src/util_3dbox.py:71-103 — fixed corner order, corners @ rotate_y(yaw)^T + c."""
    sgn = np.array(
        [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]],
        dtype=np.float64,
    )
    local = sgn * np.array([l / 2, w / 2, h / 2])
    rot = np.array(
        [[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]]
    )
    return np.dot(local, rot.T) + np.array([cx, cy, cz])


# --------------------------------------------------------------------------------------
# A1  depth_to_points  (src/util.py:52-75)
# --------------------------------------------------------------------------------------
def depth_to_points(depth, K, R=None, t=None):
    """src/util.py:52-75.  depth (B,H,W) -> (H,W,3) float64 of batch element 0.

    p = (d * Kinv) @ [u, v, 1]  with u = column index, v = row index, no half-pixel
    offset (src/util.py:62-69), then R @ p + t (src/util.py:74).  Written without the
    (H,W,3,3) temporary of the reference; the arithmetic per pixel is the same three
    products and two additions per coordinate, in float64.
    """
    depth = np.asarray(depth)
    Kinv = np.linalg.inv(np.asarray(K, dtype=np.float64))
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    t = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64)
    H, W = depth.shape[1:3]
    d = depth[0].astype(np.float64)
    u = np.arange(W, dtype=np.float64)[None, :]
    v = np.arange(H, dtype=np.float64)[:, None]
    with np.errstate(invalid="ignore", over="ignore"):
        p = np.empty((H, W, 3))
        for i in range(3):
            # (d*Kinv[i,0])*u + (d*Kinv[i,1])*v + (d*Kinv[i,2])*1   (src/util.py:71-72)
            p[..., i] = (d * Kinv[i, 0]) * u + (d * Kinv[i, 1]) * v + (d * Kinv[i, 2])
        q = np.empty_like(p)
        for i in range(3):
            q[..., i] = R[i, 0] * p[..., 0] + R[i, 1] * p[..., 1] + R[i, 2] * p[..., 2] + t[i]
    return q


def depth_to_points_refstyle(depth, K, R=None, t=None):
    """Same function written with the reference's own array operations (stacked 3x3
    matmul over an (H,W,3,3) temporary, src/util.py:71-75).  Used as the timed CPU
    baseline so the baseline keeps the reference's cost structure."""
    Kinv = np.linalg.inv(K)
    R = np.eye(3) if R is None else R
    t = np.zeros(3) if t is None else t
    H, W = depth.shape[1:3]
    coord = np.stack(np.meshgrid(np.arange(W), np.arange(H)), -1)
    coord = np.concatenate((coord, np.ones_like(coord)[:, :, [0]]), -1).astype(np.float32)[None]
    D = depth[:, :, :, None, None]
    with np.errstate(invalid="ignore", over="ignore"):
        p1 = D * Kinv[None, None, None, ...] @ coord[:, :, :, :, None]
        p2 = R[None, None, None, ...] @ p1 + t[None, None, None, :, None]
    return p2[:, :, :, :3, 0][0]


# --------------------------------------------------------------------------------------
# A4  yaw  (src/util_3dbox.py:181-186 + scikit-learn PCA, restated)
# --------------------------------------------------------------------------------------
def yaw_from_cov(a, b, c, n):
    """First principal axis of [[a,b],[b,c]] with scikit-learn's sign rule.

    theta = atan2(2b, a-c)/2 in [-pi/2, pi/2]; the unit vector (cos, sin) is flipped so
    that its larger-|.| entry is positive (svd_flip(u_based_decision=False)); yaw =
    atan2(vz, vx) (src/util_3dbox.py:185-186).  Exact isotropy (b == 0 and a == c) is
    solver dependent in the reference: the n >= 20 'covariance_eigh' branch returns
    pi/2 (incl. zero covariance), the n < 20 SVD branch is recorded as 0.
    """
    if b == 0 and a == c:
        return math.pi / 2 if n >= 20 else 0.0
    theta = 0.5 * math.atan2(2.0 * b, a - c)
    vx, vz = math.cos(theta), math.sin(theta)
    if abs(vx) >= abs(vz):
        if vx < 0:
            vx, vz = -vx, -vz
    elif vz < 0:
        vx, vz = -vx, -vz
    return math.atan2(vz, vx)


def yaw_pca_closed_form(rotated_pc):
    """src/util_3dbox.py:181-186 with PCA(2) restated in closed form."""
    x = rotated_pc[:, 0]
    z = rotated_pc[:, 2]
    n = len(x)
    # (scikit-learn validates the array - finiteness - before it looks at n_components: one infinite point gives the infinity
    # message, not the n_components one; checked against scikit-learn in the build container, round 6)
    if not (np.isfinite(x).all() and np.isfinite(z).all()):
        raise ValueError("Input X contains infinity or a value too large for dtype('float64').")
    if n < 2:
        raise ValueError(
            "n_components=2 must be between 0 and min(n_samples, n_features)=%d with svd_solver='full'" % n
        )
    if n < 20:
        # the 'full' solver: LAPACK SVD of the CENTRED data - its axis is that of the centred second moments, with no
        # cancellation however far the cloud lies from the origin (round 6: rounds 1-5 formed the raw-sum expression below for
        # every n, which differs from the reference by ~2^-52 kappa / gap, kappa = pca_kappa(): visible for slivers far away)
        xc, zc = x - x.mean(), z - z.mean()
        return yaw_from_cov(np.dot(xc, xc), np.dot(xc, zc), np.dot(zc, zc), n)
    # the 'covariance_eigh' solver forms C from RAW sums (C = X^T X - n mu mu^T): restated as it is, cancellation included - for
    # kappa >> 1 the reference's own axis carries a rounding error of ~2^-52 kappa / gap that depends on its summation order
    sx, sz = x.sum(), z.sum()
    a = np.dot(x, x) - sx * sx / n
    c = np.dot(z, z) - sz * sz / n
    b = np.dot(x, z) - sx * sz / n
    return yaw_from_cov(a, b, c, n)


def pca_conditioning(rotated_pc):
    """(kappa, gap) of the (x, z) footprint, from the centred second moments in long double.
    kappa = (sum x^2 + sum z^2) / (n * largest eigenvalue): the conditioning of a covariance formed from RAW sums - 1 for a cloud
    centred on the origin, (distance / spread)^2 for a small cloud far away; gap = (l1 - l2) / l1, the relative eigen-gap the
    axis is conditioned by.  Diagnostics for the tests (tolerances), not part of the reference."""
    x = np.asarray(rotated_pc[:, 0], np.longdouble)
    z = np.asarray(rotated_pc[:, 2], np.longdouble)
    if len(x) == 0 or not (np.isfinite(x).all() and np.isfinite(z).all()):
        return float("nan"), float("nan")
    raw = float((x * x).sum() + (z * z).sum())
    xc, zc = x - x.mean(), z - z.mean()
    a, b, c = float((xc * xc).sum()), float((xc * zc).sum()), float((zc * zc).sum())
    rad = math.sqrt((0.5 * (a - c)) ** 2 + b * b)
    l1 = 0.5 * (a + c) + rad
    return (raw / l1, 2.0 * rad / l1) if l1 > 0 else (float("inf"), 0.0)


def pca_kappa(rotated_pc):
    return pca_conditioning(rotated_pc)[0]


def yaw_convex_hull(rotated_pc):
    """src/util_3dbox.py:189-224 — min-area enclosing rectangle over hull edge directions.

    Restated without Qhull: Andrew's monotone chain gives the counter-clockwise hull; the
    reference iterates hull.vertices (counter-clockwise for 2-D input) and keeps the first
    strict minimum (src/util_3dbox.py:216).  The starting vertex of Qhull's list is an
    implementation detail, so on exact area ties (parallel hull edges) the two may pick
    different, equally minimal, edges; tests avoid exact ties.
    """
    pts = np.asarray(rotated_pc)[:, [0, 2]]
    if not np.isfinite(pts).all():
        # Qhull rejects an infinite coordinate (QhullError QH6006, checked with scipy in the build container) -> the except branch
        # (:222-224) -> PCA, where scikit-learn raises on the same coordinate
        return yaw_pca_closed_form(rotated_pc)
    hull = _monotone_chain(pts)
    if len(hull) < 3:
        return yaw_pca_closed_form(rotated_pc)  # Qhull raises -> PCA fallback (:222-224)
    hp = pts[hull]
    best, best_yaw = float("inf"), 0.0
    for i in range(len(hp)):
        e = hp[(i + 1) % len(hp)] - hp[i]
        yaw = np.arctan2(e[1], e[0])
        cs, sn = np.cos(yaw), np.sin(yaw)
        rx = cs * pts[:, 0] - sn * pts[:, 1]
        rz = sn * pts[:, 0] + cs * pts[:, 1]
        area = (rx.max() - rx.min()) * (rz.max() - rz.min())
        if area < best:
            best, best_yaw = area, yaw
    return best_yaw


def _monotone_chain(pts):
    idx = sorted(range(len(pts)), key=lambda i: (pts[i, 0], pts[i, 1]))

    def cross(o, a, b):
        return (pts[a, 0] - pts[o, 0]) * (pts[b, 1] - pts[o, 1]) - (pts[a, 1] - pts[o, 1]) * (pts[b, 0] - pts[o, 0])

    lower = []
    for i in idx:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], i) <= 0:
            lower.pop()
        lower.append(i)
    upper = []
    for i in reversed(idx):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], i) <= 0:
            upper.pop()
        upper.append(i)
    return lower[:-1] + upper[:-1]


# --------------------------------------------------------------------------------------
# A3  estimate_bbox  (src/util_3dbox.py:106-178)
# --------------------------------------------------------------------------------------
def ground_rotation(ground_equ):
    """src/util_3dbox.py:128-134: flip so that dot([0,-1,0], g) > 0, then Rodrigues."""
    if ground_equ is None:
        return np.eye(3)
    g = np.asarray(ground_equ, dtype=np.float64)
    if np.dot([0, -1, 0], g[:3]) <= 0:
        g = -g
    return rotation_matrix_from_vectors([0, -1, 0], g[:3])


def estimate_bbox(in_pc, cat_name=None, ground_equ=None, method="pca", rand_ind=None, return_aux=False):
    """src/util_3dbox.py:106-178.

    ``rand_ind``: the 500 indices the reference would draw at :124; ``None`` draws them
    from the global NumPy RNG exactly as the reference does (``np.random.randint(0, N,
    500)``), so seeding ``np.random.seed`` reproduces the reference stream.  Pass
    ``rand_ind=False`` to disable subsampling (full-cloud mode: the same code path the
    reference takes when N <= 500, applied to any N).
    Silent: the reference's per-box ``print`` (:162) is a side effect of the scalar shim,
    not of the arithmetic.
    """
    in_pc = np.asarray(in_pc)
    if rand_ind is not False and in_pc.shape[0] > SUBSAMPLE:
        if rand_ind is None:
            rand_ind = np.random.randint(0, in_pc.shape[0], SUBSAMPLE)
        in_pc = in_pc[np.asarray(rand_ind)]
    Rg = ground_rotation(ground_equ)
    with np.errstate(invalid="ignore", over="ignore"):
        rotated = np.dot(in_pc, Rg)  # :136  == (Rg^T p)^T
    rotated = rotated[~np.isnan(rotated).any(axis=1)]  # :139-140
    if len(rotated) == 0:
        raise ValueError("No valid points after removing NaN values")  # :143
    if method == "convex_hull":
        yaw = yaw_convex_hull(rotated)
    elif method == "pca":
        yaw = yaw_pca_closed_form(rotated)
    else:
        raise ValueError(f"Unknown method: {method}. Use 'pca' or 'convex_hull'")
    r2 = rotate_y(yaw) @ rotated.T  # :154
    x_min, x_max = r2[0].min(), r2[0].max()
    y_min, y_max = r2[1].min(), r2[1].max()
    z_min, z_max = r2[2].min(), r2[2].max()
    dx, dy, dz = x_max - x_min, y_max - y_min, z_max - z_min
    cx, cy, cz = (x_min + x_max) / 2, (y_min + y_max) / 2, (z_min + z_max) / 2
    with np.errstate(over="ignore", invalid="ignore"):
        verts = convert_box_vertices(cx, cy, cz, dx, dy, dz, 0).astype(np.float16)  # :165
        verts = np.dot(rotate_y(-yaw), verts.T).T  # :168
        verts = np.dot(verts, Rg.T)  # :169  (uses Rg)
        center_cam = Rg.T @ (rotate_y(-yaw) @ np.array([cx, cy, cz]))  # :172-173 (uses Rg^T)
        dimension = [dz, dy, dx]  # :175
        R_cam = Rg.T @ rotate_y(-yaw)  # :176
    if return_aux:
        return verts, center_cam, dimension, R_cam, dict(yaw=float(yaw), n_valid=len(rotated), kappa=pca_conditioning(rotated)[0], gap=pca_conditioning(rotated)[1])
    return verts, center_cam, dimension, R_cam


def pack39(verts, center, dims, R):
    """Box record layout shared with the C-ABI: center(3) dims(3) R_cam(9) verts(24)."""
    return np.concatenate([np.asarray(center, float), np.asarray(dims, float), np.asarray(R, float).ravel(),
                           np.asarray(verts, float).ravel()])


# --------------------------------------------------------------------------------------
# composed path  (glue = NumPy boolean indexing as in src/util.py:480-481)
# --------------------------------------------------------------------------------------
def fit_instance(depth, mask, K, ground=None, rand_ind=False, method="pca", refstyle=False):
    """One instance: estimate_bbox(depth_to_points(depth[None], K)[mask], None, ground).

    Returns (record39, status, aux).  Exceptions of the scalar reference path are mapped
    to the status codes of the batched C-ABI (include/la3d.h).
    """
    d2p = depth_to_points_refstyle if refstyle else depth_to_points
    pts = d2p(np.asarray(depth)[None], np.asarray(K, dtype=np.float64))[np.asarray(mask).astype(bool)]
    return fit_points(pts, ground, rand_ind, method)


def fit_points(pts, ground=None, rand_ind=False, method="pca"):
    rec = np.full(39, np.nan)
    aux = dict(yaw=float("nan"), n_valid=0, n_in=int(len(pts)))
    try:
        v, c, d, R, a = estimate_bbox(pts, None, ground, method, rand_ind=rand_ind, return_aux=True)
    except ValueError as e:
        msg = str(e)
        if "No valid points" in msg:
            g_bad = ground is not None and not np.isfinite(ground_rotation(ground)).all()
            return rec, (ST_BAD_GROUND if g_bad else ST_EMPTY), aux
        if "n_components" in msg:
            return rec, ST_TOO_FEW, aux
        if "infinity" in msg:
            return rec, ST_NONFINITE, aux
        raise
    aux.update(a)
    return pack39(v, c, d, R), ST_OK, aux


def fit_instances(depth, masks, K, ground=None, sample_idx=None, depth_index=None, method="pca", return_kappa=False):
    """Batched composition.  depth (P,H,W) or (H,W); masks (B,H,W); K (P,3,3) or (3,3).

    ``depth_index[n]`` selects the depth plane / K of instance n (default n, or 0 when a
    single plane is given).  ``sample_idx`` (B,500) int, rows ignored where the mask has
    <= 500 pixels; ``None`` = full-mask mode.
    Returns records (B,39) f64, status (B,) i32, yaw (B,), n_valid (B,) [, kappa (B,): pca_kappa of every fitted cloud].
    """
    depth = np.asarray(depth)
    if depth.ndim == 2:
        depth = depth[None]
    K = np.asarray(K, dtype=np.float64)
    if K.ndim == 2:
        K = K[None]
    masks = np.asarray(masks)
    B = masks.shape[0]
    out = np.full((B, 39), np.nan)
    status = np.zeros(B, np.int32)
    yaw = np.full(B, np.nan)
    nval = np.zeros(B, np.int64)
    kappa = np.full(B, np.nan)
    cache = {}
    for n in range(B):
        img = int(depth_index[n]) if depth_index is not None else (n if depth.shape[0] > 1 else 0)
        if img not in cache:
            cache.clear()
            cache[img] = depth_to_points(depth[img][None], K[img if K.shape[0] > 1 else 0])
        pts = cache[img][masks[n].astype(bool)]
        g = None if ground is None else ground[n]
        ri = False
        if sample_idx is not None and len(pts) > SUBSAMPLE:
            ri = np.asarray(sample_idx[n])
        out[n], status[n], aux = fit_points(pts, g, ri, method)
        yaw[n], nval[n] = aux["yaw"], aux["n_valid"]
        kappa[n] = aux.get("kappa", np.nan)
    if return_kappa:
        return out, status, yaw, nval, kappa
    return out, status, yaw, nval


# --------------------------------------------------------------------------------------
# A8  cam_utils  (src/cam_utils.py:4-52)
# --------------------------------------------------------------------------------------
def cam_length(x, eps=1e-20):
    """src/cam_utils.py:4-8 (NumPy branch)."""
    return np.sqrt(np.maximum(np.sum(x * x, axis=-1, keepdims=True), eps))


def cam_safe_normalize(x, eps=1e-20):
    """src/cam_utils.py:10-11."""
    return x / cam_length(x, eps)


def cam_look_at(campos, target, opengl=True):
    """src/cam_utils.py:14-31."""
    if not opengl:
        fwd = cam_safe_normalize(target - campos)
        up = np.array([0, 1, 0], dtype=np.float32)
        right = cam_safe_normalize(np.cross(fwd, up))
        up = cam_safe_normalize(np.cross(right, fwd))
    else:
        fwd = cam_safe_normalize(campos - target)
        up = np.array([0, 1, 0], dtype=np.float32)
        right = cam_safe_normalize(np.cross(up, fwd))
        up = cam_safe_normalize(np.cross(fwd, right))
    return np.stack([right, up, fwd], axis=1)


def cam_orbit_camera(elevation, azimuth, radius=1, is_degree=True, target=None, opengl=True):
    """src/cam_utils.py:35-52 — 4x4 float32 cam2world pose."""
    if is_degree:
        elevation = np.deg2rad(elevation)
        azimuth = np.deg2rad(azimuth)
    x = radius * np.cos(elevation) * np.sin(azimuth)
    y = -radius * np.sin(elevation)
    z = radius * np.cos(elevation) * np.cos(azimuth)
    if target is None:
        target = np.zeros([3], dtype=np.float32)
    campos = np.array([x, y, z]) + target
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = cam_look_at(campos, target, opengl)
    T[:3, 3] = campos
    return T


# --------------------------------------------------------------------------------------
# §8f-1  mask ingestion: COCO run-length masks and the reference's instance filters
# --------------------------------------------------------------------------------------
# Third-party piece: pycocotools (COCO API, `mask_utils.decode / frPyObjects`, called at reference
# src/util.py:367,401-402; absent from /root/reference and from this image).  Its published format:
# `counts` are run lengths over the (h, w) mask in COLUMN-MAJOR order, alternating zeros / ones and
# starting with zeros; the compressed string form stores each count in 5-bit groups (char - 48, bit 5 =
# continuation, bit 4 of the last group = sign) with counts beyond the third stored as a difference to
# the count two places earlier (maskApi.c rleToString / rleFrString).  The reference's own encoder,
# binary_mask_to_rle (src/download_coconut.py:167-175), produces the uncompressed list form and is the
# pinning call site: tests/golden/g8_masks.npz holds its outputs.
def rle_encode(mask):
    """src/download_coconut.py:167-175 — uncompressed RLE {'counts', 'size'} of a binary mask."""
    m = np.asarray(mask).astype(bool)
    flat = m.ravel(order="F")
    if flat.size == 0:
        return {"counts": [], "size": list(m.shape)}
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(bounds).tolist()
    if flat[0]:
        counts = [0] + counts
    return {"counts": counts, "size": list(m.shape)}


def rle_decode(counts, h, w):
    """pycocotools rleDecode: column-major runs -> (h, w) bool."""
    counts = np.asarray(counts, dtype=np.int64)
    vals = (np.arange(len(counts)) & 1).astype(bool)
    flat = np.repeat(vals, counts)[: h * w]
    if flat.size < h * w:
        flat = np.concatenate([flat, np.zeros(h * w - flat.size, bool)])
    return flat.reshape((w, h)).T.copy()


def rle_to_string(counts):
    """pycocotools rleToString (maskApi.c) — compressed ASCII form of the counts."""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s):
    """pycocotools rleFrString (maskApi.c)."""
    if isinstance(s, str):
        s = s.encode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def mask_stats(mask, boundary_threshold=10):
    """The quantities the reference's instance filter looks at (src/util.py:291-335, :367-376):
    area = mask.sum(); rows = number of rows holding a pixel (the RLE branch's `height`, :368-369);
    span = last row - first row + 1 (get_maximum_height, the polygon branch's height, :328-335);
    trunc = pixels inside the four boundary strips of `boundary_threshold` px, corners counted twice
    (analyze_mask :303-322)."""
    m = np.asarray(mask).astype(bool)
    rows_any = m.any(axis=1)
    idx = np.flatnonzero(rows_any)
    span = int(idx[-1] - idx[0] + 1) if idx.size else 0
    b = boundary_threshold
    trunc = int(m[:b].sum() + m[-b:].sum() + m[:, :b].sum() + m[:, -b:].sum())
    return int(m.sum()), int(rows_any.sum()), span, trunc


def keep_instance(stats, image_height, from_rle, scale_threshold=100):
    """src/util.py:375 — height/H > 0.0625 and not truncated (trunc < 10) and area >= 100."""
    area, rows, span, trunc = stats
    height = rows if from_rle else span
    return (height / image_height > 0.0625) and not (trunc >= 10) and (area >= scale_threshold)


# --------------------------------------------------------------------------------------
# §8f-2  box consumers: corner projection, 2-D boxes, IoU cost matrix  (src/tools/combine_results.py)
# --------------------------------------------------------------------------------------
def project_boxes(records, K, image_size):
    """src/tools/combine_results.py:105-108, :238-252 — project the 8 corners of each 39-double record with K
    (p2d = (K @ P)[:2] / (K @ P)[2]); bbox2D_proj = [min_x, min_y, max_x, max_y]; bbox2D_trunc clamps it to
    [0, W] x [0, H].  records (B,39), K (3,3) or (B,3,3), image_size (W, H) -> (B,8)."""
    rec = np.asarray(records, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    Wd, Hd = image_size
    out = np.empty((len(rec), 8))
    for i, r in enumerate(rec):
        Ki = K if K.ndim == 2 else K[i]
        corners = r[15:39].reshape(8, 3)
        with np.errstate(invalid="ignore", divide="ignore"):
            pts = [np.dot(Ki, p)[:2] / np.dot(Ki, p)[2] for p in corners]
        min_x, min_y = min(p[0] for p in pts), min(p[1] for p in pts)
        max_x, max_y = max(p[0] for p in pts), max(p[1] for p in pts)
        out[i] = [min_x, min_y, max_x, max_y, max(0, min_x), max(0, min_y), min(Wd, max_x), min(Hd, max_y)]
    return out


def iou2d_matrix(boxes0, boxes1):
    """src/tools/combine_results.py:111-124 (iou2D) for every pair: the negated entries are the Hungarian cost
    matrix of :131-135."""
    b0, b1 = np.asarray(boxes0, float), np.asarray(boxes1, float)
    out = np.zeros((len(b0), len(b1)))
    for i, a in enumerate(b0):
        for j, b in enumerate(b1):
            x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
            inter = max(0, x2 - x1) * max(0, y2 - y1)
            out[i, j] = inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-6)
    return out

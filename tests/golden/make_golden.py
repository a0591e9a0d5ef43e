"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden.py

Imports /root/reference/src/{util,util_3dbox,cam_utils}.py read-only (see _refimport.py),
feeds seeded inputs, and stores inputs + the reference's outputs.  Only data is written —
no reference source travels.  No-op (exit 0 with a message) where /root/reference is absent.
Fixture groups follow SURVEY.md §8c (G1..G7).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

K48 = np.array([[50.0, 0, 32], [0, 50.0, 24], [0, 0, 1]])
K640 = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def rec39(out):
    v, c, d, R = out
    return np.concatenate([np.asarray(c, float), np.asarray(d, float), np.asarray(R, float).ravel(), np.asarray(v, float).ravel()])


def run_bbox(ref, pc, ground=None, method="pca"):
    """-> (record39 or NaNs, exception type name or '', message)."""
    try:
        with np.errstate(all="ignore"):
            out = quiet(ref.util_3dbox.estimate_bbox, pc, None, ground, method)
        return rec39(out), "", ""
    except Exception as e:  # noqa: BLE001 - we record whatever the reference raises
        return np.full(39, np.nan), type(e).__name__, str(e)


def cloud(rs, n, scale=(1.0, 0.3, 0.5), yaw=0.3, center=(0.5, -0.2, 5.0)):
    p = rs.randn(n, 3) * np.array(scale)
    c, s = np.cos(yaw), np.sin(yaw)
    Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return p @ Ry.T + np.array(center)


def main():
    ref = _refimport.load()
    if ref is None:
        print("reference not present; nothing generated")
        return 0
    import sklearn

    meta = dict(numpy=np.__version__, sklearn=sklearn.__version__)

    # ---------------- G1 depth_to_points ----------------
    g1 = {}
    rs = np.random.RandomState(11)
    d45 = (np.arange(20, dtype=np.float32).reshape(1, 4, 5) + 1) / 4
    Ka = np.array([[2.0, 0, 2], [0, 2.0, 1.5], [0, 0, 1]])
    g1["a_depth"], g1["a_K"] = d45, Ka
    g1["a_out"] = ref.util.depth_to_points(d45, Ka)
    d48 = rs.uniform(0.5, 10, (1, 48, 64)).astype(np.float32)
    g1["b_depth"], g1["b_K"] = d48, K48
    g1["b_out"] = ref.util.depth_to_points(d48, K48)
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]) @ np.array(
        [[1, 0, 0], [0, np.cos(0.2), -np.sin(0.2)], [0, np.sin(0.2), np.cos(0.2)]])
    t = np.array([0.1, -0.2, 0.3])
    g1["c_R"], g1["c_t"] = R, t
    g1["c_out"] = ref.util.depth_to_points(d48, K48, R, t)
    d2 = rs.uniform(0.5, 10, (2, 6, 8)).astype(np.float32)
    g1["d_depth"] = d2
    g1["d_out"] = ref.util.depth_to_points(d2, K48)  # only batch element 0 comes back
    Ks = np.array([[480.0, 3.5, 310.2], [0, 505.5, 236.7], [0, 0, 1]])  # skewed K
    g1["e_K"] = Ks
    g1["e_out"] = ref.util.depth_to_points(d48, Ks)
    # 480x640: seed + 1k sampled pixels + sums
    rs640 = np.random.RandomState(0)
    d640 = rs640.uniform(0.5, 10, (1, 480, 640)).astype(np.float32)
    full = ref.util.depth_to_points(d640, K640)
    pick = np.random.RandomState(1).randint(0, 480 * 640, 1000)
    g1["f_pick"] = pick
    g1["f_out_pick"] = full.reshape(-1, 3)[pick]
    g1["f_sum"] = full.sum(axis=(0, 1))
    g1["f_abs_sum"] = np.abs(full).sum(axis=(0, 1))
    # non-finite depth
    dn = d48.copy()
    dn[0, 3, 4] = np.nan
    dn[0, 5, 6] = np.inf
    dn[0, 7, 8] = -np.inf
    g1["g_depth"] = dn
    with np.errstate(all="ignore"):
        g1["g_out"] = ref.util.depth_to_points(dn, K48)
    np.savez_compressed(os.path.join(HERE, "g1_depth_to_points.npz"), **g1)

    # ---------------- G2 estimate_bbox deterministic (N <= 500) ----------------
    rs = np.random.RandomState(22)
    pcs, grounds, methods, outs, tags = [], [], [], [], []

    def add(tag, pc, ground=None, method="pca"):
        r, en, em = run_bbox(ref, pc, ground, method)
        assert en == "", (tag, en, em)
        pcs.append(np.asarray(pc))
        grounds.append(np.full(4, np.nan) if ground is None else np.asarray(ground, float))
        methods.append(method)
        outs.append(r)
        tags.append(tag)

    for n in (2, 3, 5, 19, 20, 21, 100, 500):
        add(f"n{n}", cloud(rs, n))
    for k, yaw in enumerate(np.linspace(-3.0, 3.0, 13)):
        add(f"yawsweep{k}", cloud(rs, 200, yaw=yaw))
    for k, r in enumerate((0.01, 0.1, 0.5, 0.9, 0.99, 0.999)):
        add(f"aniso{k}", cloud(rs, 400, scale=(1.0, 0.3, r), yaw=0.7))
    add("f32", cloud(rs, 300).astype(np.float32))
    pn = cloud(rs, 50)
    pn[3, 1] = np.nan
    pn[10, :] = np.nan
    add("nanrows", pn)
    pi = cloud(rs, 50)
    pi[7, 0] = np.inf
    add("infrow_noground", pi)  # inf*0 -> NaN through the identity rotation -> dropped
    g_a = np.array([0.1, -0.95, 0.2, 1.3])
    g_b = np.array([0.05, 0.9, -0.3, -2.0])  # dot([0,-1,0],g) <= 0 -> flipped
    g_c = np.array([0.3, -2.0, 0.1, 0.0]) * 3.7  # unnormalised
    for k, g in enumerate((g_a, g_b, g_c)):
        add(f"ground{k}", cloud(rs, 300, yaw=0.2 + k), g)
        add(f"ground{k}_small", cloud(rs, 7), g)
    add("ground_nan", pn, g_a)
    for k in range(4):
        add(f"hull{k}", cloud(rs, 60 + 40 * k, yaw=0.4 * k - 0.5), None, "convex_hull")
    add("hull_ground", cloud(rs, 120, yaw=1.1), g_a, "convex_hull")
    # far-away object: stresses raw-moment cancellation
    add("far", cloud(rs, 400, scale=(0.2, 0.1, 0.05), yaw=1.0, center=(30.0, 2.0, 80.0)))
    maxn = max(len(p) for p in pcs)
    P = np.full((len(pcs), maxn, 3), np.nan)
    L = np.array([len(p) for p in pcs])
    for i, p in enumerate(pcs):
        P[i, : len(p)] = p
    np.savez_compressed(os.path.join(HERE, "g2_estimate_bbox.npz"), pcs=P, lens=L, grounds=np.array(grounds),
                        methods=np.array(methods), outs=np.array(outs), tags=np.array(tags),
                        f32_case=np.array([tags.index("f32")]))

    # ---------------- G3 error cases ----------------
    errs = {}
    e_cases = {
        "empty": (np.zeros((0, 3)), None, "pca"),
        "allnan": (np.full((5, 3), np.nan), None, "pca"),
        "n1": (cloud(rs, 1), None, "pca"),
        "ground_down": (cloud(rs, 30), np.array([0, -1.0, 0, 0.5]), "pca"),
        "ground_up": (cloud(rs, 30), np.array([0, 1.0, 0, 0.5]), "pca"),
        "ground_zero": (cloud(rs, 30), np.array([0, 0.0, 0, 0.5]), "pca"),
        "badmethod": (cloud(rs, 30), None, "nope"),
        "inf_ground": (pi, g_a, "pca"),
    }
    for name, (pc, g, m) in e_cases.items():
        r, en, em = run_bbox(ref, pc, g, m)
        errs[name + "_pc"] = pc
        errs[name + "_ground"] = np.full(4, np.nan) if g is None else g
        errs[name + "_method"] = np.array(m)
        errs[name + "_exc"] = np.array(en)
        errs[name + "_msg"] = np.array(em)
        errs[name + "_out"] = r
    errs["names"] = np.array(list(e_cases))
    np.savez_compressed(os.path.join(HERE, "g3_errors.npz"), **errs)

    # ---------------- G4 subsample mode (N > 500, global RNG) ----------------
    g4 = {}
    seeds, ns = [], []
    for j, (seed, n) in enumerate(((5, 501), (6, 5000), (7, 100000))):
        pc = cloud(np.random.RandomState(100 + j), n, yaw=0.5 * j - 0.3)
        np.random.seed(seed)
        idx = np.random.randint(0, n, 500)
        np.random.seed(seed)
        r, en, _ = run_bbox(ref, pc)
        assert en == ""
        g4[f"s{j}_cloud_seed"] = np.array(100 + j)
        g4[f"s{j}_yaw"] = np.array(0.5 * j - 0.3)
        g4[f"s{j}_n"] = np.array(n)
        g4[f"s{j}_idx"] = idx
        g4[f"s{j}_out"] = r
        g4[f"s{j}_rng_seed"] = np.array(seed)
        if n <= 5000:
            g4[f"s{j}_pc"] = pc
    # three consecutive calls: pins the order in which the global stream is consumed
    np.random.seed(77)
    seq_out, seq_n = [], (800, 300, 1200)  # the 300-point call must NOT consume the stream
    for j, n in enumerate(seq_n):
        pc = cloud(np.random.RandomState(200 + j), n, yaw=0.1 + j)
        r, en, _ = run_bbox(ref, pc)
        seq_out.append(r)
    g4["seq_n"] = np.array(seq_n)
    g4["seq_out"] = np.array(seq_out)
    np.savez_compressed(os.path.join(HERE, "g4_subsample.npz"), **g4)

    # ---------------- G5 composed path: depth + masks -> 39-vector ----------------
    rs = np.random.RandomState(55)
    H, W = 48, 64
    # smooth-ish depth: a tilted plane plus noise, so objects are compact in 3D
    vv, uu = np.mgrid[0:H, 0:W]
    depth = (3.0 + 0.02 * uu + 0.03 * vv + 0.05 * rs.randn(H, W)).astype(np.float32)
    masks = np.zeros((8, H, W), bool)
    rects = [(2, 3, 10, 20), (20, 30, 25, 30), (0, 0, 48, 64), (5, 40, 3, 3), (30, 2, 8, 40), (10, 10, 1, 2)]
    for i, (r0, c0, h, w) in enumerate(rects):
        masks[i, r0:r0 + h, c0:c0 + w] = True
    masks[6] = ((uu - 30) ** 2 / 200.0 + (vv - 20) ** 2 / 60.0) < 1.0  # ellipse
    masks[7] = rs.rand(H, W) < 0.05  # scattered
    gr = np.array([[0.1, -0.95, 0.2, 1.3]] * 8) + 0.05 * rs.randn(8, 4)
    pts = ref.util.depth_to_points(depth[None], K48)
    out_ng, out_g, idx_fixed = [], [], []
    for i in range(8):
        # masks above 500 px make the reference subsample (:123-125): seed per call and record
        # the indices it draws, so the comparison is deterministic
        n = int(masks[i].sum())
        np.random.seed(1000 + i)
        idx_fixed.append(np.random.randint(0, n, 500) if n > 500 else np.zeros(500, np.int64))
        np.random.seed(1000 + i)
        r, en, _ = run_bbox(ref, pts[masks[i]])
        assert en == "", i
        out_ng.append(r)
        np.random.seed(1000 + i)
        r, en, _ = run_bbox(ref, pts[masks[i]], gr[i])
        assert en == "", i
        out_g.append(r)
    # subsample mode on the composed path: masks with > 500 px draw from the global stream
    np.random.seed(9)
    idx_rows, out_s = [], []
    for i in range(8):
        n = int(masks[i].sum())
        st = np.random.get_state()
        if n > 500:
            idx_rows.append(np.random.randint(0, n, 500))
            np.random.set_state(st)
        else:
            idx_rows.append(np.zeros(500, np.int64))
        r, en, _ = run_bbox(ref, pts[masks[i]], gr[i])
        out_s.append(r)
    np.savez_compressed(os.path.join(HERE, "g5_composed.npz"), depth=depth, masks=masks, K=K48, ground=gr,
                        out_noground=np.array(out_ng), out_ground=np.array(out_g), idx_fixed=np.array(idx_fixed),
                        sample_idx=np.array(idx_rows), out_sampled=np.array(out_s))

    # ---------------- G6 helpers + cam_utils ----------------
    g6 = {}
    yaws = np.linspace(-4, 4, 9)
    g6["rot_yaws"] = yaws
    g6["rot_out"] = np.array([ref.util_3dbox.rotate_y(y) for y in yaws])
    v1 = rs.randn(6, 3)
    v2 = rs.randn(6, 3)
    g6["rm_v1"], g6["rm_v2"] = v1, v2
    g6["rm_out"] = np.array([ref.util_3dbox.rotation_matrix_from_vectors(a, b) for a, b in zip(v1, v2)])
    with np.errstate(all="ignore"):
        g6["rm_parallel"] = ref.util_3dbox.rotation_matrix_from_vectors([0, -1, 0], [0, -2.0, 0])
        g6["rm_antiparallel"] = ref.util_3dbox.rotation_matrix_from_vectors([0, -1, 0], [0, 3.0, 0])
    g6["norm_zero"] = ref.util_3dbox.normalize(np.zeros(3))
    g6["norm_in"] = v1[0]
    g6["norm_out"] = ref.util_3dbox.normalize(v1[0])
    cbv_in = np.array([[0.5, -0.2, 5.0, 2.0, 1.0, 0.5, 0.0], [1.5, 0.3, 3.0, 0.7, 1.9, 2.5, 0.8], [-2, 1, 9, 1, 1, 1, -2.5]])
    g6["cbv_in"] = cbv_in
    g6["cbv_out"] = np.array([ref.util_3dbox.convert_box_vertices(*row) for row in cbv_in])
    g6["p2p_in"] = np.array([0.1, -0.9, 0.3, 1.5, 0.4, 0.5, 6.0])
    g6["p2p_out"] = np.array(ref.util_3dbox.point_to_plane_distance(g6["p2p_in"][:4], *g6["p2p_in"][4:]))
    elev = np.array([-80.0, -30, 0, 15, 60])
    azim = np.array([-170.0, -90, 0, 45, 120])
    for ogl in (True, False):
        poses = np.array([[ref.cam_utils.orbit_camera(e, a, radius=2.5, opengl=ogl) for a in azim] for e in elev])
        g6[f"orbit_opengl{int(ogl)}"] = poses
        assert poses.dtype == np.float32
    g6["orbit_elev"], g6["orbit_azim"] = elev, azim
    g6["orbit_rad_target"] = ref.cam_utils.orbit_camera(0.3, -1.1, radius=1.7, is_degree=False,
                                                        target=np.array([0.5, 0.1, -0.2], dtype=np.float32))
    cp = rs.randn(5, 3).astype(np.float32) * 2
    tg = rs.randn(5, 3).astype(np.float32) * 0.1
    g6["look_campos"], g6["look_target"] = cp, tg
    g6["look_opengl1"] = ref.cam_utils.look_at(cp, tg, True)
    g6["look_opengl0"] = ref.cam_utils.look_at(cp, tg, False)
    g6["length_in"] = np.array([[3.0, 4.0, 0.0], [0, 0, 0], [1e-12, 0, 0]])
    g6["length_out"] = ref.cam_utils.length(g6["length_in"])
    g6["safe_norm_out"] = ref.cam_utils.safe_normalize(g6["length_in"])
    np.savez_compressed(os.path.join(HERE, "g6_helpers.npz"), **g6)

    # ---------------- G7 tie cases (recorded as observed; solver dependent) ----------------
    g7 = {}
    cross4 = np.array([[1.0, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]]) + [0, 0, 5]
    sq20 = np.array([[np.cos(a), 0.1 * np.sin(3 * a), np.sin(a)] for a in np.arange(20) * 2 * np.pi / 20]) + [0, 0, 5]
    sq24 = np.array([[np.cos(a), 0.1 * np.sin(3 * a), np.sin(a)] for a in np.arange(24) * 2 * np.pi / 24]) + [0, 0, 5]
    same25 = np.tile(np.array([[0.3, 0.1, 4.0]]), (25, 1))
    grid = np.array([[x, 0.1 * (x + z), z] for x in (-1.0, 0, 1) for z in (-1.0, 0, 1)] * 3) + [0, 0, 5]  # a==c, b==0, n=27
    for name, pc in (("cross4", cross4), ("ring20", sq20), ("ring24", sq24), ("same25", same25), ("grid27", grid)):
        r, en, em = run_bbox(ref, pc)
        g7[name + "_pc"] = pc
        g7[name + "_out"] = r
        g7[name + "_exc"] = np.array(en)
    np.savez_compressed(os.path.join(HERE, "g7_ties.npz"), **g7)

    with open(os.path.join(HERE, "VERSIONS.txt"), "w") as f:
        f.write("fixtures generated by tests/golden/make_golden.py from /root/reference (LabelAny3D @ 2026-02-13)\n")
        for k, v in meta.items():
            f.write(f"{k}=={v}\n")
    print("golden fixtures written to", HERE)
    return 0


if __name__ == "__main__":
    sys.exit(main())

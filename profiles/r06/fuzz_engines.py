"""Round 6: a randomised differential campaign over every engine of the batched fit (the code paths new this round - eight-band
exchange, the one-launch row engine, the full-tile class of the instance engine - get the same inputs as the old ones).

    python profiles/r06/fuzz_engines.py [--cases 400] [--seed 0] [--workers 128] [--out profiles/r06/fuzz_engines.txt]

One CASE = one call of the u8-plane entry (la3d_fit_instances through labelany3d_amd.fit_instances): a random frame size (widths
that are not a multiple of 32 and heights that are not a multiple of 8 included), a random batch size (1 ... 300), depth planes
private / shared / indexed, intrinsics with and without skew, masks of every shape the tests know (rectangles, rectangles aligned
to the 32x8 tiles, ellipses, sparse noise, whole frame, empty, one pixel, one row, one column, two distant blobs, checkerboards;
mask bytes 1 / 255 / anything non-zero), depth that is smooth / random / constant with non-finite, zero and negative pixels
sprinkled in, ground planes for all / some / none of the instances (degenerate ones included), full-mask and reference-subsample
mode.  The SAME case runs with the default dispatch and pinned to every engine (instance, band, rows, rows2, split), with the
plain and the no-cull build and with the launch order off; EVERY record of EVERY run is compared with the CPU oracle
(oracle/la3d_oracle.py, computed beforehand on the host cores): status, n_valid, n_masked exactly, center / dims / R / corners by
tests/test_gpu_parity.py::assert_records' rule (1e-9 of the scale, the axis conditioned by the eigen-gap; for clouds of 20 and more
points, where the reference itself works from raw sums, its own rounding noise ~2^-52 kappa / gap on top: reference_axis_noise).
Records whose reported eigen-gap is below 1e-9 (exact ties and clouds without any spread: the documented don't-care value)
are counted and held to status / counts.

The oracle is test infrastructure: it is the checker here.  Nothing under /root/reference is read."""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HS = [8, 16, 24, 37, 64, 96, 120, 200, 240, 375, 480, 517]
WS = [32, 64, 96, 128, 160, 250, 320, 333, 427, 500, 640, 672]
BS = [1, 1, 2, 3, 5, 8, 13, 16, 17, 33, 64, 100, 129, 150, 161, 200, 300]
TINY_HS, TINY_WS = [1, 2, 3, 5, 7, 8, 9], [1, 2, 5, 17, 31, 32, 33, 40]
if os.environ.get("LA3D_FUZZ_TINY"):
    HS, WS = list(TINY_HS), list(TINY_WS)


def one_mask(rs, H, W):
    kind = rs.randint(0, 14)
    m = np.zeros((H, W), bool)
    if kind in (0, 1):                       # rectangle
        h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
        r, c = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        m[r:r + h, c:c + w] = True
    elif kind == 2:                          # rectangle aligned to the 32x8 tiles (every tile completely inside the mask)
        th, tw = rs.randint(1, max(H // 8, 1) + 1), rs.randint(1, max(W // 32, 1) + 1)
        r, c = 8 * rs.randint(0, max(H // 8 - th, 0) + 1), 32 * rs.randint(0, max(W // 32 - tw, 0) + 1)
        m[r:r + 8 * th, c:c + 32 * tw] = True
    elif kind in (3, 4):                     # ellipse
        yy, xx = np.mgrid[:H, :W]
        cy, cx = rs.uniform(0, H), rs.uniform(0, W)
        a, b = rs.uniform(1, H / 2 + 1), rs.uniform(1, W / 2 + 1)
        m = ((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2 <= 1.0
    elif kind == 5:                          # sparse noise
        m = rs.rand(H, W) < 10 ** rs.uniform(-3, -0.3)
    elif kind == 6:                          # the whole frame
        m[:] = True
    elif kind == 7:                          # empty (status 1)
        pass
    elif kind == 8:                          # one pixel / two pixels / three pixels
        for _ in range(rs.randint(1, 4)):
            m[rs.randint(H), rs.randint(W)] = True
    elif kind == 9:                          # one row (part of it)
        c0 = rs.randint(0, W)
        m[rs.randint(H), c0:rs.randint(c0, W) + 1] = True
    elif kind == 10:                         # one column
        r0 = rs.randint(0, H)
        m[r0:rs.randint(r0, H) + 1, rs.randint(W)] = True
    elif kind == 11:                         # two distant blobs
        for _ in range(2):
            h, w = rs.randint(1, max(H // 4, 1) + 1), rs.randint(1, max(W // 4, 1) + 1)
            r, c = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            m[r:r + h, c:c + w] = True
    elif kind == 12:                         # checkerboard inside a rectangle
        h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
        r, c = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        yy, xx = np.mgrid[:H, :W]
        s = rs.randint(1, 9)
        m[r:r + h, c:c + w] = (((yy // s) + (xx // s)) % 2 == 0)[r:r + h, c:c + w]
    else:                                    # the frame without a hole
        m[:] = True
        h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
        r, c = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        m[r:r + h, c:c + w] = False
    one_mask.kind = kind
    return m


def one_polygon_mask(rs, H, W):
    """A polygon annotation (1-3 parts, 1-24 vertices each, fractional coordinates, some outside the frame, some degenerate) and
    the mask the reference's cv2.fillPoly gives for it (oracle/poly_oracle.py)."""
    from oracle import poly_oracle as P

    seg = []
    for _ in range(rs.randint(1, 4)):
        nv = int(rs.choice([1, 2, 3, 4, 5, 8, 12, 24]))
        cx, cy = rs.uniform(-0.1 * W, 1.1 * W), rs.uniform(-0.1 * H, 1.1 * H)
        ang = np.sort(rs.uniform(0, 2 * np.pi, nv))
        rad = rs.uniform(0.5, 0.45 * min(H, W)) * rs.uniform(0.3, 1.0, nv)
        xy = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
        if rs.rand() < 0.3:
            xy = np.round(xy)
        seg.append([float(v) for v in xy.ravel()])
    m, _ = P.create_boolean_mask_from_polygon((W, H), seg)
    return m, seg


def one_plane(rs, H, W):
    kind = rs.randint(0, 5)
    yy, xx = np.mgrid[:H, :W]
    if kind == 0:
        d = rs.uniform(0.5, 10, (H, W))
    elif kind == 1:                          # a slanted plane with ripples
        d = 3 + rs.uniform(-2, 2) * yy / H + rs.uniform(-2, 2) * xx / W + 0.2 * np.sin(xx / rs.uniform(3, 40)) * np.cos(yy / rs.uniform(3, 40))
    elif kind == 2:                          # constant
        d = np.full((H, W), rs.uniform(0.3, 50))
    elif kind == 3:                          # smooth + noise
        d = 5 + 2 * np.sin(xx / 50.0 + rs.uniform(0, 6)) + 0.05 * rs.randn(H, W)
    else:                                    # large dynamic range
        d = 10 ** rs.uniform(-2, 3, (H, W))
    d = d.astype(np.float32)
    if rs.rand() < 0.4:                      # invalid pixels sprinkled in
        rate = 10 ** rs.uniform(-4, -1)
        bad = rs.rand(H, W) < rate
        vals = np.array([np.nan, np.inf, -np.inf, 0.0, -1.0, -0.0], np.float32)
        d[bad] = vals[rs.randint(0, len(vals), int(bad.sum()))]
    if rs.rand() < 0.03:
        d[:] = np.nan                        # nothing valid anywhere
    one_plane.kind = kind
    return d


def make_case(seed):
    rs = np.random.RandomState(seed)
    H, W = HS[rs.randint(len(HS))], WS[rs.randint(len(WS))]
    B = BS[rs.randint(len(BS))]
    poly_case = rs.rand() < 0.3              # every mask of the case is a polygon annotation: the case also runs through the polygon entry
    if poly_case:                            # (the oracle's rasteriser is a Python loop: smaller frames, fewer instances)
        H, W, B = min(H, 240), min(W, 333), min(B, 33)
    elif rs.rand() < 0.08:                   # frames whose bit image does not fit the workgroup's LDS share (above 640 x 480): the untiled forms
        H, W = [(720, 1280), (600, 800), (1080, 1920), (481, 641), (1080, 1923)][rs.randint(5)]
        B = int(rs.choice([1, 2, 5]))
    while B * H * W > 24_000_000 and B > 1:
        B = max(1, B // 2)
    mode = rs.randint(0, 3)                  # 0: one shared plane, 1: private planes, 2: P planes + image_index
    P = 1 if mode == 0 else (B if mode == 1 else rs.randint(1, B + 1))
    if P * H * W > 12_000_000:
        P = max(1, 12_000_000 // (H * W))
        mode = 2 if P > 1 else 0
    depth, dkind = [], []
    for _ in range(P):
        depth.append(one_plane(rs, H, W)); dkind.append(one_plane.kind)
    depth = np.stack(depth)
    image_index = rs.randint(0, P, B).astype(np.int32) if mode == 2 or (mode == 1 and P != B) else None
    if P == 1:
        image_index = None
    K = np.zeros((P, 3, 3))
    skew = rs.rand() < 0.25
    for p in range(P):
        f = rs.uniform(0.4, 3.0) * W
        K[p] = [[f, rs.uniform(-5, 5) if skew else 0.0, W / 2 + rs.uniform(-0.3, 0.3) * W], [0, f * rs.uniform(0.8, 1.25), H / 2 + rs.uniform(-0.3, 0.3) * H], [0, 0, 1]]
    if rs.rand() < 0.5:
        K[:] = K[0]
    masks, mkind, segs = [], [], []
    for _ in range(B):
        if poly_case:
            m, seg = one_polygon_mask(rs, H, W)
            masks.append(m); mkind.append(14); segs.append(seg)
        else:
            masks.append(one_mask(rs, H, W)); mkind.append(one_mask.kind)
    masks = np.stack(masks)
    mb = masks.astype(np.uint8)
    bytes_kind = rs.randint(0, 3)
    if bytes_kind == 1:
        mb *= 255
    elif bytes_kind == 2:
        mb = np.where(masks, rs.randint(1, 256, masks.shape), 0).astype(np.uint8)
    gk = rs.randint(0, 5)
    ground = None
    if gk >= 2:
        ground = np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.05 * rs.randn(B, 4)
        if gk == 4:
            for n in range(B):
                r = rs.rand()
                if r < 0.25:
                    ground[n, 0] = np.nan            # "no ground" for this instance
                elif r < 0.32:
                    ground[n] = [0, -1, 0, 1.0]      # already aligned: the reference's degenerate case (status 2)
                elif r < 0.36:
                    ground[n, :3] = 0.0
    sample = rs.rand() < 0.2
    sidx = None
    if sample:
        counts = masks.reshape(B, -1).sum(1)
        sidx = np.zeros((B, 500), np.int32)
        for n, c in enumerate(counts):
            if c > 500:
                sidx[n] = rs.randint(0, int(c), 500)
    return dict(seed=seed, H=H, W=W, B=B, P=P, depth=depth, K=K, masks=masks, mb=mb, ground=ground, image_index=image_index, sidx=sidx,
                skew=skew, mkind=mkind, dkind=dkind, segs=segs if poly_case else None)


def oracle_case(seed):
    from oracle import la3d_oracle as O

    c = make_case(seed)
    g = None if c["ground"] is None else [None if np.isnan(r[0]) else r for r in c["ground"]]
    di = c["image_index"]
    if di is not None:                       # the oracle caches one plane at a time: visit the instances plane by plane
        order = np.argsort(di, kind="stable")
        rec = np.full((c["B"], 39), np.nan); st = np.zeros(c["B"], np.int32); nv = np.zeros(c["B"], np.int64)
        kap = np.full(c["B"], np.nan)
        r_, s_, _, n_, k_ = O.fit_instances(c["depth"], c["masks"][order], c["K"], ground=None if g is None else [g[i] for i in order],
                                            sample_idx=None if c["sidx"] is None else c["sidx"][order], depth_index=di[order], return_kappa=True)
        rec[order], st[order], nv[order], kap[order] = r_, s_, n_, k_
    else:
        rec, st, _, nv, kap = O.fit_instances(c["depth"], c["masks"], c["K"], ground=g, sample_idx=c["sidx"], return_kappa=True)
    return seed, rec, st, nv, kap


# the annotation entries (run lengths decoded / polygons rasterised inside the fit kernel): the same masks as run lengths for every
# case, as polygons for the polygon cases
ANN_RUNS = [dict(entry="rle"), dict(entry="rle", engine="instance"), dict(entry="rle", engine="split"),
            dict(entry="poly"), dict(entry="poly", engine="instance"), dict(entry="poly", engine="split"),
            # la3d_fit_instances_ex with everything on: the records' 2-D boxes from the record epilogue, an area hint for the launch
            # order (right for some instances, wrong for others), and - run lengths - the reference's filter fused into the launch
            dict(entry="ex_u8"), dict(entry="ex_rle"), dict(entry="ex_rle", engine="split")]
RUNS = [dict(), dict(engine="instance"), dict(engine="band"), dict(engine="rows"), dict(engine="rows2"), dict(engine="split"),
        dict(build="plain"), dict(build="nocull"), dict(engine="instance", launch_order=False), dict(engine="band", launch_order=False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--seeds", default="", help="comma-separated seeds instead of --seed / --cases")
    ap.add_argument("--tiny", action="store_true", help="frames of 1 ... 9 rows and 1 ... 40 columns (below one tile in either direction)")
    ap.add_argument("--workers", type=int, default=min(128, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "fuzz_engines.txt"))
    a = ap.parse_args()
    seeds = [int(x) for x in a.seeds.split(",")] if a.seeds else list(range(a.seed, a.seed + a.cases))
    if a.tiny:   # (make_case reads the module's lists; the spawned oracle workers get the same through the environment)
        os.environ["LA3D_FUZZ_TINY"] = "1"
        HS[:] = TINY_HS; WS[:] = TINY_WS
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # one thread per oracle worker (the pool is the parallelism)
        os.environ[v] = "1"
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.workers) as pool:
        ref = {s: (r, st, nv, kp) for s, r, st, nv, kp in pool.imap_unordered(oracle_case, seeds, chunksize=1)}
    t_or = time.time() - t0

    import torch

    import labelany3d_amd as la
    from labelany3d_amd.masks import fit_instances_ex, fit_instances_poly, fit_instances_rle, pack_polygons
    from labelany3d_amd.options import scheduling
    from oracle import la3d_oracle as O
    from tests.test_gpu_parity import assert_records, reference_axis_noise

    assert torch.cuda.is_available(), "the campaign needs the GPU"
    np_ = lambda t: t.detach().cpu().numpy()
    n_inst = n_rec = n_tie = n_poly = n_refused = n_ex = 0
    fails = []
    all_runs = RUNS + ANN_RUNS
    per_run = {repr(r): 0 for r in all_runs}
    n_ill = {repr(r): 0 for r in all_runs}
    worst = 0.0
    t0 = time.time()
    for s in seeds:
        c = make_case(s)
        rec, st_all, nv, kap = ref[s]
        n_inst += c["B"]
        n_poly += c["segs"] is not None
        nm = c["masks"].reshape(c["B"], -1).sum(1)
        rles = None
        for r in all_runs:
            entry = r.get("entry")
            if entry == "poly" and c["segs"] is None:
                continue
            with scheduling(**{k: v for k, v in r.items() if k != "entry"}):
                try:
                    kw = dict(ground=c["ground"], sample_idx=c["sidx"], image_index=c["image_index"])
                    if entry == "rle":
                        if rles is None:
                            rles = [O.rle_encode(m) for m in c["masks"]]
                        b, stg, aux = fit_instances_rle(c["depth"], rles, c["K"], **kw)
                    elif entry == "poly":
                        b, stg, aux = fit_instances_poly(c["depth"], pack_polygons(c["segs"], c["H"], c["W"]), c["K"], **kw)
                    elif entry in ("ex_u8", "ex_rle"):
                        ers = np.random.RandomState(s)
                        hint = np.where(ers.rand(c["B"]) < 0.7, nm, ers.randint(0, 2 * c["H"] * c["W"], c["B"])).astype(np.int32)
                        size = (c["W"] + int(ers.randint(0, 50)), c["H"] + int(ers.randint(0, 50)))
                        flt = None
                        if entry == "ex_rle":
                            if rles is None:
                                rles = [O.rle_encode(m) for m in c["masks"]]
                            flt = dict(boundary_threshold=int(ers.choice([10, 1, 3])), scale_threshold=int(ers.choice([100, 1, 400])))
                            res = fit_instances_ex(c["depth"], c["K"], rles=rles, filter=flt, image_size=size, area_hint=hint, **kw)
                        else:
                            res = fit_instances_ex(c["depth"], c["K"], masks=c["mb"], image_size=size, area_hint=hint, **kw)
                        b, stg, aux = res["boxes"], res["status"], res["aux"]
                        ex = (np_(res["boxes2d"]), None if flt is None else np_(res["stats"]), flt, size)
                    else:
                        b, stg, aux = la.fit_instances(c["depth"], c["mb"], c["K"], **kw)
                    b, stg, aux = np_(b), np_(stg), np_(aux)
                except Exception as e:   # noqa: BLE001 - a campaign records every failure and goes on
                    # the documented limit (include/la3d.h): run-length / polygon masks and the reference-subsample mode keep the frame's
                    # bit image in LDS - frames above 1024 x 1024 pixels are refused loudly (the u8 entry takes them in full-mask mode)
                    if c["H"] * c["W"] > 1 << 20 and ("bit image in LDS" in str(e)) and (entry is not None or c["sidx"] is not None):
                        n_refused += 1
                        continue
                    fails.append((s, r, f"call failed: {e!r}"))
                    continue
            tag = f"seed {s} {c['H']}x{c['W']} B={c['B']} P={c['P']} skew={c['skew']} ground={'no' if c['ground'] is None else 'yes'} sample={c['sidx'] is not None} {r}"
            st = st_all
            if entry in ("ex_u8", "ex_rle"):
                b2d, stats, flt, size = ex
                if flt is not None:   # the fused filter: statistics and decisions against the oracle's, dropped instances carry status 6
                    ref_stats = np.array([O.mask_stats(m, flt["boundary_threshold"]) for m in c["masks"]])
                    keep = np.array([O.keep_instance(q, c["H"], True, flt["scale_threshold"]) for q in ref_stats])
                    if not np.array_equal(stats, ref_stats):
                        fails.append((s, r, f"fused filter: statistics differ at {np.flatnonzero((stats != ref_stats).any(1))[:4].tolist()}")); continue
                    st = np.where(keep, st_all, 6).astype(np.int32)
                    n_ex += int((~keep).sum())
                okb = (st == 0)
                Kp = c["K"] if c["image_index"] is None else c["K"][c["image_index"]]
                want2d = O.project_boxes(b, Kp if (c["P"] > 1 or c["image_index"] is not None) else c["K"][0], size)
                bad2d = np.flatnonzero(okb & ~(np.isclose(b2d, want2d, rtol=1e-12, atol=1e-9, equal_nan=True).all(1)))
                # (a corner on / behind the camera plane: the projection divides by ~0 - NaN rows are reported whole, include/la3d.h)
                bad2d = [i for i in bad2d if np.isfinite(want2d[i]).all() and np.isfinite(b2d[i]).all()]
                if len(bad2d):
                    fails.append((s, r, f"2-D boxes of the epilogue differ at {bad2d[:4]}: {b2d[bad2d[0]]} vs {want2d[bad2d[0]]}")); continue
                if not np.isnan(b2d[st != 0]).all():
                    fails.append((s, r, "2-D boxes of a rejected / filtered instance are not NaN")); continue
            if stg.tolist() != st.tolist():
                bad = np.flatnonzero(stg != st)
                fails.append((s, r, f"status at {bad[:5].tolist()}: got {stg[bad][:5].tolist()} expected {st[bad][:5].tolist()} (mask kinds {[c['mkind'][i] for i in bad[:5]]})"))
                continue
            ok = st == 0
            if not np.isnan(b[~ok]).all():
                fails.append((s, r, "a rejected instance's record is not NaN")); continue
            if not np.array_equal(aux[:, 2], nm):
                fails.append((s, r, "n_masked differs")); continue
            if not np.array_equal(aux[ok, 1], nv[ok]):
                fails.append((s, r, "n_valid differs")); continue
            tie = ok & ~(aux[:, 3] >= 1e-9)
            chk = ok & ~tie
            n_tie += int(tie.sum())
            noise = reference_axis_noise(kap, aux[:, 1], aux[:, 3])   # the reference's own rounding where it works from raw sums (n >= 20)
            n_ill[repr(r)] += int((chk & (kap > 131072.0)).sum())
            for n in np.flatnonzero(chk):
                try:
                    assert_records(b[n:n + 1], rec[n:n + 1], tag, gap=aux[n:n + 1, 3], noise=noise[n:n + 1])
                    n_rec += 1; per_run[repr(r)] += 1
                    if aux[n, 3] > 1e-4:
                        worst = max(worst, float(np.abs(b[n, :6] - rec[n, :6]).max() / max(np.abs(rec[n, :6]).max(), 1.0)))
                except AssertionError as e:
                    what = [ln for ln in str(e).splitlines() if "center" in ln or "R_cam" in ln or "vertices" in ln]
                    p_ = 0 if c["image_index"] is None and c["P"] == 1 else (int(c["image_index"][n]) if c["image_index"] is not None else n)
                    fails.append((s, r, f"record {n}: {what[0].strip() if what else 'mismatch'}; mask kind {c['mkind'][n]} depth kind {c['dkind'][p_]} n_valid {int(aux[n, 1])} "
                                        f"gap {aux[n, 3]:.3g} kappa {kap[n]:.3g} | d center/dims {np.abs(b[n, :6] - rec[n, :6]).max():.3g} (scale {np.abs(rec[n, :6]).max():.3g}) "
                                        f"dR {np.abs(b[n, 6:15] - rec[n, 6:15]).max():.3g} dV {np.nanmax(np.abs(b[n, 15:] - rec[n, 15:])):.3g} dims {rec[n, 3:6].round(6).tolist()}"))
    t_gpu = time.time() - t0
    lines = [f"fuzz_engines: {len(seeds)} cases (seeds {seeds[0]}..{seeds[-1]}), {n_inst} instances, {len(RUNS)} runs per case through the u8 entry + {len(ANN_RUNS)} through the annotation / extended entries (3 as run lengths, 3 as polygons for the {n_poly} polygon cases, 3 through la3d_fit_instances_ex with the 2-D boxes of the epilogue, an area hint and - run lengths - the fused filter: {n_ex} instances dropped by it)",
             f"oracle: {t_or:.0f} s on {a.workers} host cores; GPU runs + comparison: {t_gpu:.0f} s",
             f"records compared with the oracle: {n_rec} (+ {n_tie} exact ties held to status / counts only)",
             f"worst relative error of center / dims among records with an eigen-gap above 1e-4: {worst:.2e}",
             f"calls refused as documented (frames above 1 Mpx as run lengths / polygons / in subsample mode): {n_refused}",
             f"failures: {len(fails)}"]
    lines += [f"  compared under {k}: {v} (of them ill-conditioned for raw sums, kappa > 2^17, and resolved by the second moments pass: {n_ill[k]})" for k, v in per_run.items()]
    lines += [f"  FAIL seed {s} {r}: {m}" for s, r, m in fails[:400]]
    txt = "\n".join(lines)
    print(txt)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(txt + "\n")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

"""Round 6: randomised differential campaign of the path's other entries - everything either side of the box fit.

    python profiles/r06/fuzz_aux.py [--cases 300] [--seed 0] [--out profiles/r06/fuzz_aux.txt]

Per case (random frame sizes incl. widths that are no multiple of 32 / 4, heights no multiple of 8):
  unproject     la3d_unproject / la3d_unproject_batch (reference src/util.py:52-75): depth with NaN / inf / zero / negative pixels, K with
                and without skew, per-frame K, R / t given or not, f64 and f32 output                     -> 1e-12 of the scale (f64)
  run lengths   la3d_rle_decode, la3d_mask_stats_rle, la3d_mask_stats on the decoded planes: uncompressed lists and the compressed
                string form, masks of every kind of fuzz_engines.py                                      -> bit for bit / integer for integer
  polygons      la3d_poly_decode, la3d_mask_stats_poly against oracle/poly_oracle.py (the cv2.fillPoly restatement) -> bit for bit
  filters       keep_instances for both branches of the reference's rule (src/util.py:375)               -> the same decisions
  consumers     la3d_project_boxes (K shared / per box / indexed; corners behind the camera, on its plane) and la3d_iou_matrix
                (degenerate and disjoint boxes) against oracle project_boxes / iou2d_matrix               -> 1e-12
  depth stats   la3d_masked_ratio_median against np.median of the float32 ratios (ties, odd / even counts, 0/0, x/0), the depth-alignment
                selection (one frame and batched) and scatter against the reference's NumPy expressions (depth.py:67-90) -> bit for bit
  matcher       la3d_unproject_matches against the reference's expressions (src/matching/matcher.py:70-91)  -> 1e-12
The oracle is test infrastructure: it is the checker here.  Nothing under /root/reference is read."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def rand_rot(rs):
    q, _ = np.linalg.qr(rs.randn(3, 3))
    return q * np.sign(np.linalg.det(q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "fuzz_aux.txt"))
    a = ap.parse_args()

    import torch

    import fuzz_engines as FE
    import labelany3d_amd as la
    from labelany3d_amd import consumers as C
    from labelany3d_amd import masks as M
    from oracle import la3d_oracle as O
    from oracle import poly_oracle as P

    assert torch.cuda.is_available(), "the campaign needs the GPU"
    np_ = lambda t: t.detach().cpu().numpy()
    fails = []
    n = dict(unproject=0, rle=0, poly=0, stats=0, keep=0, project=0, iou=0, median=0, align=0, matches=0)
    t0 = time.time()
    for seed in range(a.seed, a.seed + a.cases):
        rs = np.random.RandomState(seed)
        H = int(rs.choice([1, 2, 7, 8, 16, 37, 64, 120, 240, 375, 480]))
        W = int(rs.choice([1, 3, 4, 31, 32, 33, 64, 100, 250, 333, 427, 640]))
        tag = f"seed {seed} {H}x{W}"
        try:
            # ---- unproject ----
            Pn = int(rs.choice([1, 1, 2, 5]))
            depth = np.stack([FE.one_plane(rs, H, W) for _ in range(Pn)])
            K = np.zeros((Pn, 3, 3))
            for p in range(Pn):
                f = rs.uniform(0.4, 3.0) * max(W, 8)
                K[p] = [[f, rs.uniform(-5, 5) * (rs.rand() < 0.3), W / 2 + rs.uniform(-0.3, 0.3) * W], [0, f * rs.uniform(0.8, 1.25), H / 2 + rs.uniform(-0.3, 0.3) * H], [0, 0, 1]]
            R = rand_rot(rs) if rs.rand() < 0.5 else None
            t = rs.randn(3) * 3 if rs.rand() < 0.5 else None
            for mode in range(3):
                if mode == 0:      # one frame
                    got = np_(la.unproject(depth[0], K[0], R, t))
                    ref = O.depth_to_points(depth[0][None], K[0], R, t)
                elif mode == 1:    # the batch, one K
                    got = np_(la.unproject(depth, K[0], R, t))
                    ref = np.stack([O.depth_to_points(d[None], K[0], R, t) for d in depth])
                else:              # the batch, a K per frame
                    got = np_(la.unproject(depth, K, R, t))
                    ref = np.stack([O.depth_to_points(d[None], k, R, t) for d, k in zip(depth, K)])
                ref = ref.reshape(got.shape)
                fin = np.isfinite(ref)
                if not np.array_equal(np.isnan(got), np.isnan(ref)) or not np.array_equal(np.isposinf(got), np.isposinf(ref)) or not np.array_equal(np.isneginf(got), np.isneginf(ref)):
                    fails.append((tag, f"unproject mode {mode}: NaN / inf pattern differs"))
                elif fin.any():
                    sc = max(1.0, float(np.abs(ref[fin]).max()))
                    err = float(np.abs(got[fin] - ref[fin]).max())
                    # (per point: |p| * a few ulp; the planes hold depths up to 1e3 next to 1e-2)
                    if err > 1e-12 * sc:
                        fails.append((tag, f"unproject mode {mode}: {err:.3g} off (scale {sc:.3g})"))
                n["unproject"] += 1
            g32 = np_(la.unproject(depth[0], K[0], R, t, out_dtype=torch.float32))
            r32 = O.depth_to_points(depth[0][None], K[0], R, t).reshape(g32.shape)
            fin = np.isfinite(r32) & (np.abs(r32) < 1e30)
            if fin.any() and float((np.abs(g32[fin] - r32[fin]) / np.maximum(np.abs(r32[fin]), 1e-30)).max()) > 2e-7:
                # (a float32 store of the float64 result: half an ulp = 6e-8 relative)
                bad = float((np.abs(g32[fin] - r32[fin]) / np.maximum(np.abs(r32[fin]), 1e-30)).max())
                fails.append((tag, f"unproject f32: relative {bad:.3g}"))

            # ---- run lengths / mask statistics / filters ----
            if H >= 8 and W >= 32:
                B = int(rs.choice([1, 3, 17, 40]))
                masks = np.stack([FE.one_mask(rs, H, W) for _ in range(B)])
                rles = [O.rle_encode(m) for m in masks]
                if rs.rand() < 0.5:   # the compressed string form of the annotation files
                    rles = [dict(size=r["size"], counts=(O.rle_to_string(r["counts"]) if rs.rand() < 0.7 else r["counts"])) for r in rles]
                bt = int(rs.choice([10, 10, 1, 3, 25]))
                dec = np_(M.rle_decode(rles))
                if not np.array_equal(dec, masks):
                    fails.append((tag, f"rle_decode differs at {int((dec != masks).sum())} pixels"))
                n["rle"] += B
                ref_stats = np.array([O.mask_stats(m, bt) for m in masks])
                for name, got in (("mask_stats", np_(M.mask_stats(masks.astype(np.uint8) * int(rs.choice([1, 255])), bt))),
                                  ("mask_stats_rle", np_(M.mask_stats_rle(rles, bt)))):
                    if not np.array_equal(got, ref_stats):
                        bad = np.flatnonzero((got != ref_stats).any(1))
                        fails.append((tag, f"{name} (boundary {bt}) differs at {bad[:4].tolist()}: got {got[bad[:2]].tolist()} expected {ref_stats[bad[:2]].tolist()}"))
                    n["stats"] += B
                for from_rle in (True, False):
                    st = int(rs.choice([100, 1, 1000]))
                    got = np_(M.keep_instances(torch.as_tensor(ref_stats, dtype=torch.int32), H, from_rle, st))
                    ref = np.array([O.keep_instance(s, H, from_rle, st) for s in ref_stats])
                    if not np.array_equal(got, ref):
                        fails.append((tag, f"keep_instances(from_rle={from_rle}) differs"))
                    n["keep"] += B

                # ---- polygons ----
                if H * W <= 120 * 333:
                    Bp = int(rs.choice([1, 4, 12]))
                    segs, pm = [], []
                    for _ in range(Bp):
                        m, seg = FE.one_polygon_mask(rs, H, W)
                        segs.append(seg); pm.append(m)
                    pm = np.stack(pm)
                    polys = M.pack_polygons(segs, H, W)
                    dec = np_(M.poly_decode(polys))
                    if not np.array_equal(dec, pm):
                        fails.append((tag, f"poly_decode differs at {int((dec != pm).sum())} pixels"))
                    got = np_(M.mask_stats_poly(polys, bt))
                    ref = np.array([O.mask_stats(m, bt) for m in pm])
                    if not np.array_equal(got, ref):
                        fails.append((tag, f"mask_stats_poly differs: got {got[:2].tolist()} expected {ref[:2].tolist()}"))
                    n["poly"] += Bp

            # ---- consumers ----
            Bb = int(rs.choice([1, 5, 64, 300]))
            rec = rs.randn(Bb, 39) * 3
            rec[:, 15:] = (rs.randn(Bb, 8, 3) * [2, 1, 2] + [0, 0, rs.uniform(-1, 12)]).reshape(Bb, 24)
            if rs.rand() < 0.3:
                rec[rs.randint(Bb), 17] = 0.0        # a corner on the camera plane: division by zero, as in the reference
            if rs.rand() < 0.2:
                rec[rs.randint(Bb)] = np.nan         # a rejected box
            size = (int(rs.choice([640, 500, 427])), int(rs.choice([480, 375, 640])))
            kk = rs.randint(0, 3)
            Kb = K[0] if kk == 0 else np.stack([K[0] * [[rs.uniform(0.5, 2)], [rs.uniform(0.5, 2)], [1]] for _ in range(Bb)])
            ii = None
            if kk == 2:
                ii = rs.randint(0, Bb, Bb).astype(np.int32)
            got = np_(C.project_boxes(rec, Kb, size, image_index=ii))
            ref = O.project_boxes(rec, Kb if ii is None else Kb[ii], size)
            # include/la3d.h: a box with a corner whose projection is NaN (a rejected box's NaN record, 0 / 0) gives 8 NaNs - Python's
            # min() / max() over a NaN depend on its position in the list, which the oracle restates; those rows are held to "all NaN"
            Kr = np.broadcast_to(Kb if ii is None else Kb[ii], (Bb, 3, 3)) if np.ndim(Kb) == 3 else np.broadcast_to(Kb, (Bb, 3, 3))
            with np.errstate(invalid="ignore", divide="ignore"):
                hp = np.einsum("bij,bvj->bvi", Kr, rec[:, 15:].reshape(Bb, 8, 3))
                nan_row = np.isnan(hp[..., :2] / hp[..., 2:3]).any((1, 2))
            if not np.isnan(got[nan_row]).all():
                fails.append((tag, "project_boxes: a box with a NaN projection is not reported as 8 NaNs"))
            got, ref = got[~nan_row], ref[~nan_row]
            same_nan = np.array_equal(np.isnan(got), np.isnan(ref))
            fin = np.isfinite(ref) & np.isfinite(got)
            if not same_nan:
                fails.append((tag, "project_boxes: NaN pattern differs"))
            elif not np.array_equal(np.isinf(got), np.isinf(ref)):
                fails.append((tag, "project_boxes: inf pattern differs"))
            elif fin.any() and float((np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1.0)).max()) > 1e-12:
                fails.append((tag, f"project_boxes: {float((np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1.0)).max()):.3g}"))
            n["project"] += Bb
            n0, n1 = int(rs.choice([1, 7, 60])), int(rs.choice([1, 9, 80]))
            b0 = np.sort(rs.uniform(0, 640, (n0, 2, 2)), 1).reshape(n0, 4)[:, [0, 2, 1, 3]]
            b1 = np.sort(rs.uniform(0, 640, (n1, 2, 2)), 1).reshape(n1, 4)[:, [0, 2, 1, 3]]
            if rs.rand() < 0.5:
                b1[0] = b0[0]                        # identical boxes
                b1[-1, 2:] = b1[-1, :2]              # an empty box
            got = np_(C.iou2d_matrix(b0, b1))
            ref = O.iou2d_matrix(b0, b1)
            if float(np.abs(got - ref).max()) > 1e-12:
                fails.append((tag, f"iou2d_matrix: {float(np.abs(got - ref).max()):.3g}"))
            n["iou"] += n0 * n1
            # ---- masked depth-ratio median (src/util.py:476-486), depth-alignment selection / scatter (depth.py:67-90), matcher ----
            if H >= 2 and W >= 4:
                Bm = int(rs.choice([1, 3, 9]))
                num = np.stack([FE.one_plane(rs, H, W) for _ in range(Bm)])
                den = np.stack([FE.one_plane(rs, H, W) for _ in range(Bm)])
                if rs.rand() < 0.4:   # heavy ties: a handful of distinct ratios
                    num = np.round(num).astype(np.float32); den = (np.round(np.abs(den)) + 1).astype(np.float32)
                ma = rs.rand(Bm, H, W) < 10 ** rs.uniform(-3, 0)
                mb = rs.rand(Bm, H, W) < rs.uniform(0.2, 1.0) if rs.rand() < 0.7 else None
                med, cnt = (np_(t) for t in la.masked_ratio_median(num, den, ma, mb))
                for i in range(Bm):
                    ov = ma[i] if mb is None else ma[i] & mb[i]
                    if cnt[i] != int(ov.sum()):
                        fails.append((tag, f"ratio median: count {cnt[i]} expected {int(ov.sum())}")); continue
                    with np.errstate(all="ignore"):
                        want = np.median(num[i][ov] / den[i][ov]) if ov.any() else np.float32(np.nan)
                    if not (np.float32(want) == med[i] or (np.isnan(want) and np.isnan(med[i]))):
                        fails.append((tag, f"ratio median: got {med[i]!r} expected {want!r} (count {cnt[i]})"))
                    n["median"] += 1
                from labelany3d_amd import depth_align as DA

                rel, met = num[0].copy(), np.abs(den[0]) * rs.uniform(1, 100)
                mk = ma[0] if rs.rand() < 0.5 else None
                cap = float(rs.choice([400.0, 50.0, 1e9]))
                with np.errstate(all="ignore"):
                    valid = (~np.isinf(rel)) & (met < cap) & (True if mk is None else mk)
                r_, m_ = DA.align_select(rel, met, mk, cap)
                if not (np.array_equal(np_(r_), rel[valid], equal_nan=True) and np.array_equal(np_(m_), met[valid], equal_nan=True)):
                    fails.append((tag, "align_select differs"))
                rb, mbt, cb = DA.align_select_batch(num, np.abs(den) * 3, ma if mk is not None else None, cap)
                for i in range(Bm):
                    with np.errstate(all="ignore"):
                        v = (~np.isinf(num[i])) & (np.abs(den[i]) * 3 < cap) & (True if mk is None else ma[i])
                    k = int(np_(cb)[i])
                    if k != int(v.sum()) or not np.array_equal(np_(rb)[i, :k], num[i][v], equal_nan=True) or not np.array_equal(np_(mbt)[i, :k], (np.abs(den[i]) * 3)[v].astype(np.float32), equal_nan=True):
                        fails.append((tag, f"align_select_batch differs at frame {i}"))
                coef, icpt = np.float32(rs.uniform(0.1, 50)), np.float32(rs.uniform(-1, 1) * (rs.rand() < 0.5))
                got = np_(DA.align_apply(rel, coef, icpt, mk))
                want = np.full_like(rel, 10000.0)
                sel = mk if mk is not None else ~np.isinf(rel)
                with np.errstate(all="ignore"):
                    want[sel] = rel[sel] * coef + icpt
                if not np.array_equal(got, want, equal_nan=True):
                    fails.append((tag, f"align_apply differs at {int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum())} pixels"))
                n["align"] += 2 + Bm
                # matcher unprojection (src/matching/matcher.py:70-91)
                dm = np.abs(num[0]) + 0.5
                dm[rs.rand(H, W) < 0.2] = -1
                N = int(rs.choice([1, 17, 300]))
                uv = np.stack([rs.uniform(0, W - 1e-3, N), rs.uniform(0, H - 1e-3, N)], 1)
                Rm, Tm = (rand_rot(rs), rs.randn(3)) if rs.rand() < 0.6 else (None, None)
                flip = float(rs.choice([512.0, 100.0])) if rs.rand() < 0.7 else None
                fx, fy, cx, cy = rs.uniform(100, 900), rs.uniform(100, 900), rs.uniform(0, W), rs.uniform(0, H)
                pts, vld = C.unproject_matches(dm, uv, fx, fy, cx, cy, flip, Rm, Tm)
                pts, vld = np_(pts), np_(vld)
                d_of = dm[uv[:, 1].astype(int), uv[:, 0].astype(int)]
                okm = d_of != -1
                u = (flip - uv[:, 0]) if flip is not None else uv[:, 0]
                v = (flip - uv[:, 1]) if flip is not None else uv[:, 1]
                p3 = np.stack(((u - cx) * d_of / fx, (v - cy) * d_of / fy, d_of), -1).astype(np.float64)
                if Rm is not None:
                    p3 = np.matmul(Rm, (p3.T - Tm.reshape(3, 1))).T
                if not np.array_equal(vld, okm) or not np.isnan(pts[~okm]).all():
                    fails.append((tag, "unproject_matches: validity differs"))
                elif okm.any() and float(np.abs(pts[okm] - p3[okm]).max()) > 1e-12 * max(1.0, float(np.abs(p3[okm]).max())):
                    fails.append((tag, f"unproject_matches: {float(np.abs(pts[okm] - p3[okm]).max()):.3g}"))
                n["matches"] += N
        except Exception as e:   # noqa: BLE001 - a campaign records every failure and goes on
            fails.append((tag, f"raised {e!r}"))
    lines = [f"fuzz_aux: {a.cases} cases (seeds {a.seed}..{a.seed + a.cases - 1}) in {time.time() - t0:.0f} s",
             f"checked: {n['unproject']} unproject calls, {n['rle']} run-length masks decoded, {n['stats']} mask statistics rows, {n['keep']} filter decisions, "
             f"{n['poly']} polygon annotations rasterised, {n['project']} boxes projected, {n['iou']} IoU entries, {n['median']} masked ratio medians, {n['align']} depth-alignment selections / scatters, {n['matches']} match points unprojected",
             f"failures: {len(fails)}"]
    lines += [f"  FAIL {t}: {m}" for t, m in fails[:300]]
    txt = "\n".join(lines)
    print(txt)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(txt + "\n")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

"""Round 6: randomised differential campaign of the per-image annotation entries - the reference's real calling pattern
(read_bounding_boxes_segmentations, src/util.py:336-383, followed by the box fit of every kept instance).

    python profiles/r06/fuzz_annotations.py [--cases 600] [--seed 0] [--workers 128] [--out profiles/r06/fuzz_annotations.txt]

One CASE = one image's annotation list (1 ... 40 annotations; frames up to 240 x 333 incl. odd widths): polygon annotations (1-3
parts, vertices outside the frame, degenerate parts), run-length annotations (uncompressed lists and the compressed string form),
crowd annotations and annotations without a segmentation (both skipped by the reference), `area` fields present for all / some /
none (the launch-order hint; deliberately wrong for some), random filter thresholds, a shared depth plane or several planes with a
per-annotation image index, ground planes for all / some / none.  Through
  fit_annotations(...)                  the GPU-resident form  (kept bboxes / indices / categories, boxes, status)
  fit_annotations(..., to_host=True)    the host-pointer entry la3d_fit_annotations_host (one C call per segmentation kind)
  fit_annotations_all(filter=None | True | thresholds)   one record per annotation, status 6 for skipped / filtered ones
and every decision (skip, keep rule of the right branch: rows holding a pixel for run lengths, first-to-last row for polygons) and
every record is compared with the CPU oracle, whose masks come from its own decoder / rasteriser.
The oracle is test infrastructure: it is the checker here.  Nothing under /root/reference is read."""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make_case(seed):
    import fuzz_engines as FE
    from oracle import la3d_oracle as O

    rs = np.random.RandomState(seed)
    H = int(rs.choice([24, 37, 64, 96, 120, 200, 240]))
    W = int(rs.choice([32, 64, 100, 128, 250, 320, 333]))
    n = int(rs.choice([1, 2, 5, 9, 20, 40]))
    P = int(rs.choice([1, 1, 1, 3]))
    depth = np.stack([FE.one_plane(rs, H, W) for _ in range(P)])
    K = np.zeros((P, 3, 3))
    for p in range(P):
        f = rs.uniform(0.4, 3.0) * W
        K[p] = [[f, rs.uniform(-5, 5) * (rs.rand() < 0.25), W / 2 + rs.uniform(-0.3, 0.3) * W], [0, f * rs.uniform(0.8, 1.25), H / 2 + rs.uniform(-0.3, 0.3) * H], [0, 0, 1]]
    image_index = rs.randint(0, P, n).astype(np.int32) if P > 1 else None
    area_mode = rs.randint(0, 3)             # 0: no area fields, 1: all (some wrong), 2: some
    anns, masks, kinds = [], [], []
    for i in range(n):
        r = rs.rand()
        a = dict(bbox=[float(v) for v in rs.uniform(0, 50, 4)], category_id=int(rs.randint(1, 134)))
        if r < 0.08:
            a["iscrowd"] = 1
            a["segmentation"] = O.rle_encode(FE.one_mask(rs, H, W))
            m, kind = None, "crowd"
        elif r < 0.14:
            m, kind = None, "none"           # no segmentation at all
        elif r < 0.55:
            m, seg = FE.one_polygon_mask(rs, H, W)
            a["segmentation"] = seg
            kind = "poly"
        else:
            m = FE.one_mask(rs, H, W)
            rle = O.rle_encode(m)
            if rs.rand() < 0.5:
                rle = dict(size=rle["size"], counts=O.rle_to_string(rle["counts"]))
            a["segmentation"] = rle
            kind = "rle"
        if rs.rand() < 0.7:
            a["iscrowd"] = a.get("iscrowd", 0)
        if m is not None and (area_mode == 1 or (area_mode == 2 and rs.rand() < 0.5)):
            a["area"] = float(m.sum()) if rs.rand() < 0.8 else float(rs.uniform(0, 2 * H * W))
        anns.append(a); masks.append(m); kinds.append(kind)
    gk = rs.randint(0, 4)
    ground = None
    if gk >= 1:
        ground = np.array([[0.05, -0.97, 0.1, 1.2]] * n) + 0.05 * rs.randn(n, 4)
        if gk == 3:
            for i in range(n):
                if rs.rand() < 0.25:
                    ground[i, 0] = np.nan
                elif rs.rand() < 0.1:
                    ground[i] = [0, -1, 0, 1.0]
    bt, st = int(rs.choice([10, 10, 1, 3])), int(rs.choice([100, 100, 1, 400]))
    return dict(seed=seed, H=H, W=W, n=n, P=P, depth=depth, K=K, image_index=image_index, anns=anns, masks=masks, kinds=kinds,
                ground=ground, bt=bt, st=st)


def oracle_case(seed):
    from oracle import la3d_oracle as O

    c = make_case(seed)
    n = c["n"]
    rec = np.full((n, 39), np.nan); st = np.full(n, 6, np.int32); keep = np.zeros(n, bool); keep_default = np.zeros(n, bool)
    nv = np.zeros(n, np.int64); kap = np.full(n, np.nan); gap = np.full(n, np.nan)
    for i in range(n):
        m = c["masks"][i]
        if m is None:
            continue
        p = 0 if c["image_index"] is None else int(c["image_index"][i])
        g = None if c["ground"] is None or np.isnan(c["ground"][i, 0]) else c["ground"][i]
        rec[i], st[i], aux = O.fit_instance(c["depth"][p], m, c["K"][p], g)
        nv[i], kap[i], gap[i] = aux.get("n_valid", 0), aux.get("kappa", np.nan), aux.get("gap", np.nan)
        keep[i] = O.keep_instance(O.mask_stats(m, c["bt"]), c["H"], c["kinds"][i] == "rle", c["st"])
        keep_default[i] = O.keep_instance(O.mask_stats(m, 10), c["H"], c["kinds"][i] == "rle", 100)
    return seed, (rec, st, keep, keep_default, nv, kap, gap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=600)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--workers", type=int, default=min(128, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "fuzz_annotations.txt"))
    a = ap.parse_args()
    seeds = list(range(a.seed, a.seed + a.cases))
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.workers) as pool:
        ref = dict(pool.imap_unordered(oracle_case, seeds, chunksize=1))
    t_or = time.time() - t0

    import torch

    from labelany3d_amd import masks as M
    from tests.test_gpu_parity import assert_records, reference_axis_noise

    assert torch.cuda.is_available(), "the campaign needs the GPU"
    np_ = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    fails = []
    n_ann = n_rec = n_calls = n_dropped = 0
    t0 = time.time()

    def compare(tag, b, s, idx, rec, st, kap, gap, nv):
        nonlocal n_rec
        b, s = np_(b), np_(s)
        if s.tolist() != st[idx].tolist():
            bad = np.flatnonzero(s != st[idx])
            fails.append((tag, f"status of annotations {np.asarray(idx)[bad][:5].tolist()}: got {s[bad][:5].tolist()} expected {st[idx][bad][:5].tolist()}"))
            return
        for j, i in enumerate(idx):
            if st[i] != 0:
                if not np.isnan(b[j]).all():
                    fails.append((tag, f"annotation {i}: status {st[i]} but the record is not NaN"))
                continue
            try:
                # (no aux in these entries: the oracle's own conditioning numbers - long double, centred - stand in.  A record is
                # compared unless its axis is unresolved: an exact / near-exact tie, or a footprint ill-conditioned for raw sums, which
                # the default engines resolve and the split engine reports as such)
                if not (kap[i] <= 2.0 ** 17) or not (gap[i] >= 1e-9):
                    continue
                assert_records(b[j:j + 1], rec[i:i + 1], tag, gap=gap[i:i + 1], noise=reference_axis_noise(kap[i:i + 1], nv[i:i + 1], gap[i:i + 1]))
                n_rec += 1
            except AssertionError as e:
                msg = [ln for ln in str(e).splitlines() if "center" in ln or "R_cam" in ln or "vertices" in ln]
                d = np.abs(b[j, :15] - rec[i, :15]).max()
                fails.append((tag, f"annotation {i}: {msg[0].strip() if msg else 'mismatch'} (max |d| {d:.3g}, kappa {kap[i]:.3g})"))

    for s in seeds:
        c = make_case(s)
        rec, st, keep, keep_default, nv, kap, gap = ref[s]
        n_ann += c["n"]
        tag0 = f"seed {s} {c['H']}x{c['W']} n={c['n']} P={c['P']} ground={'no' if c['ground'] is None else 'yes'}"
        depth_t = torch.as_tensor(c["depth"] if c["P"] > 1 else c["depth"][0], device="cuda")
        Kc = c["K"] if c["P"] > 1 else c["K"][0]
        want = np.flatnonzero(keep)
        n_dropped += int((~keep & np.array([m is not None for m in c["masks"]])).sum())
        for to_host in (False, True):
            tag = f"{tag0} fit_annotations(to_host={to_host})"
            try:
                bb, kept, cats, b, stt = M.fit_annotations(c["anns"], (c["W"], c["H"]), depth_t, Kc, ground=c["ground"], boundary_threshold=c["bt"],
                                                           scale_threshold=c["st"], image_index=c["image_index"], to_host=to_host)
                n_calls += 1
            except Exception as e:   # noqa: BLE001
                fails.append((tag, f"raised {e!r}")); continue
            if np.asarray(kept).tolist() != want.tolist():
                fails.append((tag, f"kept {np.asarray(kept).tolist()[:10]} expected {want.tolist()[:10]} (kinds {[c['kinds'][i] for i in want[:10]]})")); continue
            if bb != [c["anns"][i]["bbox"] for i in want] or list(cats) != [c["anns"][i]["category_id"] for i in want]:
                fails.append((tag, "bboxes / category ids of the kept annotations differ")); continue
            compare(tag, b, stt, want, rec, st, kap, gap, nv)
        for flt in (None, True, dict(boundary_threshold=c["bt"], scale_threshold=c["st"])):
            tag = f"{tag0} fit_annotations_all(filter={flt})"
            try:
                b, stt = M.fit_annotations_all(c["anns"], (c["W"], c["H"]), depth_t, Kc, ground=c["ground"], image_index=c["image_index"], filter=flt)
                n_calls += 1
            except Exception as e:   # noqa: BLE001
                fails.append((tag, f"raised {e!r}")); continue
            k = np.array([m is not None for m in c["masks"]]) if flt is None else (keep_default if flt is True else keep)
            exp_st = np.where(k, st, 6).astype(np.int32)
            exp_rec = np.where(k[:, None], rec, np.nan)
            compare(tag, b, stt, np.arange(c["n"]), exp_rec, exp_st, kap, gap, nv)
    t_gpu = time.time() - t0
    lines = [f"fuzz_annotations: {len(seeds)} images (seeds {seeds[0]}..{seeds[-1]}), {n_ann} annotations, {n_calls} calls (fit_annotations on the GPU and through the host entry, fit_annotations_all with three filters)",
             f"oracle: {t_or:.0f} s on {a.workers} host cores; GPU runs + comparison: {t_gpu:.0f} s",
             f"every skip / keep decision compared ({n_dropped} annotations dropped by the rule with the case's thresholds); records compared with the oracle: {n_rec}",
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

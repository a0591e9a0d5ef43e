"""Round 6: the differential record at BASELINE's scale (VERDICT r05 "next round" item 6).

    python profiles/r06/parity_config3_full.py [--images 5000] [--config5 16384] [--workers 128] [--out profiles/r06/parity_config3_full.txt]

Part 1 - BASELINE config 3 stand-in (COCO-val2017 size): IMAGES images with a SHARED 480x640 depth plane and their own K, ~Poisson(7)
instances per image (~36 k at 5000 images) whose masks are the annotation formats of the reference's data (24-vertex polygon
outlines, every third instance as uncompressed column-major run lengths - src/download_coconut.py:167-199), one ground plane per
instance (the reference's harness always passes one, src/util_3dbox.py:273-278).  EVERY instance goes through
  (a) the u8-plane entry   la3d_fit_instances          in reference-subsample mode (indices drawn in instance order from the seeded
                                                        global NumPy stream, src/util_3dbox.py:123-125) and in full-mask mode,
  (b) the annotation entry fit_annotations_all         (polygons rasterised / run lengths decoded inside the fit kernel), full-mask,
  (c) the annotation entry in reference-subsample mode (fit_instances_ex with polys / rles + sample_idx),
and EVERY record is compared with the CPU oracle (oracle/la3d_oracle.py - NumPy restatement of src/util.py:52-75 and
src/util_3dbox.py:106-178, pinned against reference-run fixtures) computed on this box's host cores from masks the oracle side
rasterises / decodes ITSELF (oracle/poly_oracle.py, la3d_oracle.rle_decode; a checksum ties them to the planes the GPU decoded).
Part 2 - a BASELINE config-5 batch (private depth planes, mask areas log-uniform 8..100k px) of --config5 instances, u8 entry,
full-mask mode, every record against the oracle.

Reported per entry: instances, status mismatches, worst relative error of center / dims (|got - ref| / max(|ref|, 1e-3)), worst
absolute yaw error over the instances whose eigen-gap exceeds 1e-6 (below it the axis is conditioned like 1 / gap), worst corner
error (fp16-quantised in the reference: one half-precision ulp allowed).  The oracle is test infrastructure: it is the checker here,
nothing on the product path touches it.  Nothing under /root/reference is read."""
import argparse
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
H, W = 480, 640


# ------------------------------------------------------------------ host-side oracle workers (no torch, no HIP) ---------------
def _checksum(mask):
    idx = np.flatnonzero(mask.ravel())
    return int(len(idx)), int((idx.astype(np.int64) * 2654435761 % 1000003).sum())


def oracle_image(task):
    """One image: its instances' masks rebuilt from the annotations by the oracle's own rasteriser / decoder, then the oracle fit in
    reference-subsample mode (given indices) and in full-mask mode."""
    from oracle import la3d_oracle as O
    from oracle import poly_oracle as P

    d = np.load(os.path.join(task["dir"], "depth.npy"), mmap_mode="r")
    depth = np.asarray(d[task["plane"]])
    K = task["K"]
    pts_all = O.depth_to_points(depth[None], K)
    out = []
    for seg, g, idx in zip(task["segs"], task["ground"], task["idx"]):
        if isinstance(seg, dict):
            m = O.rle_decode(np.asarray(seg["counts"], np.int64), H, W).astype(bool)
        else:
            m = P.create_boolean_mask_from_polygon((W, H), seg)[0]
        pts = pts_all[m]
        ri = np.asarray(idx) if (idx is not None and len(pts) > O.SUBSAMPLE) else False
        rs_, ss_, as_ = O.fit_points(pts, g, ri)
        rf_, sf_, af_ = O.fit_points(pts, g, False)
        out.append((rs_, ss_, as_["yaw"], rf_, sf_, af_["yaw"], _checksum(m)))
    return task["first"], out


def oracle_private(task):
    """config 5: one instance with its private depth plane, full-mask mode."""
    from oracle import la3d_oracle as O

    d = np.load(os.path.join(task["dir"], "depth.npy"), mmap_mode="r")
    m = np.load(os.path.join(task["dir"], "masks.npy"), mmap_mode="r")
    out = []
    for i in range(task["lo"], task["hi"]):
        rec, st, aux = O.fit_instance(np.asarray(d[i]), np.asarray(m[i]).astype(bool), task["K"])
        out.append((rec, st, aux["yaw"]))
    return task["lo"], out


# ------------------------------------------------------------------ comparison -----------------------------------------------
class Tally:
    def __init__(self, name):
        self.name, self.n, self.bad_status, self.nok = name, 0, 0, 0
        self.center = self.dims = self.yaw = self.rot = self.corner = 0.0
        self.gated = 0
        self.worst = None

    def add(self, got, gst, gaux, ref, rst, ryaw):
        self.n += len(gst)
        self.bad_status += int((gst != rst).sum())
        ok = (gst == 0) & (rst == 0)
        if not ok.any():
            return
        g, r, a = got[ok], ref[ok], gaux[ok]
        self.nok += len(g)
        rel = np.abs(g[:, :6] - r[:, :6]) / np.maximum(np.abs(r[:, :6]), 1e-3)
        c, d = rel[:, :3].max(), rel[:, 3:6].max()
        if max(c, d) > max(self.center, self.dims):
            i = int(np.argmax(rel.max(1)))
            self.worst = (float(rel[i].max()), g[i, :6].tolist(), r[i, :6].tolist(), float(a[i, 3]))
        self.center, self.dims = max(self.center, c), max(self.dims, d)
        wide = a[:, 3] > 1e-6
        self.gated += int((~wide).sum())
        if wide.any():
            dy = np.abs(a[wide, 0] - ryaw[ok][wide])
            self.yaw = max(self.yaw, float(dy.max()))
            self.rot = max(self.rot, float(np.abs(g[wide, 6:15] - r[wide, 6:15]).max()))
            self.corner = max(self.corner, float((np.abs(g[wide, 15:] - r[wide, 15:]) / np.maximum(np.abs(r[wide, 15:]), 1.0)).max()))

    def line(self):
        return (f"{self.name:58s} instances {self.n:6d}  status mismatches {self.bad_status}  fitted {self.nok:6d}  "
                f"worst rel. center {self.center:.2e}  dims {self.dims:.2e}  |yaw| {self.yaw:.2e}  R_cam {self.rot:.2e}  "
                f"corners (fp16) {self.corner:.2e}  [yaw / R / corners over eigen-gap > 1e-6: {self.nok - self.gated} of {self.nok}]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=5000)
    ap.add_argument("--chunk", type=int, default=250, help="images per GPU call / oracle round")
    ap.add_argument("--config5", type=int, default=16384)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "parity_config3_full.txt"))
    args = ap.parse_args()
    nw = args.workers or max(1, min(os.cpu_count() or 1, 128))
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ["PYTHONPATH"] = ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")
    pool = mp.get_context("spawn").Pool(nw)          # started before torch / HIP come up in this process
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="la3d_parity_", dir=shm)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    try:
        import torch

        import labelany3d_amd as la
        from oracle import la3d_oracle as O

        dev = torch.device("cuda", 0)
        np_ = lambda t: t.detach().cpu().numpy()  # noqa: E731
        say(f"# parity at BASELINE scale: {args.images} images (config-3 stand-in) + {args.config5} instances (config 5); oracle on {nw} host "
            f"processes; library {la._lib.lib.la3d_build_info().decode()[:48]}..")
        t_all = time.time()
        # ---------------- part 1: config 3 ----------------
        rs = np.random.RandomState(2017)
        np.random.seed(1)                              # the reference's global stream, consumed in instance order across the chunks
        vv, uu = np.mgrid[0:H, 0:W]
        ang = np.linspace(0, 2 * np.pi, 24, endpoint=False)
        ca, sa = np.cos(ang), np.sin(ang)
        tallies = {k: Tally(k) for k in ("config 3 | u8 planes        | reference-subsample", "config 3 | u8 planes        | full mask",
                                         "config 3 | annotations (all) | full mask", "config 3 | polygons / run len.| reference-subsample")}
        mask_mismatch = n_inst = n_poly = n_rle = n_sub = 0
        gpu_s = cpu_s = 0.0
        for c0 in range(0, args.images, args.chunk):
            P = min(args.chunk, args.images - c0)
            depth = np.empty((P, H, W), np.float32)
            Ks = np.empty((P, 3, 3))
            for p in range(P):
                depth[p] = (rs.uniform(2, 6) + rs.uniform(-0.004, 0.004) * uu + rs.uniform(0, 0.01) * vv + 0.03 * rs.randn(H, W)).astype(np.float32)
                f = rs.uniform(450, 700)
                Ks[p] = [[f, 0, 320 + rs.uniform(-8, 8)], [0, f * rs.uniform(0.98, 1.02), 240 + rs.uniform(-8, 8)], [0, 0, 1]]
            per = np.maximum(1, rs.poisson(7, P))
            img = np.repeat(np.arange(P), per).astype(np.int32)
            B = len(img)
            area = np.exp(rs.uniform(np.log(400), np.log(100000), B))
            asp = np.exp(rs.uniform(-0.7, 0.7, B))
            hh = np.clip(np.sqrt(area * asp), 8, H); ww = np.clip(area / hh, 8, W)
            r0 = rs.rand(B) * (H - hh); cc0 = rs.rand(B) * (W - ww)
            segs = [[np.stack([cc0[n] + ww[n] / 2 + ww[n] / 2 * ca, r0[n] + hh[n] / 2 + hh[n] / 2 * sa], 1).reshape(-1).tolist()] for n in range(B)]
            ground = np.array([[0.0, -1.0, 0.0, 1.5]] * B) + 0.05 * rs.randn(B, 4)
            t0 = time.time()
            d_t, k_t = torch.as_tensor(depth, device=dev), torch.as_tensor(Ks, device=dev)
            masks_t = la.poly_decode(la.pack_polygons(segs, H, W), device=dev)            # (B,H,W) bool on the GPU
            # every third instance travels as run lengths (encoded from the rasterised outline, as the converter does for crowd masks)
            for n in range(0, B, 3):
                segs[n] = O.rle_encode(np_(masks_t[n]))     # {"counts": [...], "size": [H, W]}: the converter's uncompressed form
            counts = np_(la.mask_counts(masks_t))
            idx = la.draw_sample_idx(counts)              # np.random, instance order: src/util_3dbox.py:123-125
            n_sub += int((counts > 500).sum())
            anns = [{"id": n, "image_id": int(img[n]), "category_id": 1, "iscrowd": 0, "bbox": [0, 0, 1, 1], "segmentation": segs[n]} for n in range(B)]
            res = {}
            b, s, a = la.fit_instances(d_t, masks_t, k_t, ground=ground, sample_idx=idx, image_index=img)
            res["config 3 | u8 planes        | reference-subsample"] = (np_(b), np_(s), np_(a))
            b, s, a = la.fit_instances(d_t, masks_t, k_t, ground=ground, image_index=img)
            res["config 3 | u8 planes        | full mask"] = (np_(b), np_(s), np_(a))
            b, s = la.fit_annotations_all(anns, (W, H), d_t, Ks, ground=ground, image_index=img)
            res["config 3 | annotations (all) | full mask"] = (np_(b), np_(s), None)
            # annotation formats in reference-subsample mode: one call per kind
            bb = np.full((B, 39), np.nan); ss = np.full(B, -1, np.int32); aa = np.full((B, 4), np.nan)
            pi = [n for n in range(B) if not isinstance(segs[n], dict)]
            ri = [n for n in range(B) if isinstance(segs[n], dict)]
            n_poly += len(pi); n_rle += len(ri)
            for sel, kw in ((pi, "polys"), (ri, "rles")):
                if not sel:
                    continue
                sel = np.asarray(sel)
                arg = la.pack_polygons([segs[n] for n in sel], H, W) if kw == "polys" else [segs[n] for n in sel]
                r_ = la.fit_instances_ex(d_t, k_t, ground=ground[sel], sample_idx=idx[sel], image_index=img[sel], device=dev, **{kw: arg})
                bb[sel], ss[sel], aa[sel] = np_(r_["boxes"]), np_(r_["status"]), np_(r_["aux"])
            res["config 3 | polygons / run len.| reference-subsample"] = (bb, ss, aa)
            flat_idx = masks_t.reshape(B, -1)
            wts = (torch.arange(H * W, device=dev, dtype=torch.int64) * 2654435761 % 1000003)
            cks = np_((flat_idx.to(torch.int64) * wts).sum(1))
            torch.cuda.synchronize()
            gpu_s += time.time() - t0
            # oracle round
            t0 = time.time()
            cdir = os.path.join(tmp, f"c{c0}")
            os.makedirs(cdir)
            np.save(os.path.join(cdir, "depth.npy"), depth)
            first = np.concatenate([[0], np.cumsum(per)])
            tasks = [{"dir": cdir, "plane": p, "K": Ks[p], "first": int(first[p]), "segs": segs[first[p]:first[p + 1]],
                      "ground": ground[first[p]:first[p + 1]], "idx": idx[first[p]:first[p + 1]]} for p in range(P)]
            ref_s = np.full((B, 39), np.nan); st_s = np.zeros(B, np.int32); yaw_s = np.full(B, np.nan)
            ref_f = np.full((B, 39), np.nan); st_f = np.zeros(B, np.int32); yaw_fr = np.full(B, np.nan)
            for fst, out in pool.imap_unordered(oracle_image, tasks, chunksize=1):
                for k, (r1, s1, y1, r2, s2, y2, ck) in enumerate(out):
                    n = fst + k
                    ref_s[n], st_s[n], yaw_s[n], ref_f[n], st_f[n], yaw_fr[n] = r1, s1, y1, r2, s2, y2
                    if ck != (int(counts[n]), int(cks[n])):
                        mask_mismatch += 1
            shutil.rmtree(cdir, ignore_errors=True)
            cpu_s += time.time() - t0
            for name, (gb, gs, ga) in res.items():
                sub = "reference-subsample" in name
                rr, rst, ry = (ref_s, st_s, yaw_s) if sub else (ref_f, st_f, yaw_fr)
                if ga is None:   # the annotation entry returns no aux: gap and yaw from the u8 full-mask call of the same masks
                    ga = res["config 3 | u8 planes        | full mask"][2]
                tallies[name].add(gb, gs, ga, rr, rst, ry)
            n_inst += B
            print(f"  .. images {c0 + P}/{args.images}, instances {n_inst}, gpu {gpu_s:.1f}s oracle {cpu_s:.1f}s", flush=True)
        say(f"config 3 stand-in: {args.images} images, {n_inst} instances ({n_poly} as 24-vertex polygons, {n_rle} as run lengths; {n_sub} masks above 500 px "
            f"draw 500 indices), shared depth + K per image, one ground plane per instance; masks rebuilt by the oracle side differing from the GPU's planes: {mask_mismatch}")
        for t in tallies.values():
            say(t.line())
            if t.worst:
                say(f"      worst record: rel {t.worst[0]:.2e} gap {t.worst[3]:.2e} got {np.round(t.worst[1], 9).tolist()} ref {np.round(t.worst[2], 9).tolist()}")
        # ---------------- part 2: config 5 ----------------
        if args.config5 > 0:
            import bench
            t5 = Tally("config 5 | u8 planes        | full mask")
            CH = 1024
            K5 = np.array(bench.K640)
            done = 0
            for c0 in range(0, args.config5, CH):
                Bc = min(CH, args.config5 - c0)
                d_t, m_t, k_t, _, _ = bench.make_config5(Bc, dev, 4321 + c0)
                b, s, a = la.fit_instances(d_t, m_t, k_t)
                cdir = os.path.join(tmp, f"p{c0}")
                os.makedirs(cdir)
                np.save(os.path.join(cdir, "depth.npy"), np_(d_t))
                np.save(os.path.join(cdir, "masks.npy"), np_(m_t))
                step = max(1, Bc // (4 * nw))
                tasks = [{"dir": cdir, "lo": lo, "hi": min(lo + step, Bc), "K": K5} for lo in range(0, Bc, step)]
                ref = np.full((Bc, 39), np.nan); rst = np.zeros(Bc, np.int32); ry = np.full(Bc, np.nan)
                for lo, out in pool.imap_unordered(oracle_private, tasks, chunksize=1):
                    for k, (r1, s1, y1) in enumerate(out):
                        ref[lo + k], rst[lo + k], ry[lo + k] = r1, s1, y1
                shutil.rmtree(cdir, ignore_errors=True)
                t5.add(np_(b), np_(s), np_(a), ref, rst, ry)
                done += Bc
                print(f"  .. config 5 instances {done}/{args.config5}", flush=True)
            say(f"config 5 workload: {args.config5} instances, private 480x640 depth planes ~U(0.5,10), mask areas log-uniform 8..100k px, ground=None, full-mask mode")
            say(t5.line())
        say(f"# wall {time.time() - t_all:.0f} s")
    finally:
        pool.terminate()
        shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

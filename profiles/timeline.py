#!/usr/bin/env python3
"""Per-workgroup phase timeline of the instance engine (measurement build: -DLA3D_TIMELINE, see profiles/timeline.sh).
Every workgroup stamps wall_clock64() (100 MHz) at its phase boundaries; this prints when the phases start and end
relative to the first workgroup's start.    python profiles/timeline.py [B ...]"""
import os
import sys

import numpy as np
import torch

os.environ["LA3D_ENGINE"] = "instance"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from labelany3d_amd import InstanceFitter  # noqa: E402

H, W = 480, 640
dev = torch.device("cuda", 0)
K = torch.tensor([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], dtype=torch.float64, device=dev)
names = ["start", "stream done", "list done", "pass A (wave 0)", "axis done", "pass B (wave 0)", "end"]


def one(B, sizes="bench"):
    rs = np.random.RandomState(1234)
    depth = torch.rand((B, H, W), device=dev) * 9.5 + 0.5
    masks = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    counts, offs = [], [0]
    for i in range(B):
        h, w = (rs.randint(8, 301), rs.randint(8, 331)) if sizes == "bench" else (154, 169)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = 1
        counts += [int(c0 * H + r0)] + [int(h), int(H - h)] * (int(w) - 1) + [int(h), int((W - c0 - w) * H + (H - r0 - h))]
        offs.append(len(counts))
    f = InstanceFitter(B, H, W, dev)
    if os.environ.get("TL_RLE"):  # same masks as COCO run lengths (la3d_fit_instances_rle)
        import ctypes as C

        from labelany3d_amd._lib import check, lib
        rc = torch.as_tensor(np.asarray(counts, np.int32), device=dev)
        ro = torch.as_tensor(np.asarray(offs, np.int64), device=dev)
        kf = K[None].expand(B, 3, 3).contiguous()
        st = torch.cuda.current_stream()

        def run_rle(*_):
            check(lib.la3d_fit_instances_rle(C.c_void_p(depth.data_ptr()), H * W, None, C.c_void_p(rc.data_ptr()),
                                             C.c_void_p(ro.data_ptr()), C.c_void_p(kf.data_ptr()), 9, None, None, B, H, W,
                                             C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()),
                                             C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()),
                                             C.c_void_p(st.cuda_stream)), "la3d_fit_instances_rle")
        f.run = run_rle
        sizes += " (run-length input)"
    if os.environ.get("TL_POLY"):  # same rectangles as 4-vertex polygon rings (la3d_fit_instances_poly)
        import ctypes as C

        from labelany3d_amd import pack_polygons
        from labelany3d_amd._lib import check, lib
        rs2 = np.random.RandomState(1234)
        segs = []
        for i in range(B):
            h, w = (rs2.randint(8, 301), rs2.randint(8, 331)) if sizes == "bench" else (154, 169)
            r0, c0 = rs2.randint(0, H - h + 1), rs2.randint(0, W - w + 1)
            segs.append([[int(c0), int(r0), int(c0 + w - 1), int(r0), int(c0 + w - 1), int(r0 + h - 1), int(c0), int(r0 + h - 1)]])
        pxy, pro, pir, _, _ = pack_polygons(segs, H, W)
        pxy, pro, pir = (torch.as_tensor(x, device=dev) for x in (pxy, pro, pir))
        kf = K[None].expand(B, 3, 3).contiguous()
        st = torch.cuda.current_stream()

        def run_poly(*_):
            check(lib.la3d_fit_instances_poly(C.c_void_p(depth.data_ptr()), H * W, None, C.c_void_p(pxy.data_ptr()), C.c_void_p(pro.data_ptr()),
                                              C.c_void_p(pir.data_ptr()), C.c_void_p(kf.data_ptr()), 9, None, None, B, H, W,
                                              C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()),
                                              C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()),
                                              C.c_void_p(st.cuda_stream)), "la3d_fit_instances_poly")
        f.run = run_poly
        sizes += " (polygon input)"
    if os.environ.get("TL_SAMPLE"):  # reference-subsample mode (500 drawn points per mask above 500 px)
        from labelany3d_amd import draw_sample_idx
        si = torch.as_tensor(draw_sample_idx(masks.reshape(B, -1).sum(1, dtype=torch.int64), np.random.RandomState(5)), device=dev)
        _run = f.run
        f.run = lambda d, m, k: _run(d, m, k, sample_idx=si)
        sizes += " (subsample mode)"
    if os.environ.get("TL_GROUND"):  # one ground plane per instance (the two-pass form; u8 planes only)
        gr = torch.as_tensor(np.array([[0.02, -0.97, 0.1, 1.2]] * B) + 0.03 * np.random.RandomState(77).randn(B, 4), device=dev)
        _run2 = f.run
        f.run = lambda d, m, k: _run2(d, m, k, ground=gr)
        sizes += " (grounded)"
    for _ in range(5):
        f.run(depth, masks, K)
    torch.cuda.synchronize()
    tl = f.workspace[0][8192: 8192 + B * 128].view(torch.float64).cpu().numpy().reshape(B, 16)
    t0 = tl[:, 8].min()                        # the first workgroup's entry (before any memory access)
    t = (tl[:, :7] - t0) / 100.0               # us
    ent = (tl[:, 8] - t0) / 100.0
    print(f"\n== B={B}: kernel entry (before the first load) min / mean / max: {ent.min():.1f} {ent.mean():.1f} {ent.max():.1f} us; "
          f"entry -> first stamp (perm / kernarg loads): mean {np.mean(t[:, 0] - ent):.1f} max {np.max(t[:, 0] - ent):.1f} us")
    if os.environ.get("TL_DETAIL"):
        blk = tl[:, 7].astype(int)
        for g in range(0, B, 256):
            sel = (blk >= g) & (blk < g + 256)
            print(f"   blocks {g:5d}..{g + 255:5d}: entry mean {ent[sel].mean():6.1f} max {ent[sel].max():6.1f} | first stamp mean {t[sel, 0].mean():6.1f}")
    print(f"\n== B={B} sizes={sizes}: launch-relative times in us (min / mean / max over workgroups)")
    for k, n in enumerate(names):
        print(f"  {n:18s} {t[:, k].min():7.1f} {t[:, k].mean():7.1f} {t[:, k].max():7.1f}")
    sub = (tl[:, 9:13] - t0) / 100.0           # finer stamps inside the two reduction stages (thread 0)
    print("  reduction stages, mean us: pass A end (wave 0) -> all waves in %.2f -> axis computed %.2f -> released %.2f | "
          "pass B end (wave 0) -> all waves in %.2f -> box written %.2f -> end %.2f" %
          (np.mean(sub[:, 0] - t[:, 3]), np.mean(sub[:, 1] - sub[:, 0]), np.mean(t[:, 4] - sub[:, 1]),
           np.mean(sub[:, 2] - t[:, 5]), np.mean(sub[:, 3] - sub[:, 2]), np.mean(t[:, 6] - sub[:, 3])))
    ls = (tl[:, 13:16] - t0) / 100.0           # stamps inside the list stage (thread 0): geometry read, first barrier, before the last barrier
    late = np.argsort(t[:, 6])[-64:]
    for nm, sel in (("all", slice(None)), ("the 64 latest finishers", late)):
        print("  list stage (%s), mean us: stream done -> M read %.2f -> row words / ballots / barrier %.2f -> list + compaction written %.2f -> barrier + survivor list %.2f" %
              (nm, np.mean(ls[sel, 0] - t[sel, 1]), np.mean(ls[sel, 1] - ls[sel, 0]), np.mean(ls[sel, 2] - ls[sel, 1]), np.mean(t[sel, 2] - ls[sel, 2])))
    d = np.diff(t, axis=1)
    print("  durations: " + "  ".join(f"{n}={d[:, k].mean():.1f} (max {d[:, k].max():.1f})"
                                      for k, n in enumerate(["stream", "list", "passA", "axis", "passB", "box"])))


for B in [int(a) for a in sys.argv[1:]] or [16, 1024]:
    one(B)
    one(B, "mean")

if os.environ.get("TL_DETAIL"):
    B = 1024
    sizes = "detail"
    rs = np.random.RandomState(1234)
    depth = torch.rand((B, H, W), device=dev) * 9.5 + 0.5
    masks = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    for i in range(B):
        h, w = rs.randint(8, 301), rs.randint(8, 331)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = 1
    f = InstanceFitter(B, H, W, dev)
    if os.environ.get("TL_SAMPLE"):  # reference-subsample mode (500 drawn points per mask above 500 px)
        from labelany3d_amd import draw_sample_idx
        si = torch.as_tensor(draw_sample_idx(masks.reshape(B, -1).sum(1, dtype=torch.int64), np.random.RandomState(5)), device=dev)
        _run = f.run
        f.run = lambda d, m, k: _run(d, m, k, sample_idx=si)
        sizes += " (subsample mode)"
    for _ in range(5):
        f.run(depth, masks, K)
    torch.cuda.synchronize()
    tl = f.workspace[0][8192: 8192 + B * 128].view(torch.float64).cpu().numpy().reshape(B, 16)
    t = (tl[:, :7] - tl[:, 8].min()) / 100.0
    ta = (tl[:, 13] - tl[:, 8].min()) / 100.0     # stream turn acquired (0 when the turns are off)
    blk = tl[:, 7].astype(int)
    order = np.argsort(blk)
    t = t[order]
    ta = ta[order]
    print("\nstart time percentiles (us):", np.percentile(t[:, 0], [10, 25, 50, 75, 90, 95, 99, 100]).round(1))
    print("end   time percentiles (us):", np.percentile(t[:, 6], [10, 25, 50, 75, 90, 95, 99, 100]).round(1))
    for g in range(4):
        sl = slice(g * 256, (g + 1) * 256)
        print(f"blocks {g*256:4d}..{g*256+255:4d}: start mean {t[sl, 0].mean():5.1f} max {t[sl, 0].max():5.1f} | turn acquired mean {ta[sl].mean():5.1f} max {ta[sl].max():5.1f} | stream done mean {t[sl, 1].mean():5.1f} max {t[sl, 1].max():5.1f} | "
              f"passA done mean {t[sl, 3].mean():5.1f} | passB done mean {t[sl, 5].mean():5.1f} | end mean {t[sl, 6].mean():5.1f} max {t[sl, 6].max():5.1f}")
    for x in range(8):
        sl = np.arange(B)[np.arange(B) % 8 == x]
        print(f"XCD {x}: start mean {t[sl, 0].mean():5.1f} max {t[sl, 0].max():5.1f} | end mean {t[sl, 6].mean():5.1f} max {t[sl, 6].max():5.1f}")

    if os.environ.get("TL_BW"):
        # aggregate demand profile: every workgroup's mask bytes spread over its stream interval, its active-tile bytes over its
        # pass A interval, the not-retained share over its pass B interval -> TB/s per 5 us bucket
        tiles = torch.nn.functional.max_pool2d(masks.float().view(B, 1, H, W), (8, 32)).view(B, -1).sum(1).cpu().numpy()
        tiles = tiles[np.argsort(np.argsort(blk))] if False else tiles
        inst_of_row = np.arange(B)                      # stamp rows are indexed by instance
        tt = (tl[:, :7] - tl[:, 8].min()) / 100.0
        keep = float(os.environ.get("TL_KEEP_TILES", "160"))
        edges = np.arange(0, tt[:, 6].max() + 5, 5.0)
        prof = np.zeros((3, len(edges) - 1))
        def spread(row, a, b, nbytes):
            if b <= a:
                b = a + 0.1
            for k in range(len(edges) - 1):
                ov = max(0.0, min(b, edges[k + 1]) - max(a, edges[k]))
                prof[row, k] += nbytes * ov / (b - a)
        for i in range(B):
            spread(0, tt[i, 0], tt[i, 1], H * W)
            spread(1, tt[i, 2], tt[i, 3], tiles[i] * 1024.0)
            spread(2, tt[i, 4], tt[i, 5], max(0.0, tiles[i] - keep) * 1024.0)
        print("\nmodelled traffic per 5 us bucket, TB/s (mask stream | pass A tiles | pass B re-read | sum):")
        for k in range(len(edges) - 1):
            m, a, b2 = prof[:, k] / 5e-6 / 1e12
            print(f"  {edges[k]:5.0f}-{edges[k+1]:3.0f} us: {m:5.2f} {a:5.2f} {b2:5.2f} | {m + a + b2:5.2f}")
    late = np.argsort(-t[:, 6])[:8]
    print("latest finishers (block, start, stream, list, passA, axis, passB, end):")
    for b in late:
        print("  ", b, t[b].round(1))

#!/usr/bin/env python3
"""A/B harness for kernel experiments: times la3d_fit_instances for several builds of libla3d.so (LA3D_LIB) and
environment switches on the same inputs, one subprocess per variant, and prints a table plus a checksum of the
records (variants must agree bit for bit unless stated).  Run on an MI355X:

    python profiles/sweep_variants.py label=path/to/lib.so[,ENV=val,...] ...  [--batches 512,1024,2048,8192] [--config3 800]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(args):
    import numpy as np  # noqa: F401
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from labelany3d_amd import InstanceFitter

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {}
    for B in args.batches:
        if args.config5:
            depth, masks, K, n_masked, _ = bench.make_config5(B, dev, 1234)
        else:
            depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
        f = InstanceFitter(B, bench.H, bench.W, dev)
        for _ in range(10):
            f.run(depth, masks, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = max(20, min(300, int(200 * 1024 / B)))
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                f.run(depth, masks, K)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / steps * 1e3)
        boxes, status, aux = f.run(depth, masks, K)
        torch.cuda.synchronize()
        h = hashlib.sha1(boxes.cpu().numpy().tobytes() + status.cpu().numpy().tobytes()).hexdigest()[:12]
        out[f"B{B}"] = {"us": best, "Mboxes_s": B / best, "sha": h, "ok": int((status == 0).sum())}
        del depth, masks, f
        torch.cuda.empty_cache()
    if args.rle:
        import ctypes as C

        from labelany3d_amd._lib import check, lib
        B = 1024
        depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
        rc_np, ro_np = bench.rect_rle(rects)
        rle_c, rle_o = torch.as_tensor(rc_np, device=dev), torch.as_tensor(ro_np, device=dev)
        kfull = K[None].expand(B, 3, 3).contiguous()
        f = InstanceFitter(B, bench.H, bench.W, dev)
        st = torch.cuda.current_stream()

        def run():
            check(lib.la3d_fit_instances_rle(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(rle_c.data_ptr()),
                                             C.c_void_p(rle_o.data_ptr()), C.c_void_p(kfull.data_ptr()), 9, None, None, B, bench.H,
                                             bench.W, C.c_void_p(f.boxes[0].data_ptr()), C.c_void_p(f.status[0].data_ptr()),
                                             C.c_void_p(f.aux[0].data_ptr()), C.c_void_p(f.workspace[0].data_ptr()),
                                             C.c_void_p(st.cuda_stream)), "rle")
        for _ in range(10):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        h = hashlib.sha1(f.boxes[0].cpu().numpy().tobytes()).hexdigest()[:12]
        out["rle1024"] = {"us": best, "Mboxes_s": B / best, "sha": h}
        del depth, masks, f
        torch.cuda.empty_cache()
    if args.poly:
        import ctypes as C

        import numpy as np
        from labelany3d_amd import pack_polygons
        from labelany3d_amd._lib import check, lib
        B = 1024
        depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
        r0, c0, hh, ww = rects
        rs = np.random.RandomState(7)
        segsets = {"rect4": [[[int(b), int(a), int(b + w - 1), int(a), int(b + w - 1), int(a + h - 1), int(b), int(a + h - 1)]]
                             for a, b, h, w in zip(r0, c0, hh, ww)]}
        star = []
        for a, b, h, w in zip(r0, c0, hh, ww):
            ang = np.sort(rs.uniform(0, 2 * np.pi, 60))
            rad = rs.uniform(0.7, 1.0, 60)
            star.append([np.stack([b + w / 2 + w / 2 * rad * np.cos(ang), a + h / 2 + h / 2 * rad * np.sin(ang)], 1).ravel().tolist()])
        segsets["star60"] = star
        kfull = K[None].expand(B, 3, 3).contiguous()
        f = InstanceFitter(B, bench.H, bench.W, dev)
        st = torch.cuda.current_stream()
        for name, segs in segsets.items():
            xy, ro, ir, _, _ = pack_polygons(segs, bench.H, bench.W)
            xy, ro, ir = (torch.as_tensor(x, device=dev) for x in (xy, ro, ir))

            def run():
                check(lib.la3d_fit_instances_poly(C.c_void_p(depth.data_ptr()), bench.H * bench.W, None, C.c_void_p(xy.data_ptr()),
                                                  C.c_void_p(ro.data_ptr()), C.c_void_p(ir.data_ptr()), C.c_void_p(kfull.data_ptr()), 9,
                                                  None, None, B, bench.H, bench.W, C.c_void_p(f.boxes[0].data_ptr()),
                                                  C.c_void_p(f.status[0].data_ptr()), C.c_void_p(f.aux[0].data_ptr()),
                                                  C.c_void_p(f.workspace[0].data_ptr()), C.c_void_p(st.cuda_stream)), "poly")
            for _ in range(10):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                e0.record()
                for _ in range(200):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
            h = hashlib.sha1(f.boxes[0].cpu().numpy().tobytes()).hexdigest()[:12]
            out[f"poly_{name}"] = {"us": best, "Mboxes_s": B / best, "sha": h}
        del depth, masks, f
        torch.cuda.empty_cache()
    if args.config3:
        depth, masks, K, n_masked, img = bench.make_config3(args.config3, dev, 1234)
        B = masks.shape[0]
        f = InstanceFitter(B, bench.H, bench.W, dev)
        for _ in range(3):
            f.run(depth, masks, K, image_index=img)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                f.run(depth, masks, K, image_index=img)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        boxes, status, aux = f.run(depth, masks, K, image_index=img)
        torch.cuda.synchronize()
        h = hashlib.sha1(boxes.cpu().numpy().tobytes()).hexdigest()[:12]
        out[f"c3_{args.config3}"] = {"us": best, "Mboxes_s": B / best, "sha": h, "B": B}
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--batches", type=lambda s: [int(x) for x in s.split(",")], default=[512, 1024, 2048, 8192])
    ap.add_argument("--config3", type=int, default=0)
    ap.add_argument("--config5", action="store_true")
    ap.add_argument("--rle", action="store_true")
    ap.add_argument("--poly", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    rows = []
    for v in args.variants:
        label, spec = v.split("=", 1)
        parts = spec.split(",")
        env = dict(os.environ)
        env.setdefault("LA3D_ENGINE", "instance")
        if parts[0]:
            env["LA3D_LIB"] = os.path.join(ROOT, parts[0]) if not os.path.isabs(parts[0]) else parts[0]
        for kv in parts[1:]:
            k, val = kv.split("=", 1)
            env[k] = val
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--batches", ",".join(map(str, args.batches))]
        if args.config3:
            cmd += ["--config3", str(args.config3)]
        if args.config5:
            cmd += ["--config5"]
        if args.rle:
            cmd += ["--rle"]
        if args.poly:
            cmd += ["--poly"]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        except subprocess.TimeoutExpired:
            print(f"{label}: TIMEOUT", flush=True)
            continue
        res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if not res:
            print(f"{label}: FAILED\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}", flush=True)
            continue
        rows.append((label, json.loads(res[-1][7:])))
        keys = list(rows[-1][1])
        print(f"{label:28s} " + "  ".join(f"{k}: {rows[-1][1][k]['us']:8.1f} us {rows[-1][1][k]['Mboxes_s']:6.2f} M/s [{rows[-1][1][k]['sha']}]"
                                          for k in keys), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Throughput of the auxiliary kernels (not the headline metric; bench.py is).  Run on an MI355X:
    python profiles/bench_aux.py > gpurun_out/bench_aux.json
Each entry: HIP-event time per call on the launch stream, algorithmic bytes per call, GB/s."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import labelany3d_amd as la  # noqa: E402
from labelany3d_amd.util import depth_to_points  # noqa: E402,F401

H, W = 480, 640
K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    torch.cuda.set_device(0)
    out = {}
    rs = np.random.RandomState(0)
    # depth_to_points, one 640x480 frame, f64 output (reference dtype): 4 B in + 24 B out per pixel
    d = torch.rand((H, W), device="cuda") * 9.5 + 0.5
    t = timed(lambda: la.unproject(d, K))
    out["unproject_640x480_f64"] = dict(s=t, bytes=H * W * 28, GBps=H * W * 28 / t / 1e9, frames_per_s=1 / t)
    stack = torch.rand((256, H, W), device="cuda") * 9.5 + 0.5
    Kst = torch.as_tensor(np.repeat(np.asarray(K)[None], 256, 0), device="cuda")
    t = timed(lambda: la.unproject(stack, Kst), n=10)
    out["unproject_batch_256x640x480_f64"] = dict(s=t, bytes=256 * H * W * 28, GBps=256 * H * W * 28 / t / 1e9, frames_per_s=256 / t,
                                                  note="256 frames, one launch (la3d_unproject_batch), K per frame inverted in the kernel; includes the output allocation")
    del stack
    big = torch.rand((2160, 3840), device="cuda") * 9.5 + 0.5
    t = timed(lambda: la.unproject(big, K), n=20)
    out["unproject_3840x2160_f64"] = dict(s=t, bytes=big.numel() * 28, GBps=big.numel() * 28 / t / 1e9)
    # estimate_bbox on 500-point clouds (the reference's actual call), 4096 clouds per launch
    B = 4096
    pts = torch.as_tensor(rs.randn(B * 500, 3) * [1.0, 0.3, 0.5] + [0, 0, 5.0], device="cuda")
    off = torch.arange(0, B * 500 + 1, 500, device="cuda", dtype=torch.int64)
    for method in ("pca", "convex_hull"):
        t = timed(lambda: la.fit_points((pts, off), None, None, method, hull_512=True), n=20)    # (500-row clouds: LA3D_HINT_HULL_512)
        out[f"fit_points_500pt_{method}"] = dict(s=t, boxes_per_s=B / t, bytes=B * 500 * 24 * 2, GBps=B * 500 * 48 / t / 1e9)
        if method == "pca":
            t = timed(lambda: la.fit_points((pts, off), None, None, method, small_clouds=True), n=20)
            out["fit_points_500pt_pca_wave_per_cloud"] = dict(s=t, boxes_per_s=B / t, bytes=B * 500 * 24, GBps=B * 500 * 24 / t / 1e9,
                                                              note="LA3D_HINT_SMALL_CLOUDS: one wave per cloud (bytes: the points once; the second walk hits the cache)")
    # masks: rle decode, stats
    Bm = 1024
    hh, ww = rs.randint(8, 301, Bm), rs.randint(8, 331, Bm)
    r0 = (rs.rand(Bm) * (H - hh + 1)).astype(int)
    c0 = (rs.rand(Bm) * (W - ww + 1)).astype(int)
    counts, offs = [], [0]
    for a, b, h, w in zip(r0, c0, hh, ww):
        counts += [int(b * H + a)] + [int(h), int(H - h)] * (int(w) - 1) + [int(h), int((W - b - w) * H + (H - a - h))]
        offs.append(len(counts))
    packed = (np.asarray(counts, np.int32), np.asarray(offs, np.int64), H, W)
    masks = la.rle_decode(packed)
    t = timed(lambda: la.rle_decode(packed), n=20)
    out["rle_decode_1024x640x480"] = dict(s=t, bytes=Bm * H * W, GBps=Bm * H * W / t / 1e9, masks_per_s=Bm / t,
                                          note="includes the host->device copy of the run lengths")
    t = timed(lambda: la.mask_stats(masks), n=20)
    out["mask_stats_1024x640x480"] = dict(s=t, bytes=Bm * H * W, GBps=Bm * H * W / t / 1e9)
    dev_packed = (torch.as_tensor(packed[0], device="cuda"), torch.as_tensor(packed[1], device="cuda"), H, W)
    t = timed(lambda: la.mask_stats_rle(dev_packed), n=20)
    out["mask_stats_rle_1024x640x480"] = dict(s=t, masks_per_s=Bm / t, run_bytes=int(packed[0].nbytes),
                                              note="run lengths resident on the device; equivalent u8 planes: 315 MB")
    t = timed(lambda: la.mask_counts(masks), n=20)
    out["mask_counts_1024x640x480"] = dict(s=t, bytes=Bm * H * W, GBps=Bm * H * W / t / 1e9)
    # polygons: the same rectangles as 4-vertex parts + star-shaped parts with ~60 vertices of about the same area
    rect_segs = [[[int(b), int(a), int(b + w - 1), int(a), int(b + w - 1), int(a + h - 1), int(b), int(a + h - 1)]]
                 for a, b, h, w in zip(r0, c0, hh, ww)]
    star_segs = []
    for a, b, h, w in zip(r0, c0, hh, ww):
        ang = np.sort(rs.uniform(0, 2 * np.pi, 60))
        rad = rs.uniform(0.7, 1.0, 60)
        star_segs.append([np.stack([b + w / 2 + w / 2 * rad * np.cos(ang), a + h / 2 + h / 2 * rad * np.sin(ang)], 1).ravel().tolist()])
    depth = torch.rand((Bm, H, W), device="cuda") * 9.5 + 0.5
    for name, segs in (("rect4", rect_segs), ("star60", star_segs)):
        polys = la.pack_polygons(segs, H, W)
        dpolys = tuple(torch.as_tensor(x, device="cuda") for x in polys[:3]) + polys[3:]
        t = timed(lambda: la.poly_decode(dpolys), n=20)
        out[f"poly_decode_1024x640x480_{name}"] = dict(s=t, masks_per_s=Bm / t, GBps=Bm * H * W / t / 1e9, vertices=int(len(polys[0])))
        t = timed(lambda: la.mask_stats_poly(dpolys), n=20)
        out[f"mask_stats_poly_1024_{name}"] = dict(s=t, masks_per_s=Bm / t)
        t = timed(lambda: la.fit_instances_poly(depth, dpolys, K), n=20)
        out[f"fit_instances_poly_1024_{name}"] = dict(s=t, boxes_per_s=Bm / t,
                                                       note="polygon parts rasterised inside the fit kernel; includes output allocation")
        t = timed(lambda: la.fit_instances_poly(depth, dpolys, K, filter=True), n=20)
        kept = int((la.fit_instances_poly(depth, dpolys, K, filter=True)[1] != 6).sum())
        out[f"fit_instances_poly_filtered_1024_{name}"] = dict(
            s=t, annotations_per_s=Bm / t, kept=kept,
            note="the reference's instance filter (src/util.py:375) evaluated inside the fit launch; compare with "
                 "mask_stats_poly + fit_instances_poly as two launches over the polygons")
    # masked depth-ratio median (align_to_depth_match): two u8 masks + two f32 planes per instance
    den = torch.rand((Bm, H, W), device="cuda") * 2.8 + 0.2
    mb = torch.rand((Bm, H, W), device="cuda") < 0.8
    t = timed(lambda: la.masked_ratio_median(depth, den, masks, mb), n=20)
    ov = int((masks & mb).sum())
    out["masked_ratio_median_1024x640x480"] = dict(s=t, instances_per_s=Bm / t, mask_bytes=2 * Bm * H * W, overlap_px=ov,
                                                   GBps=(2 * Bm * H * W + 3 * 8 * ov) / t / 1e9,
                                                   note="bytes = both masks once + 8 B per overlap pixel in each of the three radix rounds")
    # align_depth selection (one 640x480 frame) and prediction scatter
    rel, met = depth[0].clone(), den[0] * 100
    rel[torch.rand((H, W), device="cuda") < 0.05] = float("inf")
    t = timed(lambda: la.align_select(rel, met, None, 200.0), n=50)
    out["align_select_640x480"] = dict(s=t, frames_per_s=1 / t, note="three small launches + one 8-byte read-back of the count")
    P = 64
    relb, metb = rel[None].expand(P, -1, -1).contiguous(), met[None].expand(P, -1, -1).contiguous()
    from labelany3d_amd.depth_align import align_select_batch
    t = timed(lambda: align_select_batch(relb, metb, None, 200.0), n=20)
    out["align_select_batch_64x640x480"] = dict(s=t, frames_per_s=P / t, note="three launches for 64 frames, counts stay on the device (incl. the wrapper's output allocation)")
    t = timed(lambda: la.align_apply(rel, 2.5, 0.0), n=50)
    out["align_apply_640x480"] = dict(s=t, frames_per_s=1 / t)
    # consumers
    boxes, _, _ = la.fit_instances(depth, masks, K)
    t = timed(lambda: la.project_boxes(boxes, K, (W, H)))
    out["project_boxes_1024"] = dict(s=t, boxes_per_s=Bm / t)
    t0 = timed(lambda: la.fit_instances_ex(depth, K, masks=masks), n=20)
    t1 = timed(lambda: la.fit_instances_ex(depth, K, masks=masks, image_size=(W, H)), n=20)
    out["fit_instances_ex_1024"] = dict(s=t0, s_with_boxes2d=t1,
                                        note="u8 planes through the Python wrapper; with image_size the records' 2-D boxes come from the same epilogue")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

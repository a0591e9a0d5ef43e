#!/usr/bin/env python3
"""DESIGN.md section 5 table from the round-6 measurement set (profiles/r06_run.sh -> profiles/r06_bench_*.json,
profiles/traffic_per_launch.json):     python profiles/r06/make_design_table.py [dir with bench_*.json]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06")
tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_launch.json")))["modes"]
order = ["config2", "config2_ground", "config2_twopass", "config2_area_hint", "config2_subsample", "config2_rle", "config2_rle_area_hint", "config2_poly", "config2_B256", "config2_B8192",
         "config5", "config5_B16384", "config3_5000", "config3_14750"]
names = {
    "config2": "**config 2** (headline): 1024 instances, private depth, u8 rectangles",
    "config2_ground": "config 2 masks and depth with one ground plane per instance (`--ground`: the two-pass form; the reference's harness always passes one)",
    "config2_twopass": "config 2 with the two-pass form pinned (`LA3D_SEP=0`: round 4's default)",
    "config2_retaining": "config 2 with the retaining build pinned (`LA3D_RETAIN=1`: 128 VGPRs, two workgroups per CU; the default of rounds 2–3)",
    "config2_area_hint": "config 2, mask areas handed to the fit (`--area-hint`): no helper launch",
    "config2_subsample": "config 2, reference-subsample mode (`--subsample`)",
    "config2_rle": "config 2 masks as COCO run lengths (`--rle`)",
    "config2_rle_area_hint": "`--rle --area-hint` (the annotation's `area`)",
    "config2_poly": "config 2 masks as polygon parts (`--poly`, 4-vertex rings)",
    "config2_B256": "config 2 at B = 256 (instance engine since round 5; round 4: band engine)",
    "config2_B8192": "config 2 at B = 8192 per call",
    "config5": "**config 5** workload (areas log-uniform 8..100k px), B = 1024",
    "config5_B16384": "config 5 workload, B = 16 384 in one call",
    "config3_5000": "**config 3** stand-in: 5000 shared depth planes, ~35 k instances, one call",
    "config3_14750": "**config 4**, one GPU's share at full size (14 750 planes, ~103 k instances, one call)",
}
print("| workload | µs / step, HIP events: rotating inputs (every step on one batch, as rounds 1–5 measured) | boxes/s (wall) | required MB / step → GB/s (frac of 8 TB/s; of the measured stream) | PMC traffic MB / step (÷ required) | L2 hit | dominant kernel under rocprof: avg µs → traffic ÷ time (frac of 8 TB/s) |")
print("|---|---|---|---|---|---|---|")
for k in order:
    f = os.path.join(d, f"bench_{k}.json")
    if not os.path.exists(f):
        continue
    lines = [ln for ln in open(f) if ln.startswith("{")]
    if not lines:
        continue
    b = json.loads(lines[-1])
    r = b["roofline"]
    t = tj.get(k)
    req = r["required_bytes_per_launch"]
    row = [names.get(k, k), f"{r['avg_launch_ms'] * 1e3:.1f}" + (f" ({b['rotation']['same_batch_ms_per_step'] * 1e3:.1f})" if b.get("rotation", {}).get("same_batch_ms_per_step") else ""), f"{b['value'] / 1e6:.2f} M",
           f"{req / 1e6:.1f} → {r['achieved']:.0f} ({r['frac']:.2f}; {r['frac_of_measured_stream']:.2f} of {r['measured_stream_GBps']:.0f})"]
    if t:
        tr = t["hbm_bytes_per_step"]
        kns = t["dominant_kernel_avg_ns"]
        row += [f"{tr / 1e6:.1f} ({tr / req:.2f})", f"{100 * t['l2_hit_rate']:.1f} %" if t.get("l2_hit_rate") is not None else "—",
                f"{kns / 1e3:.1f} → {tr / kns:.0f} GB/s ({tr / kns / 8000:.2f})" if kns else "—"]
    else:
        row += ["—", "—", "—"]
    print("| " + " | ".join(row) + " |")
for name in ("bench_driver_style_final.json", "bench_driver_style.json", "bench_default_final.json"):
    f = os.path.join(d, name)
    if os.path.exists(f):
        lines = [ln for ln in open(f) if ln.startswith("{")]
        if lines:
            b = json.loads(lines[-1])
            extra = ""
            if "pipelined" in b:
                extra = f"; pipelined on two streams {b['pipelined']['value'] / 1e6:.2f} M ({b['pipelined']['ms_per_step'] * 1e3:.1f} µs)"
            if "cpu_baseline" in b:
                c = b["cpu_baseline"]
                extra += f"; cpu_baseline {c['value']:.0f} boxes/s on {c['cores']} cores ({c['kind']}), single thread {c.get('single_thread_value', 0):.1f}"
            print(f"\n`{name}`: steps {b['steps']} / warm-up {b['warmup']}: **{b['value'] / 1e6:.2f} M boxes/s**, {b['ms_per_step'] * 1e3:.1f} µs per step (wall), "
                  f"{b['roofline']['avg_launch_ms'] * 1e3:.1f} µs (HIP events), frac {b['roofline']['frac']:.3f}, traffic_stale {b['roofline']['traffic_stale']}{extra}")

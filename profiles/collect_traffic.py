#!/usr/bin/env python3
"""HBM bytes per bench step from the rocprofv3 counter passes of one profiled command (profiles/run_profile.sh):
    python profiles/collect_traffic.py <prof_dir> <tag> <steps incl. warm-up>   ->  one JSON line on stdout
read  = sum over the library's kernels of TCC_EA0_RDREQ_sum x 128 B (cross-check: FETCH_SIZE KB x 1024 x 2, the guide's gfx950
correction, calibrated on the fit kernel's own pattern by profiles/calibrate_fetch.sh), write = WRITE_SIZE KB x 1024, both divided
by the number of steps of the pass.  Kernels of the input generation (torch) are not counted."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import short  # noqa: E402

OURS = ("fit_instances_kernel", "fit_bands_kernel", "fit_rows_kernel", "merge_rows_kernel", "size_estimate_kernel", "scan_kernel", "plan_kernel", "walk_kernel", "axis_kernel", "final_kernel", "geo_kernel",
        "fit_points", "project_boxes_kernel", "mask_counts_kernel")


def ours(name):
    return any(k in name for k in OURS)


def main(d, tag, steps):
    tot = defaultdict(float)
    per_kernel = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if not ours(r["Kernel_Name"]):
                continue
            if "mask_counts_kernel" in r["Kernel_Name"]:      # the stream-ceiling measurement before the warm-up: not a step
                continue
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            tot[c] += v
            k = short(r["Kernel_Name"])
            per_kernel[k][c] += v
            disp[k][c] += 1
    dom, dom_ns, stats = None, None, {}
    for f in glob.glob(os.path.join(d, "tcc", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ours(r["Name"]) and "mask_counts_kernel" not in r["Name"]:
                stats[short(r["Name"])] = (float(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["Calls"]))
    if stats:
        dom = max(stats, key=lambda k: stats[k][0])
        dom_ns = stats[dom][1]
    # the same kernel in the DEFAULT-LENGTH command (run_profile.sh pass 1, LIGHT=0 only: 1000 timed steps + its secondary loops):
    # the 55-launch counter passes start from an idle chip and carry the counter collection - their average reads 2-3 % longer
    long_ns, long_calls = None, None
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if dom is not None and short(r["Name"]) == dom:
                long_ns, long_calls = float(r["AverageNs"]), int(r["Calls"])
    rd = tot.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0 / steps
    fs = tot.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / steps
    wr = tot.get("WRITE_SIZE", 0.0) * 1024.0 / steps
    hit, miss = tot.get("TCC_HIT_sum", 0.0), tot.get("TCC_MISS_sum", 0.0)
    out = {
        "tag": tag, "steps_in_pass": steps,
        "hbm_bytes_per_step": int(rd + wr), "read_bytes_per_step": int(rd), "read_bytes_per_step_from_FETCH_SIZE_x2": int(fs),
        "write_bytes_per_step": int(wr),
        "l2_hit_rate": (hit / (hit + miss)) if hit + miss else None,
        "dominant_kernel": dom, "dominant_kernel_avg_ns": dom_ns,
        "dominant_kernel_avg_ns_default_length": long_ns, "dominant_kernel_calls_default_length": long_calls,
        "step_ns_sum_of_kernels": sum(v[0] for v in stats.values()) / steps if stats else None,
        "per_kernel_read_bytes_per_step": {k: int(v.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0 / steps) for k, v in per_kernel.items()},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))

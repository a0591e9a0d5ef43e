#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + counter collection) into a small markdown summary.
    python profiles/summarize.py <prof_dir> <out.md>
Counter rows are averaged per kernel name over all dispatches in the pass."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("fit_instances_kernel", "fit_bands_kernel", "fit_points_kernel", "prep_kernel", "unproject_kernel", "mask_counts_kernel",
              "size_estimate_kernel", "launch_order_kernel", "poly_decode_kernel", "mask_stats_poly_kernel", "ratio_median_kernel",
              "align_count_kernel", "align_scan_kernel", "align_scatter_kernel", "align_apply_kernel", "rle_decode_kernel", "mask_stats", "scan_kernel", "plan_kernel", "walk_kernel", "axis_kernel", "final_kernel", "geo_kernel"):
        if k in name:
            i = name.find(k)
            return name[i:].split("(")[0][:70]
    return name.split("(")[0][:60]


def main(d, out):
    lines = [f"# rocprofv3 summary of `{os.path.basename(d.rstrip('/'))}`", ""]
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        lines += [f"## kernel stats ({os.path.relpath(f, d)})", "", "| kernel | calls | total ns | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(f)):
            lines.append(f"| {short(r['Name'])} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
        lines += [f"## counters ({os.path.relpath(f, d)})", ""]
        for k, cs in acc.items():
            lines.append(f"**{k}** (VGPR {meta[k][0]}, SGPR {meta[k][1]}, LDS {meta[k][2]} B, scratch {meta[k][3]}, wg {meta[k][4]}, grid {meta[k][5]})")
            lines += ["", "| counter | dispatches | mean per dispatch | min | max |", "|---|---|---|---|---|"]
            for c, v in sorted(cs.items()):
                lines.append(f"| {c} | {len(v)} | {sum(v) / len(v):.6g} | {min(v):.6g} | {max(v):.6g} |")
            lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

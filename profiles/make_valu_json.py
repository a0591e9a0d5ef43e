#!/usr/bin/env python3
"""Shader-side counters of the dominant kernel per profiled bench mode -> profiles/valu_per_launch.json (read by bench.py for the
`roofline.valu` object).  Source: the rocprofv3 summaries written by profiles/run_profile.sh with LIGHT=0 (SQ_INSTS_VALU,
SQ_ACTIVE_INST_VALU in quad-cycles, and the kernel's average duration in the same counter pass).
    python profiles/make_valu_json.py"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SIMDS = 256 * 4
CLOCK_GHZ = 2.4
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
files = {"config2": f"{TAG}_config2_summary.md", "config2_rle": f"{TAG}_config2_rle_sq_summary.md", "config2_poly": f"{TAG}_config2_poly_sq_summary.md",
         "config2_B8192": f"{TAG}_config2_B8192_sq_summary.md"}
out = {}
for mode, fn in files.items():
    p = os.path.join(ROOT, "profiles", fn)
    if not os.path.exists(p):
        continue
    t = open(p).read()
    # the sq pass: its kernel table, then the fit kernel's counter block
    sec = t[t.index("## kernel stats (sq/sq_kernel_stats.csv)"):]
    m = re.search(r"\| (fit_instances_kernel<[^|]+>|walk_kernel<[^|]+>) \| (\d+) \| (\d+) \| (\d+) \|", sec)
    kname, calls, avg_ns = m.group(1), int(m.group(2)), int(m.group(4))
    cs = t[t.index("## counters (sq/sq_counter_collection.csv)"):]
    blk = cs[cs.index("**" + kname + "**"):]
    blk = blk[:blk.index("\n**", 4)] if "\n**" in blk[4:] else blk
    val = lambda name: float(re.search(r"\| " + name + r" \| \d+ \| ([0-9.e+]+) \|", blk).group(1))  # noqa: E731
    insts, act = val("SQ_INSTS_VALU"), val("SQ_ACTIVE_INST_VALU")
    busy_us = act * 4 / SIMDS / (CLOCK_GHZ * 1e3)
    out[mode] = {"kernel": kname, "wave_instructions_valu": insts, "active_inst_valu_quadcycles": act,
                 "valu_busy_us_per_simd": busy_us, "kernel_avg_us_in_the_counter_pass": avg_ns / 1e3,
                 "valu_utilisation": busy_us / (avg_ns / 1e3), "summary": "profiles/" + fn}
doc = {"kernel_source_sha256": bench.kernel_source_sha256(), "modes": out,
       "note": "valu_busy_us_per_simd = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz: the time one SIMD spends issuing VALU "
               "instructions of this kernel per launch - the second roof of the path next to HBM bytes (DESIGN.md section 5.1)"}
json.dump(doc, open(os.path.join(ROOT, "profiles", "valu_per_launch.json"), "w"), indent=1)
print(json.dumps({k: (round(v["valu_busy_us_per_simd"], 1), round(v["kernel_avg_us_in_the_counter_pass"], 1), round(v["valu_utilisation"], 2)) for k, v in out.items()}))

#!/usr/bin/env python3
"""gpurun_out/traffic_modes.jsonl (one line per profiled bench mode, written by profiles/run_profile.sh -> collect_traffic.py)
-> gpurun_out/traffic_per_launch.json, stamped with the SHA-256 of the kernel sources it was measured on.
    python profiles/make_traffic_json.py r03          (run at the end of profiles/r03_run.sh, same tree)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(tag):
    modes = {}
    for line in open(os.path.join(ROOT, "gpurun_out", "traffic_modes.jsonl")):
        e = json.loads(line)
        key = e["tag"][len(tag) + 1:] if e["tag"].startswith(tag + "_") else e["tag"]
        e["source"] = (f"rocprofv3 --kernel-trace --stats --pmc {{FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum}} (separate passes, "
                       f"profiles/run_profile.sh {e['tag']}), all of the library's kernels of `bench.py --steps 50 --warmup 5` divided by its 55 "
                       f"steps; read = TCC_EA0_RDREQ_sum x 128 B (FETCH_SIZE x 1024 x 2 as the cross-check), write = WRITE_SIZE x 1024; summary "
                       f"profiles/{tag}_{key}_summary.md")
        modes[key] = e
    out = {"kernel_source_sha256": bench.kernel_source_sha256(), "modes": modes,
           "note": "key = bench.traffic_mode_key(args, B); hbm_bytes_per_step covers every kernel of a step (fit kernel + estimate kernel, or "
                   "the split engine's kernels), not only the dominant one"}
    dst = os.path.join(ROOT, "gpurun_out", "traffic_per_launch.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: (v["hbm_bytes_per_step"], v["dominant_kernel_avg_ns"], v["l2_hit_rate"]) for k, v in modes.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")

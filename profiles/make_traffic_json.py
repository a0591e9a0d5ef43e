#!/usr/bin/env python3
"""Turn the rocprofv3 summary of the default bench command (gpurun_out/profile_<tag>.md, written by profiles/run_profile.sh)
into profiles/traffic_per_launch.json, stamped with the SHA-256 of the kernel sources it was measured on.
    python profiles/make_traffic_json.py r02          (run right after profiles/run_profile.sh r02, same tree)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counter(text, kernel_prefix, name):
    for m in re.finditer(r"\*\*" + re.escape(kernel_prefix) + r"[^\n]*\n\n\| counter[^\n]*\n\|[-|]*\n((?:\|[^\n]*\n)+)", text):
        for line in m.group(1).splitlines():
            c = [x.strip() for x in line.strip("|").split("|")]
            if c[0] == name:
                return float(c[2]), int(c[1])
    return None, 0


def main(tag):
    text = open(os.path.join(ROOT, "gpurun_out", f"profile_{tag}.md")).read()
    kern = "fit_instances_kernel<true, true, false, true, 0, 4>"   # B = 1024 takes the retaining build (RET = 4)
    if kern not in text:
        kern = "fit_instances_kernel<true, true, false, true, 0, 0>"
    rd, n = counter(text, kern, "TCC_EA0_RDREQ_sum")
    fs, _ = counter(text, kern, "FETCH_SIZE")
    ws, _ = counter(text, kern, "WRITE_SIZE")
    est, _ = counter(text, "size_estimate_kernel", "FETCH_SIZE")
    avg = re.search(re.escape("| " + kern) + r" \| (\d+) \| \d+ \| (\d+) \|", text)
    read_b, write_b = int(rd * 128), int(ws * 1024)
    out = {
        "kernel": kern + " = <VEC,LDSMASK,SAMPLE,TILED,SRC,RET> (0/1-byte mask stream, fetch/compute tile steps, size-balanced launch "
                  "order, optimistic passes; RET=4: depth tiles of the first 5 steps per wave kept on chip for pass B, second workgroup "
                  "of every CU staggered)",
        "hbm_bytes_per_launch": read_b + write_b,
        "read_bytes_per_launch": read_b,
        "write_bytes_per_launch": write_b,
        "avg_kernel_ns_under_rocprof": int(avg.group(2)) if avg else None,
        "companion_kernels": {"size_estimate_kernel_read_bytes": int(est * 1024 * 1.999) if est else None},
        "kernel_source_sha256": bench.kernel_source_sha256(),
        "source": f"rocprofv3 --pmc TCC_EA0_RDREQ_sum / FETCH_SIZE / WRITE_SIZE (separate passes, profiles/run_profile.sh {tag}), mean over "
                  f"{n} dispatches of `bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined`; read = TCC_EA0_RDREQ_sum "
                  f"({rd:.0f}) x 128 B, cross-checked by FETCH_SIZE ({fs:.0f} KB) x 1024 x 1.999 = {fs * 1024 * 1.999 / 1e6:.1f} MB (the guide's "
                  "gfx950 x2 correction, calibrated on this kernel's own pattern by profiles/calibrate_fetch.sh); WRITE_SIZE(KB) x 1024 "
                  "uncalibrated; kernel_source_sha256 = bench.kernel_source_sha256() of the tree that was profiled",
        "summary": f"profiles/{tag}_instance_engine_B1024_summary.md",
    }
    dst = os.path.join(ROOT, "gpurun_out", "traffic_per_launch.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")

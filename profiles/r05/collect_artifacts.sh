#!/bin/bash
# After `gpurun -- bash profiles/r05_run.sh`: copy the measurement set from gpurun_out/ into profiles/ (tracked), regenerate the
# shader-side counter table and DESIGN.md section 5's table.   bash profiles/r05/collect_artifacts.sh
set -e
cd "$(dirname "$0")/../.."
cp gpurun_out/traffic_per_launch.json profiles/traffic_per_launch.json
for f in gpurun_out/r05/bench_*.json; do cp $f profiles/r05_$(basename $f); done
for f in gpurun_out/r05/profile_*.md; do b=$(basename $f .md); cp $f profiles/r05_${b#profile_}_summary.md; done
cp gpurun_out/r05/bench_aux_mi355x.json profiles/bench_aux_mi355x.json
cp gpurun_out/r05/small_batches_formats.txt profiles/r05/r05_small_batches.txt; cp gpurun_out/r05/per_image.txt profiles/r05/r05_per_image_raw.txt; cp gpurun_out/r05/host_pointer_latency.txt profiles/r05/r05_host_pointer_latency.txt
cp gpurun_out/r05/per_image_host.txt profiles/r05/r05_per_image_host.txt
python profiles/make_valu_json.py
python -c "import bench,json; a=bench.kernel_source_sha256(); print('traffic stamp current:', a==json.load(open('profiles/traffic_per_launch.json'))['kernel_source_sha256'], '| valu stamp current:', a==json.load(open('profiles/valu_per_launch.json'))['kernel_source_sha256'])"
python profiles/r05/make_design_table.py gpurun_out/r05 > profiles/r05/r05_design_table.md

#!/bin/bash
# After `gpurun -- bash profiles/r06_run.sh`: copy the measurement set from gpurun_out/ into profiles/ (tracked), regenerate the
# shader-side counter table and DESIGN.md section 5's table.   bash profiles/r06/collect_artifacts.sh
set -e
cd "$(dirname "$0")/../.."
cp gpurun_out/traffic_per_launch.json profiles/traffic_per_launch.json
for f in gpurun_out/r06/bench_*.json; do cp $f profiles/r06_$(basename $f); done
for f in gpurun_out/r06/profile_*.md; do b=$(basename $f .md); cp $f profiles/r06_${b#profile_}_summary.md; done
cp gpurun_out/r06/bench_aux_mi355x.json profiles/bench_aux_mi355x.json
cp gpurun_out/r06/small_batches_formats.txt profiles/r06/r06_small_batches.txt; cp gpurun_out/r06/per_image.txt profiles/r06/r06_per_image_raw.txt; cp gpurun_out/r06/host_pointer_latency.txt profiles/r06/r06_host_pointer_latency.txt
cp gpurun_out/r06/per_image_host.txt profiles/r06/r06_per_image_host.txt
cp gpurun_out/r06/rows_engine.txt profiles/r06/r06_rows_engine_final.txt; cp gpurun_out/r06/quick_bench.txt profiles/r06/r06_quick_bench.txt
python profiles/make_valu_json.py r06
python -c "import bench,json; a=bench.kernel_source_sha256(); print('traffic stamp current:', a==json.load(open('profiles/traffic_per_launch.json'))['kernel_source_sha256'], '| valu stamp current:', a==json.load(open('profiles/valu_per_launch.json'))['kernel_source_sha256'])"
python profiles/r06/make_design_table.py gpurun_out/r06 > profiles/r06/r06_design_table.md

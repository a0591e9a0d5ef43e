#!/bin/bash
# After `gpurun -- bash profiles/r04_run.sh`: copy the measurement set from gpurun_out/ into profiles/ (tracked), regenerate the
# shader-side counter table and DESIGN.md section 5's table.   bash profiles/r04/collect_artifacts.sh
set -e
cd "$(dirname "$0")/../.."
cp gpurun_out/traffic_per_launch.json profiles/traffic_per_launch.json
for f in gpurun_out/r04/bench_*.json; do cp $f profiles/r04_$(basename $f); done
for f in gpurun_out/r04/profile_*.md; do b=$(basename $f .md); cp $f profiles/r04_${b#profile_}_summary.md; done
cp gpurun_out/r04/bench_aux_mi355x.json profiles/bench_aux_mi355x.json
cp gpurun_out/r04/small_batches.txt profiles/r04/r04_small_batches_raw.txt; cp gpurun_out/r04/per_image.txt profiles/r04/r04_per_image_raw.txt
python profiles/make_valu_json.py
python -c "import bench,json; a=bench.kernel_source_sha256(); print('traffic stamp current:', a==json.load(open('profiles/traffic_per_launch.json'))['kernel_source_sha256'], '| valu stamp current:', a==json.load(open('profiles/valu_per_launch.json'))['kernel_source_sha256'])"
python profiles/r04/update_design_table.py

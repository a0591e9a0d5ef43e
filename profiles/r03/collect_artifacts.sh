#!/bin/bash
# After `gpurun -- bash profiles/r03_run.sh`: copy the measurement set from gpurun_out/ into profiles/ (tracked), regenerate the
# shader-side counter table and DESIGN.md section 5's table.   bash profiles/r03/collect_artifacts.sh
set -e
cd "$(dirname "$0")/../.."
cp gpurun_out/traffic_per_launch.json profiles/traffic_per_launch.json
for f in gpurun_out/r03/bench_*.json; do cp $f profiles/r03_$(basename $f); done
for f in gpurun_out/r03/profile_*.md; do b=$(basename $f .md); cp $f profiles/r03_${b#profile_}_summary.md; done
cp gpurun_out/r03/bench_aux_mi355x.json profiles/bench_aux_mi355x.json
python profiles/make_valu_json.py
python -c "import bench,json; a=bench.kernel_source_sha256(); print('traffic stamp current:', a==json.load(open('profiles/traffic_per_launch.json'))['kernel_source_sha256'], '| valu stamp current:', a==json.load(open('profiles/valu_per_launch.json'))['kernel_source_sha256'])"
python profiles/r03/update_design_table.py

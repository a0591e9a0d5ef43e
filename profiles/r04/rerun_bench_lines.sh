#!/bin/bash
# Re-run only the bench lines of profiles/r04_run.sh (no profiles) against the traffic table already in profiles/ - used when the
# table was refreshed after the lines were taken.   gpurun -- bash profiles/r04/rerun_bench_lines.sh ; bash profiles/r04/collect_artifacts.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; O=$REPO/gpurun_out/r04; mkdir -p $O; cd $REPO
cp profiles/traffic_per_launch.json gpurun_out/traffic_per_launch.json   # (run collect_artifacts.sh FIRST if r04_run.sh has just produced a new table: this copy goes the other way)
modes=("config2|" "config2_rle|--rle" "config2_poly|--poly" "config2_subsample|--subsample" "config2_area_hint|--area-hint" "config2_rle_area_hint|--rle --area-hint" \
       "config5|--config5" "config5_B16384|--config5 --batch 16384" "config3_5000|--config3 5000" "config3_14750|--config3 14750" \
       "config2_B256|--batch 256" "config2_B8192|--batch 8192")
for m in "${modes[@]}"; do tag=${m%%|*}; args=${m#*|}; python bench.py --no-cpu-baseline $args > $O/bench_$tag.json 2>/dev/null; done
LA3D_RETAIN=1 python bench.py --no-cpu-baseline > $O/bench_config2_retaining.json 2>/dev/null
python bench.py > $O/bench_default_final.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style_final.json 2>/dev/null
cp $O/bench_default_final.json $O/bench_default.json; cp $O/bench_driver_style_final.json $O/bench_driver_style.json

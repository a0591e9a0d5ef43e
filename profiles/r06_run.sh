#!/bin/bash
# Round-6 measurement set (run on the GPU box: gpurun -- bash profiles/r06_run.sh): every bench mode as the driver would run it
# + a light rocprofv3 profile (kernel stats and HBM counters from the SAME 55-launch command) of every mode, the full profile of
# the headline, the auxiliary kernels, and traffic_per_launch.json with one entry per mode.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06
mkdir -p $O
rm -f $REPO/gpurun_out/traffic_modes.jsonl
cd $REPO
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python bench.py > $O/bench_default.json 2>/dev/null
python bench.py --rotate 1 --no-cpu-baseline > $O/bench_default_rotate1.json 2>/dev/null   # the protocol of rounds 1-5: every step reads the same batch
modes=("config2|" "config2_rle|--rle" "config2_poly|--poly" "config2_subsample|--subsample" "config2_area_hint|--area-hint" "config2_rle_area_hint|--rle --area-hint" \
       "config5|--config5" "config5_B16384|--config5 --batch 16384" "config3_5000|--config3 5000" "config3_14750|--config3 14750" \
       "config2_B256|--batch 256" "config2_B8192|--batch 8192" "config2_ground|--ground")
for m in "${modes[@]}"; do
  tag=${m%%|*}; args=${m#*|}
  python bench.py --no-cpu-baseline $args > $O/bench_$tag.json 2>/dev/null
  if [ "$tag" == "config2" ]; then LIGHT=0 bash profiles/run_profile.sh r06_$tag $args > /dev/null 2>&1
  else LIGHT=1 bash profiles/run_profile.sh r06_$tag $args > /dev/null 2>&1; fi
  cp $REPO/gpurun_out/profile_r06_$tag.md $O/profile_$tag.md
done
# the headline workload with the two-pass plain build pinned (the default of round 4): what the separable single pass replaced
LA3D_SEP=0 python bench.py --no-cpu-baseline > $O/bench_config2_twopass.json 2>/dev/null
LA3D_SEP=0 LIGHT=1 bash profiles/run_profile.sh r06_config2_twopass > /dev/null 2>&1
cp $REPO/gpurun_out/profile_r06_config2_twopass.md $O/profile_config2_twopass.md
python profiles/make_traffic_json.py r06
# shader-side counters of the run-length / polygon / B = 8192 commands (the headline has them from its full profile above)
for m in "config2_rle|--rle" "config2_poly|--poly" "config2_B8192|--batch 8192"; do
  tag=${m%%|*}; args=${m#*|}
  LIGHT=0 bash profiles/run_profile.sh r06sq_$tag $args > /dev/null 2>&1
  cp $REPO/gpurun_out/profile_r06sq_$tag.md $O/profile_${tag}_sq.md
done
python profiles/bench_aux.py > $O/bench_aux_mi355x.json 2>/dev/null
# round 4: the north_star partitioning at one rank, the host-resident end-to-end rate, small batches, the per-image wrappers
python bench.py --no-cpu-baseline --config4 14750 > $O/bench_config4_14750.json 2>/dev/null
python bench.py --no-cpu-baseline --config4 14750 --poly > $O/bench_config4_14750_poly.json 2>/dev/null
python bench.py --no-cpu-baseline --config4 14750 --rle > $O/bench_config4_14750_rle.json 2>/dev/null
python bench.py --end-to-end 2048 > $O/bench_end_to_end.json 2>/dev/null
python profiles/r05/exp_small_batches.py > $O/small_batches_formats.txt 2>&1
python profiles/r06/exp_rows.py > $O/rows_engine.txt 2>&1
bash profiles/r06/quick_bench.sh > $O/quick_bench.txt 2>&1
python -m pytest tests/test_gpu_cabi.py -m gpu -q -s -k scalar_dropins 2>&1 | grep "host-pointer" > $O/host_pointer_latency.txt
L=labelany3d_amd/lib/libla3d.so
python profiles/r04/exp_per_image.py > $O/per_image.txt 2>&1
python profiles/r05/exp_per_image_host.py 2>&1 | grep annotations > $O/per_image_host.txt
# re-run the headline with the fresh traffic table in place (traffic_stale must read false)
cp $REPO/gpurun_out/traffic_per_launch.json $REPO/profiles/traffic_per_launch.json
python bench.py > $O/bench_default_final.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_style_final.json 2>/dev/null
ls -la $O

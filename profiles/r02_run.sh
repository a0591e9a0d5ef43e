#!/bin/bash
# Round-2 measurement run (on the GPU box through gpurun): bench lines for the default workload and the secondary modes, the
# auxiliary-kernel bench, and the rocprofv3 passes behind profiles/r02_*.  Everything lands in gpurun_out/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --config5 > gpurun_out/r02_bench_config5.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --config5 --batch 16384 --steps 50 --warmup 5 > gpurun_out/r02_bench_config5_16k.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --config3 5000 --steps 50 --warmup 5 > gpurun_out/r02_bench_config3.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --config3 14750 --steps 5 --warmup 1 > gpurun_out/r02_bench_config4_shard.json 2>> gpurun_out/r02_bench_default.err   # one GPU's share of config 4 (55 GB resident)
python bench.py --no-cpu-baseline --area-hint > gpurun_out/r02_bench_area_hint.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --rle > gpurun_out/r02_bench_rle.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --poly > gpurun_out/r02_bench_poly.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --subsample > gpurun_out/r02_bench_subsample.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --subsample --config5 > gpurun_out/r02_bench_subsample_config5.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --subsample --batch 8192 --steps 100 --warmup 10 > gpurun_out/r02_bench_subsample_B8192.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --batch 8192 --steps 100 --warmup 10 > gpurun_out/r02_bench_B8192.json 2>> gpurun_out/r02_bench_default.err
python bench.py --no-cpu-baseline --streams 2 > gpurun_out/r02_bench_streams2.json 2>> gpurun_out/r02_bench_default.err
LA3D_BALANCE=0 python bench.py --no-cpu-baseline --streams 2 > gpurun_out/r02_bench_streams2_nobal.json 2>> gpurun_out/r02_bench_default.err
python profiles/bench_aux.py > gpurun_out/r02_bench_aux.json 2> gpurun_out/r02_bench_aux.err
bash profiles/run_profile.sh r02 > gpurun_out/r02_profile_run.log 2>&1
python profiles/make_traffic_json.py r02 >> gpurun_out/r02_profile_run.log 2>&1
tail -3 gpurun_out/r02_bench_default.err
for f in default area_hint config5 config5_16k config3 config4_shard rle poly subsample subsample_config5 subsample_B8192 B8192 streams2 streams2_nobal; do python - <<PY
import json
d=json.load(open("gpurun_out/r02_bench_$f.json"))
r=d["roofline"]
print("$f", round(d["value"]/1e6,2),"M boxes/s", round(d["ms_per_step"]*1e3,1),"us/step frac",round(r["frac"],3),"req MB",round(r["required_bytes_per_launch"]/1e6,1), "traffic_GBps", r.get("traffic_GBps"))
PY
done

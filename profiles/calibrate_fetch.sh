#!/bin/bash
# FETCH_SIZE calibration on a known byte count in this kernel's own access pattern: the phase-0-only
# build (passes A and B compiled out) streams exactly B*H*W mask bytes with 16-B non-temporal loads.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_calib
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
LA3D_LIB=$REPO/build/abl/base_DLA3D_ABL_NO_PASSB_DLA3D_ABL_NO_PASSA.so rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $REPO/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1
python - <<PY
import csv,glob
rows=[r for f in glob.glob("$OUT/fetch/*counter_collection.csv") for r in csv.DictReader(open(f)) if "fit_instances_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
v=[float(r["Counter_Value"]) for r in rows]
known=1024*480*640
print(f"phase-0-only kernel: FETCH_SIZE mean {sum(v)/len(v):.1f} (KB) over {len(v)} dispatches; known bytes read = {known} = {known/1024:.0f} KB; ratio known/reported = {known/1024/(sum(v)/len(v)):.3f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete

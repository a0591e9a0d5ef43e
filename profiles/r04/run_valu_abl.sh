#!/bin/bash
# round 4: where the VALU instructions of a launch go - SQ_INSTS_VALU / kernel time of ablation builds (LA3D_ABL_*), B = 1024
O=$GRAFT_REPO_ROOT/gpurun_out/r04abl; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for mode in "--rle" "--poly" ""; do
for tag in full nocull nopb nopab nodec; do
  lib=build/abl/libla3d_$tag.so
  m=${mode#--}; m=${m:-u8plain}
  LA3D_RETAIN=0 LA3D_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --output-format csv --kernel-trace --stats --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/${m}_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined $mode > $O/${m}_$tag.log 2>&1
  python - <<PY
import csv,glob
tot={}
for f in glob.glob("$O/${m}_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_instances_kernel" in r["Kernel_Name"]:
            tot.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
dur=[]
for f in glob.glob("$O/${m}_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_instances_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("$m $tag", {k:round(sum(v)/len(v)/1e6,3) for k,v in tot.items()}, "M per launch;", "kernel us", round(sum(dur)/max(len(dur),1),1), len(dur))
PY
done; done 2>&1 | tee $O/summary.txt

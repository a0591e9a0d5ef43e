#!/bin/bash
# SQ_INSTS_VALU of the fit kernel for run-length input: before / after the SPEC forms (is the specialised path taken?)
O=$GRAFT_REPO_ROOT/gpurun_out/r04valu; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for v in nospec:build/abl/libla3d_nospec.so cur:labelany3d_amd/lib/libla3d.so; do
  tag=${v%%:*}; lib=${v#*:}
  LA3D_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --output-format csv --kernel-trace --stats --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --rle > $O/$tag.log 2>&1
  python - <<PY
import csv,glob
tot={}
for f in glob.glob("$O/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_instances_kernel" in r["Kernel_Name"]:
            tot.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("$tag", {k:(sum(v)/len(v), len(v)) for k,v in tot.items()})
PY
done

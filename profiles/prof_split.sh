set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_split
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for sub in 1 4; do
LA3D_ENGINE=split LA3D_SPLIT_SUB=$sub rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/s$sub -o s -- python $REPO/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/s$sub.log 2>&1
echo "== sub $sub"; python - <<PY
import csv,glob
for f in glob.glob("$OUT/s$sub/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "la3d" in r["Name"]: print(r["Name"].split("(")[0][-40:], r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete

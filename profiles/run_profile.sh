#!/bin/bash
# Profiling recipe for one bench command (run on the GPU box through gpurun).
#   profiles/run_profile.sh <tag> [extra bench args]          LIGHT=1: kernel stats + HBM counters only (no shader-side passes)
# Writes rocprofv3 CSV outputs under gpurun_out/prof_<tag>/ and a condensed markdown summary gpurun_out/profile_<tag>.md
# (copy the summaries that matter into profiles/).  Kernel time AND counters come from the SAME command (every counter pass also
# collects --kernel-trace --stats: the summary prints the kernel table of each pass); the default bench command (1000 steps) is
# profiled on top of that unless LIGHT=1, because a 55-launch run catches the chip before its clocks have settled.
set -u
TAG=${1:-r03}; shift || true
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined --no-steady --no-same-batch $EXTRA"
# 1) per-kernel time of the default-length command
if [ "${LIGHT:-0}" != "1" ]; then
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --no-cpu-baseline --no-pipelined --no-same-batch $EXTRA > $OUT/stats.log 2>&1
fi
# 2) HBM traffic counters, one pass each (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2); never with sys/hip traces
rocprofv3 --output-format csv --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/tcc -o tcc -- $BENCH > $OUT/tcc.log 2>&1
# 3) shader-side counters
if [ "${LIGHT:-0}" != "1" ]; then
  rocprofv3 --output-format csv --kernel-trace --stats --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --stats --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o sq2 -- $BENCH > $OUT/sq2.log 2>&1
fi
python $REPO/profiles/summarize.py $OUT $REPO/gpurun_out/profile_$TAG.md > /dev/null
python $REPO/profiles/collect_traffic.py $OUT $TAG 55 >> $REPO/gpurun_out/traffic_modes.jsonl
# keep only the small summaries: traces and databases are bulky
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT

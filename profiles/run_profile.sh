#!/bin/bash
# Profiling recipe for the bench command (run on the GPU box through gpurun).
#   profiles/run_profile.sh <tag> [extra bench args]
# Writes rocprofv3 CSV outputs under gpurun_out/prof_<tag>/ and a condensed markdown summary
# gpurun_out/profile_<tag>.md (copy the summaries that matter into profiles/).
set -u
TAG=${1:-r01}; shift || true
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined $EXTRA"
# 1) per-kernel time: the default command itself (1000 timed steps: the steady state bench.py measures; a 50-step run catches the chip
#    before its clocks have settled and reads 4-5 % longer kernels)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --no-cpu-baseline --no-pipelined $EXTRA > $OUT/stats.log 2>&1
# 2) HBM traffic counters, one pass each (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2); never with sys/hip traces
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
# 3) shader-side counters
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o sq2 -- $BENCH > $OUT/sq2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/tcc -o tcc -- $BENCH > $OUT/tcc.log 2>&1
python $REPO/profiles/summarize.py $OUT $REPO/gpurun_out/profile_$TAG.md > /dev/null
# keep only the small summaries: traces and databases are bulky
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT

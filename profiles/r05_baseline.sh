#!/bin/bash
# Round-5 opening measurement: test suite, the contract's command under the round-5 protocol (no steady loop before the timed region), timeline.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r05
mkdir -p $O
cd $REPO
python -m pytest tests -m gpu -x -q > $O/pytest_open.log 2>&1; echo "pytest rc=$?" >> $O/pytest_open.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/open_driver_$i.json 2>/dev/null; done
python bench.py --no-cpu-baseline > $O/open_default.json 2>/dev/null
python bench.py --no-cpu-baseline --rle > $O/open_rle.json 2>/dev/null
python bench.py --no-cpu-baseline --poly > $O/open_poly.json 2>/dev/null
bash profiles/timeline.sh 1024 > $O/open_timeline.txt 2>&1
tail -3 $O/pytest_open.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05/open_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(os.path.basename(f), round(d['value']/1e6,3),'M', round(d['ms_per_step']*1e3,2),'us wall', round(d['roofline']['avg_launch_ms']*1e3,2),'us ev', 'steady', d.get('steady_state',{}).get('ms_per_step'))
    except Exception as e: print(f, e)
PY

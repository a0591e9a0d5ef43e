#!/bin/bash
# Round-5 iteration step: (optional) tests, then the bench lines that matter; one table at the end.
#   bash profiles/r05_step.sh <tag> [pytest args...]     (TESTS=0 skips pytest; TL=1 adds the per-phase timeline)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-step}; shift || true
O=$REPO/gpurun_out/r05
mkdir -p $O
cd $REPO
if [ "${TESTS:-1}" != "0" ]; then
  python -m pytest tests -m gpu -x -q "$@" > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
  tail -15 $O/${TAG}_pytest.log
fi
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/${TAG}_driver_$i.json 2>/dev/null; done
python bench.py --no-cpu-baseline > $O/${TAG}_default.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipelined --rle > $O/${TAG}_rle.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipelined --poly > $O/${TAG}_poly.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipelined --config5 > $O/${TAG}_config5.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipelined --batch 8192 --steps 100 > $O/${TAG}_B8192.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipelined --config3 5000 --steps 30 > $O/${TAG}_config3.json 2>/dev/null
if [ "${TL:-0}" == "1" ]; then bash profiles/timeline.sh 1024 > $O/${TAG}_timeline.txt 2>&1; fi
python - "$O" "$TAG" <<'PY'
import json,glob,os,sys
O,TAG=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(f"{O}/{TAG}_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        r=d['roofline']
        print(f"{os.path.basename(f):28s} {d['value']/1e6:7.3f} M  wall {d['ms_per_step']*1e3:8.2f} us  ev {r['avg_launch_ms']*1e3:8.2f} us  frac_req {r['frac']:.3f}  steady {(d.get('steady_state') or {}).get('ms_per_step')}  pipelined {(d.get('pipelined') or {}).get('ms_per_step')}")
    except Exception as e: print(f, 'ERR', e)
PY

#!/bin/bash
# round 4, experiment 6: two-launch form (first resident set in plain order, remainder estimated + ordered on an internal stream)
O=gpurun_out/r04two1; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 900 python profiles/sweep_variants.py base=$L two=$L,LA3D_TWO_LAUNCH=1 base2=$L two2=$L,LA3D_TWO_LAUNCH=1 hint_free_order_off=$L,LA3D_BALANCE=0 --batches 640,768,896,1024 > $O/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py base=$L two=$L,LA3D_TWO_LAUNCH=1 --batches 768,1024 --config5 > $O/sweep_c5.txt 2>&1
LA3D_TWO_LAUNCH=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_two_driver.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_base_driver.json 2>/dev/null
LA3D_TWO_LAUNCH=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_two.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline > $O/bench_base.json 2>/dev/null
echo "== sweep"; tail -8 $O/sweep.txt | cut -c1-400; echo "== c5"; tail -3 $O/sweep_c5.txt | cut -c1-300
python - <<'PY'
import json
for n in ("bench_base_driver","bench_two_driver","bench_base","bench_two"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r04two1/{n}.json") if l.startswith("{")][-1]); print(n, round(d["value"]/1e6,3), round(d["ms_per_step"]*1e3,1), round(d["roofline"]["avg_launch_ms"]*1e3,1))
    except Exception as e: print(n,"failed",e)
PY

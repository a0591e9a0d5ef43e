#!/bin/bash
# round 4, experiment 3: the band engine (two workgroups per instance) - parity, then A/B timing
O=gpurun_out/r04band1; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 600 python -m pytest tests/test_gpu_band.py -x -q 2>&1 | tail -25 > $O/tests_band.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_all.txt
timeout 1200 python profiles/sweep_variants.py inst=$L,LA3D_ENGINE=instance band2=$L,LA3D_ENGINE=band band4=$L,LA3D_ENGINE=band,LA3D_BANDS=4 band2_noorder=$L,LA3D_ENGINE=band,LA3D_BALANCE=0 --batches 384,512,1024,1536,2048,4096 > $O/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py inst=$L,LA3D_ENGINE=instance band2=$L,LA3D_ENGINE=band band4=$L,LA3D_ENGINE=band,LA3D_BANDS=4 --batches 1024,4096 --config5 > $O/sweep_c5.txt 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_style.json 2>$O/bench_driver_style.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>$O/bench_default.err
for f in tests_band tests_all; do echo "== $f"; cat $O/$f.txt; done
echo "== sweep"; tail -40 $O/sweep.txt | cut -c1-420; echo "== c5"; tail -14 $O/sweep_c5.txt
python - <<'PY'
import json
for n in ("bench_driver_style","bench_default"):
    try:
        d=json.load(open(f"gpurun_out/r04band1/{n}.json")); print(n, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("pipelined",{}).get("ms_per_step"))
    except Exception as e: print(n,"failed",e)
PY

#!/bin/bash
# round 4, experiment 5: the un-grounded / skew-free forms of the pixel math (SPEC) - whole suite, A/B timing, tools
O=gpurun_out/r04spec1; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so; A=build/abl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/tests_all.txt
timeout 1500 python profiles/sweep_variants.py nospec=$A/libla3d_nospec.so spec=$L nospec_plain=$A/libla3d_nospec.so,LA3D_RETAIN=0 spec_plain=$L,LA3D_RETAIN=0 --batches 512,1024,2048,8192 --rle --poly --config3 800 > $O/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py nospec=$A/libla3d_nospec.so spec=$L --batches 1024,16384 --config5 > $O/sweep_c5.txt 2>&1
timeout 300 python profiles/r04/exp_per_image.py > $O/per_image.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --config4 2000 > $O/bench_config4.json 2>$O/bench_config4.err
timeout 600 python bench.py --end-to-end 1024 > $O/bench_e2e.json 2>$O/bench_e2e.err
cat $O/tests_all.txt
echo "== sweep"; tail -8 $O/sweep.txt | cut -c1-620; echo "== c5"; tail -4 $O/sweep_c5.txt
echo "== per image"; tail -8 $O/per_image.txt | cut -c1-300
python - <<'PY'
import json
for n in ("bench_config4","bench_e2e"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r04spec1/{n}.json") if l.startswith("{")][-1])
        print(n, {k:d[k] for k in d if k in ("value","ms_per_step","per_rank_fit_ms","gather_ms","images_per_s","split_s","host_link")})
    except Exception as e: print(n,"failed",e, open(f"gpurun_out/r04spec1/{n}.err").read()[-800:])
PY

#!/bin/bash
# round 4, experiment 2: the whole GPU suite on the refactored library (per-call options, pass-B culling), culling statistics, A/B timing
O=gpurun_out/r04cull2; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so; A=build/abl
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_all.txt
LA3D_LIB=$A/libla3d_cull1.so LA3D_RETAIN=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_masks.py tests/test_gpu_poly.py tests/test_gpu_cull.py -x -q 2>&1 | tail -15 > $O/tests_min1_plain.txt
LA3D_LIB=$A/libla3d_stats.so timeout 300 python profiles/r04/cull_stats.py > $O/cull_stats.txt 2>&1
timeout 300 python profiles/r04/debug_cull_diff.py 2>&1 | head -20 > $O/diff.txt
timeout 1500 python profiles/sweep_variants.py ret=$L plain_nocull=$A/libla3d_nocull.so,LA3D_RETAIN=0 plain_cull96=$L,LA3D_RETAIN=0 plain_cull160=$A/libla3d_cull160.so,LA3D_RETAIN=0 plain_cull224=$A/libla3d_cull224.so,LA3D_RETAIN=0 --batches 512,1024,2048,8192 --rle --poly --config3 800 > $O/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py ret=$L plain_nocull=$A/libla3d_nocull.so,LA3D_RETAIN=0 plain_cull96=$L,LA3D_RETAIN=0 plain_cull224=$A/libla3d_cull224.so,LA3D_RETAIN=0 --batches 1024,16384 --config5 > $O/sweep_c5.txt 2>&1
for f in tests_all tests_min1_plain cull_stats diff; do echo "== $f"; cat $O/$f.txt; done
echo "== sweep"; tail -40 $O/sweep.txt | cut -c1-420; echo "== c5"; tail -14 $O/sweep_c5.txt

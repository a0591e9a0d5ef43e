#!/bin/bash
# round 4, experiment 1: pass-B tile culling in the plain build - parity first, then A/B timing
O=gpurun_out/r04cull1; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so; A=build/abl
timeout 600 python -m pytest tests/test_gpu_cull.py -x -q 2>&1 | tail -15 > $O/tests_cull.txt
LA3D_LIB=$A/libla3d_cull1.so timeout 600 python -m pytest tests/test_gpu_cull.py -x -q 2>&1 | tail -15 > $O/tests_cull_min1.txt
LA3D_LIB=$A/libla3d_cull1.so LA3D_RETAIN=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_masks.py tests/test_gpu_poly.py -x -q 2>&1 | tail -15 > $O/tests_parity_min1_plain.txt
timeout 1200 python profiles/sweep_variants.py ret=$L plain_nocull=$A/libla3d_nocull.so,LA3D_RETAIN=0 plain_cull96=$L,LA3D_RETAIN=0 plain_cull48=$A/libla3d_cull48.so,LA3D_RETAIN=0 plain_cull160=$A/libla3d_cull160.so,LA3D_RETAIN=0 --batches 512,1024,2048,8192 --rle --poly --config3 800 > $O/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py ret=$L plain_nocull=$A/libla3d_nocull.so,LA3D_RETAIN=0 plain_cull96=$L,LA3D_RETAIN=0 --batches 1024,16384 --config5 > $O/sweep_c5.txt 2>&1
for f in tests_cull tests_cull_min1 tests_parity_min1_plain; do echo "== $f"; cat $O/$f.txt; done
echo "== sweep"; tail -40 $O/sweep.txt; echo "== c5"; tail -14 $O/sweep_c5.txt

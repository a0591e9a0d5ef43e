#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04tcs; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python profiles/sweep_variants.py pre=build/abl/libla3d_pre_tcs.so new=labelany3d_amd/lib/libla3d.so pre2=build/abl/libla3d_pre_tcs.so new2=labelany3d_amd/lib/libla3d.so --batches 1024,8192 --rle --poly --config3 800 > $O/sweep.txt 2>&1
tail -5 $O/sweep.txt | cut -c1-520
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cull.py tests/test_gpu_band.py tests/test_gpu_poly.py -q 2>&1 | tail -3
bash profiles/r04/run_valu_ab.sh 2>&1 | tail -2

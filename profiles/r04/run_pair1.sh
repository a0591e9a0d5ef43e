#!/bin/bash
# round 4, experiment 7: the pair engine (one 16-wave workgroup per pair of instances, passes shared through LDS)
O=gpurun_out/r04pair1; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 600 python -m pytest tests/test_gpu_pair.py -x -q 2>&1 | tail -25 > $O/tests_pair.txt
timeout 1200 python profiles/sweep_variants.py inst=$L,LA3D_ENGINE=instance inst_plain=$L,LA3D_ENGINE=instance,LA3D_RETAIN=0 pair=$L,LA3D_ENGINE=pair pair_noorder=$L,LA3D_ENGINE=pair,LA3D_BALANCE=0 --batches 256,512,768,1024,1536,2048,4096 > $O/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py inst=$L,LA3D_ENGINE=instance pair=$L,LA3D_ENGINE=pair --batches 1024,4096 --config5 > $O/sweep_c5.txt 2>&1
cat $O/tests_pair.txt
echo "== sweep"; tail -6 $O/sweep.txt | cut -c1-520; echo "== c5"; tail -3 $O/sweep_c5.txt

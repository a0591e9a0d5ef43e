#!/bin/bash
# round 4, experiment 14: self-estimating launch (no helper kernel in front of ordered launches of up to one resident set)
O=gpurun_out/r04self; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "launch_order or full_size or config5" 2>&1 | tail -3 > $O/tests.txt
V=""
for r in 1 2; do V="$V self_$r=$L,LA3D_ENGINE=instance helper_$r=$L,LA3D_ENGINE=instance,LA3D_ORDER_SELF=0"; done
timeout 1500 python profiles/sweep_variants.py $V --batches 320,512,768,1024,1536 --rle > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py $V --batches 512,1024 --config5 > $O/sweep_c5.txt 2>&1
cat $O/tests.txt; echo "== c2"; tail -4 $O/sweep.txt | cut -c1-620; echo "== c5"; tail -4 $O/sweep_c5.txt | cut -c1-300

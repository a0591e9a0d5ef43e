#!/bin/bash
# round 4, experiment 14b: self-estimating launch also above one resident set (the first set's workgroups estimate everything)
O=gpurun_out/r04self2; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V=""
for r in 1 2; do V="$V self_$r=$L,LA3D_ENGINE=instance helper_$r=$L,LA3D_ENGINE=instance,LA3D_ORDER_SELF=0"; done
timeout 1500 python profiles/sweep_variants.py $V --batches 1024,1280,1536,2048,3072 --rle > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py $V --batches 1536,2048 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -4 $O/sweep.txt | cut -c1-620; echo "== c5"; tail -4 $O/sweep_c5.txt | cut -c1-300

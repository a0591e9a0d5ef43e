#!/bin/bash
# round 4: stagger period re-checked on top of the self-estimating launch
O=gpurun_out/r04stag4; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V=""
for r in 1 2; do for u in 6 8 10 12 14; do V="$V g${u}_$r=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u"; done; done
timeout 1500 python profiles/sweep_variants.py $V --batches 512,1024,1536 > $O/sweep.txt 2>&1
timeout 1200 python profiles/sweep_variants.py $V --batches 512,1024 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -10 $O/sweep.txt | cut -c1-320; echo "== c5"; tail -10 $O/sweep_c5.txt | cut -c1-220

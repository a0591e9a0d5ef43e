#!/bin/bash
# round 4, experiment 10: wave priorities by launch-order group on top of the stagger
O=gpurun_out/r04prio; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V="p0=$L,LA3D_ENGINE=instance"
for m in 1 2 3 4; do V="$V p$m=$L,LA3D_ENGINE=instance,LA3D_PRIO_MODE=$m"; done
timeout 1500 python profiles/sweep_variants.py $V --batches 512,768,1024,1280,2048 > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py $V --batches 1024 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -5 $O/sweep.txt | cut -c1-420; echo "== c5"; tail -5 $O/sweep_c5.txt | cut -c1-200

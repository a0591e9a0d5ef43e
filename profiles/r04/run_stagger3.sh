#!/bin/bash
# round 4, experiment 9c: grouped stagger of the plain build - period sweep, large batches, run lengths / polygons untouched
O=gpurun_out/r04stag3; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V="s0=$L,LA3D_ENGINE=instance"
for u in 8 10 12 14 16 20; do V="$V g$u=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u"; done
timeout 1500 python profiles/sweep_variants.py $V --batches 448,640,896,1024,1152,2048,4096,8192 > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py s0=$L,LA3D_ENGINE=instance g10=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=10 g14=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=14 --batches 512,1024,2048,16384 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -7 $O/sweep.txt | cut -c1-620; echo "== c5"; tail -3 $O/sweep_c5.txt | cut -c1-400

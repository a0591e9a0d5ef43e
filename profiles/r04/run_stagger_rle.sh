#!/bin/bash
# round 4, experiment 9: staggered start of the plain build's four resident groups (LA3D_STAGGER_US), run lengths and u8 planes
O=gpurun_out/r04stag; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V="s0=$L,LA3D_ENGINE=instance"
for u in 2 4 6 8 12; do V="$V s$u=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u"; done
timeout 900 python profiles/sweep_variants.py $V --batches 1024 --rle > $O/sweep_rle.txt 2>&1
echo "== rle/u8"; tail -8 $O/sweep_rle.txt | cut -c1-300

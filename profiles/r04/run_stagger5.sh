#!/bin/bash
# round 4, experiment 17: non-uniform delays of the three later groups (LA3D_STAGGER_PAT=a:b:c us), two repetitions
O=gpurun_out/r04stag5; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V=""
for r in 1 2; do
V="$V u12_$r=$L,LA3D_ENGINE=instance"
for pat in 10:20:30 8:18:30 12:22:30 14:24:32 10:22:36 6:16:28 12:20:26; do V="$V p${pat//:/_}_$r=$L,LA3D_ENGINE=instance,LA3D_STAGGER_PAT=$pat"; done; done
timeout 1800 python profiles/sweep_variants.py $V --batches 512,1024,1536 > $O/sweep.txt 2>&1
timeout 1500 python profiles/sweep_variants.py $V --batches 512,1024 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -16 $O/sweep.txt | cut -c1-320; echo "== c5"; tail -16 $O/sweep_c5.txt | cut -c1-220

#!/bin/bash
# round 4, experiment 9b: stagger patterns of the plain build, u8 planes
O=gpurun_out/r04stag2; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V="s0=$L,LA3D_ENGINE=instance"
for u in 6 8 10; do V="$V g$u=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u"; done
for u in 4 6 8 10 12; do V="$V c$u=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u,LA3D_STAGGER_MODE=1"; done
for u in 6 10; do V="$V h$u=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u,LA3D_STAGGER_MODE=2"; done
for u in 6 12 18; do V="$V f$u=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=$u,LA3D_STAGGER_MODE=3"; done
timeout 1500 python profiles/sweep_variants.py $V --batches 512,768,1024,1280,1536 > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py s0=$L,LA3D_ENGINE=instance g8=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=8 c8=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=8,LA3D_STAGGER_MODE=1 c12=$L,LA3D_ENGINE=instance,LA3D_STAGGER_US=12,LA3D_STAGGER_MODE=1 --batches 768,1024,1536 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -14 $O/sweep.txt | cut -c1-420; echo "== c5"; tail -4 $O/sweep_c5.txt | cut -c1-300

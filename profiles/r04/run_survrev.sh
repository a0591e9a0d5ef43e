#!/bin/bash
# round 4, experiment 11: pass B takes the survivor list from its end (L2 reuse of the tiles pass A read last)
O=gpurun_out/r04survrev; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so; A=build/abl
V="fwd1=$L,LA3D_ENGINE=instance rev1=$A/libla3d_survrev.so,LA3D_ENGINE=instance fwd2=$L,LA3D_ENGINE=instance rev2=$A/libla3d_survrev.so,LA3D_ENGINE=instance"
timeout 1500 python profiles/sweep_variants.py $V --batches 512,1024,2048,8192 --rle > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py $V --batches 1024,16384 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -4 $O/sweep.txt | cut -c1-520; echo "== c5"; tail -4 $O/sweep_c5.txt | cut -c1-300

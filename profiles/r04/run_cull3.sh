#!/bin/bash
# round 4: the culling threshold after the cheaper tile range (26 instead of 56 instructions per tile)
O=gpurun_out/r04cull3; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so; A=build/abl
V="c224=$L,LA3D_ENGINE=instance"
for v in 160 128 96 64; do V="$V c$v=$A/libla3d_cull$v.so,LA3D_ENGINE=instance"; done
timeout 1500 python profiles/sweep_variants.py $V --batches 512,1024,2048,8192 --rle > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py $V --batches 1024,2048,16384 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -5 $O/sweep.txt | cut -c1-520; echo "== c5"; tail -5 $O/sweep_c5.txt | cut -c1-300

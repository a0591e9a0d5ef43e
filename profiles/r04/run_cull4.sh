#!/bin/bash
# round 4: culling threshold for u8 planes (runtime LA3D_CULL_MIN_U8), repeated A/B
O=gpurun_out/r04cull4; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V=""
for r in 1 2; do for v in 224 160 128 96; do V="$V c${v}_$r=$L,LA3D_ENGINE=instance,LA3D_CULL_MIN_U8=$v"; done; done
timeout 1500 python profiles/sweep_variants.py $V --batches 512,768,1024,1536 > $O/sweep.txt 2>&1
timeout 1500 python profiles/sweep_variants.py $V --batches 512,1024,2048,16384 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -8 $O/sweep.txt | cut -c1-420; echo "== c5"; tail -8 $O/sweep_c5.txt | cut -c1-420

#!/bin/bash
# round 4, experiment 13: issue priority by size group for run-length input (VALU-bound per CU: the largest instance of a CU first)
O=gpurun_out/r04prio_rle; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V=""
for r in 1 2; do for m in 0 1 2 3; do V="$V p${m}_$r=$L,LA3D_ENGINE=instance,LA3D_PRIO_MODE=$m"; done; done
timeout 1500 python profiles/sweep_variants.py $V --batches 1024 --rle > $O/sweep.txt 2>&1
echo "== rle"; tail -8 $O/sweep.txt | cut -c1-220

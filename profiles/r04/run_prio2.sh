#!/bin/bash
# round 4, experiment 15: higher issue priority for the later (smaller) groups AFTER their mask stream (their short post-stream chain is the launch's tail)
O=gpurun_out/r04prio2; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
V=""
for r in 1 2; do for m in 0 1 2 3 4; do V="$V p${m}_$r=$L,LA3D_ENGINE=instance,LA3D_PRIO_MODE=$m"; done; done
timeout 1500 python profiles/sweep_variants.py $V --batches 512,1024,1536 > $O/sweep.txt 2>&1
timeout 900 python profiles/sweep_variants.py $V --batches 1024 --config5 > $O/sweep_c5.txt 2>&1
echo "== c2"; tail -10 $O/sweep.txt | cut -c1-320; echo "== c5"; tail -10 $O/sweep_c5.txt | cut -c1-200

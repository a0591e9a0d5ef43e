#!/bin/bash
# round 4, experiment 4: the band engine on SMALL batches (one launch instead of the split engine's chain of six)
O=gpurun_out/r04band2; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 1200 python profiles/sweep_variants.py default=$L,LA3D_ENGINE=,LA3D_BAND_DEFAULT=0 split=$L,LA3D_ENGINE=split inst=$L,LA3D_ENGINE=instance band2=$L,LA3D_ENGINE=band band4=$L,LA3D_ENGINE=band,LA3D_BANDS=4 --batches 1,4,16,32,64,128,192,256,320 > $O/sweep_small.txt 2>&1
echo "== small"; tail -12 $O/sweep_small.txt | cut -c1-700

#!/bin/bash
# round 4, experiment 8b: mixed grid with the alternating deal of the single workgroups
O=gpurun_out/r04mix2; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 600 python -m pytest tests/test_gpu_band.py -x -q 2>&1 | tail -5 > $O/tests_band.txt
V="inst=$L,LA3D_ENGINE=instance inst_plain=$L,LA3D_ENGINE=instance,LA3D_RETAIN=0 inst_plain_noorder=$L,LA3D_ENGINE=instance,LA3D_RETAIN=0,LA3D_BALANCE=0"
for nb in 4 2; do for k in 32 64 128; do V="$V mix${nb}_$k=$L,LA3D_ENGINE=band,LA3D_BANDS=$nb,LA3D_BAND_TOPK=$k"; done; done
V="$V one=$L,LA3D_ENGINE=band,LA3D_BANDS=2,LA3D_BAND_TOPK=8 one_nopair=$L,LA3D_ENGINE=band,LA3D_BANDS=2,LA3D_BAND_TOPK=8,LA3D_BAND_PAIRING=0"
timeout 1500 python profiles/sweep_variants.py $V --batches 768,1024,1536 > $O/sweep.txt 2>&1
cat $O/tests_band.txt
echo "== sweep"; tail -12 $O/sweep.txt | cut -c1-420

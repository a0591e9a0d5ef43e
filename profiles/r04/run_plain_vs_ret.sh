#!/bin/bash
# round 4: plain (64 VGPRs, 4 workgroups per CU) vs retaining (128 VGPRs, 2 per CU + stagger) build after the VALU diet
O=gpurun_out/r04pvr; mkdir -p $O
L=labelany3d_amd/lib/libla3d.so
timeout 900 python profiles/sweep_variants.py ret=$L,LA3D_ENGINE=instance,LA3D_RETAIN=1 plain=$L,LA3D_ENGINE=instance,LA3D_RETAIN=0 --batches 448,512,768,1024,1280,1536 > $O/sweep_c2.txt 2>&1
timeout 900 python profiles/sweep_variants.py ret=$L,LA3D_ENGINE=instance,LA3D_RETAIN=1 plain=$L,LA3D_ENGINE=instance,LA3D_RETAIN=0 --batches 448,512,768,1024,1280,1536 --config5 > $O/sweep_c5.txt 2>&1
echo "== config 2"; tail -2 $O/sweep_c2.txt | cut -c1-600; echo "== config 5"; tail -2 $O/sweep_c5.txt | cut -c1-600

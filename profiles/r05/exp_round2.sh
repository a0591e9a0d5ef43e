#!/bin/bash
# experiments after the separable pass: launch order off, stagger for run-length / polygon input
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; O=$REPO/gpurun_out/r05; mkdir -p $O; cd $REPO
L=labelany3d_amd/lib/libla3d.so
python profiles/sweep_variants.py default=$L nobalance=$L,LA3D_BALANCE=0 helper=$L,LA3D_ORDER_SELF=0 --batches 512,1024,1536 > $O/exp2_order.txt 2>&1
python profiles/sweep_variants.py n0=$L n2=$L,LA3D_STAGGER_NOMASK_US=2 n4=$L,LA3D_STAGGER_NOMASK_US=4 n6=$L,LA3D_STAGGER_NOMASK_US=6 n8=$L,LA3D_STAGGER_NOMASK_US=8 n12=$L,LA3D_STAGGER_NOMASK_US=12 --rle --poly --batches 1024 > $O/exp2_stagger_nomask.txt 2>&1
tail -4 $O/exp2_order.txt; tail -8 $O/exp2_stagger_nomask.txt

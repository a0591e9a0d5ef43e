#!/bin/bash
# Round 5, after the separable single pass: which engine for which batch size now, the stagger period again, and the fixed tests.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r05
mkdir -p $O
cd $REPO
python -m pytest tests/test_gpu_shard.py tests/test_gpu_cabi.py tests/test_gpu_sep.py tests/test_gpu_cull.py -m gpu -x -q -s 2>&1 | grep -v "^\[pca\]\|^\[convex" | tail -12 > $O/sweeps1_pytest.log
L=labelany3d_amd/lib/libla3d.so
python profiles/sweep_variants.py default=$L split=$L,LA3D_ENGINE=split instance=$L,LA3D_ENGINE=instance band4=$L,LA3D_ENGINE=band,LA3D_BANDS=4 band2=$L,LA3D_ENGINE=band,LA3D_BANDS=2 inst2pass=$L,LA3D_ENGINE=instance,LA3D_SEP=0 --batches 1,4,16,32,64,128,192,256,320,384,512,768,1024 > $O/sweeps1_small_batches.txt 2>&1
python profiles/sweep_variants.py s0=$L,LA3D_STAGGER_US=0 s6=$L,LA3D_STAGGER_US=6 s8=$L,LA3D_STAGGER_US=8 s10=$L,LA3D_STAGGER_US=10 s12=$L s14=$L,LA3D_STAGGER_US=14 s17=$L,LA3D_STAGGER_US=17 --batches 512,1024,1536,2048 > $O/sweeps1_stagger.txt 2>&1
python profiles/sweep_variants.py s0=$L,LA3D_STAGGER_US=0 s6=$L,LA3D_STAGGER_US=6 s8=$L,LA3D_STAGGER_US=8 s10=$L,LA3D_STAGGER_US=10 s12=$L s14=$L,LA3D_STAGGER_US=14 s17=$L,LA3D_STAGGER_US=17 --batches 512,1024,2048 --config5 > $O/sweeps1_stagger_c5.txt 2>&1
python profiles/sweep_variants.py default=$L split=$L,LA3D_ENGINE=split instance=$L,LA3D_ENGINE=instance --rle --poly --batches 1024 > $O/sweeps1_rle_poly.txt 2>&1
python bench.py --no-cpu-baseline --no-pipelined --rle > $O/sweeps1_rle.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipelined --rle > $O/sweeps1_rle_again.json 2>/dev/null
cat $O/sweeps1_pytest.log; tail -25 $O/sweeps1_small_batches.txt; tail -12 $O/sweeps1_stagger.txt; tail -10 $O/sweeps1_stagger_c5.txt; tail -8 $O/sweeps1_rle_poly.txt
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05'
for f in ('sweeps1_rle.json','sweeps1_rle_again.json'):
    d=json.loads([l for l in open(O+'/'+f) if l.startswith('{')][0]); print(f, d['ms_per_step']*1e3, (d.get('steady_state') or {}).get('ms_per_step'))
PY

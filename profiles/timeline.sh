#!/bin/bash
# Per-workgroup phase timeline of the instance engine (run on the GPU box through gpurun):
#   bash profiles/timeline.sh [B ...]        -> gpurun_out/timeline.txt
set -eu
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/build/abl $REPO/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $REPO/include -DLA3D_TIMELINE \
  $REPO/labelany3d_amd/csrc/la3d.hip $REPO/labelany3d_amd/csrc/la3d_instance.hip $REPO/labelany3d_amd/csrc/la3d_band.hip $REPO/labelany3d_amd/csrc/la3d_rows.hip $REPO/labelany3d_amd/csrc/la3d_split.hip $REPO/labelany3d_amd/csrc/la3d_points.hip $REPO/labelany3d_amd/csrc/la3d_masks.hip $REPO/labelany3d_amd/csrc/la3d_consumers.hip $REPO/labelany3d_amd/csrc/la3d_json.cpp -o $REPO/build/abl/libla3d_timeline.so
TL_DETAIL=1 LA3D_LIB=$REPO/build/abl/libla3d_timeline.so python $REPO/profiles/timeline.py "$@" 2>&1 | grep -v amdgpu.ids | tee $REPO/gpurun_out/timeline.txt

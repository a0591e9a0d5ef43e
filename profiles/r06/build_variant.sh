#!/bin/bash
# build_variant.sh NAME [-DFLAG ...]: an experiment build of the library into build/abl/libla3d_NAME.so (loaded with LA3D_LIB=...)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
mkdir -p $REPO/build/abl
C=$REPO/labelany3d_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $REPO/include "$@" \
  $C/la3d.hip $C/la3d_instance.hip $C/la3d_band.hip $C/la3d_rows.hip $C/la3d_split.hip $C/la3d_points.hip $C/la3d_masks.hip $C/la3d_consumers.hip $C/la3d_json.cpp \
  -o $REPO/build/abl/libla3d_$NAME.so && echo "built build/abl/libla3d_$NAME.so"

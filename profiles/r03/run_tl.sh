mkdir -p gpurun_out/r03tl build/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -DLA3D_TIMELINE labelany3d_amd/csrc/la3d.hip labelany3d_amd/csrc/la3d_split.hip -o build/abl/libla3d_timeline.so
TL_RLE=1 TL_DETAIL=1 LA3D_LIB=build/abl/libla3d_timeline.so python profiles/timeline.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r03tl/tl_rle.txt
TL_DETAIL=1 LA3D_RETAIN=0 LA3D_LIB=build/abl/libla3d_timeline.so python profiles/timeline.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r03tl/tl_plain.txt
TL_DETAIL=1 LA3D_LIB=build/abl/libla3d_timeline.so python profiles/timeline.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r03tl/tl_ret.txt
head -70 gpurun_out/r03tl/tl_rle.txt

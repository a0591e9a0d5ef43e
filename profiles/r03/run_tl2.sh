mkdir -p gpurun_out/r03tl build/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -DLA3D_TIMELINE labelany3d_amd/csrc/la3d.hip labelany3d_amd/csrc/la3d_split.hip -o build/abl/libla3d_timeline.so
for B in 1 8 64; do
echo "=== RLE plain B=$B"; TL_RLE=1 LA3D_LIB=build/abl/libla3d_timeline.so python profiles/timeline.py $B 2>&1 | grep -v amdgpu.ids | grep -A9 "sizes=bench"
echo "=== RLE retaining B=$B"; LA3D_RETAIN_NOMASK=1 TL_RLE=1 LA3D_LIB=build/abl/libla3d_timeline.so python profiles/timeline.py $B 2>&1 | grep -v amdgpu.ids | grep -A9 "sizes=bench"
echo "=== u8 retaining B=$B"; LA3D_LIB=build/abl/libla3d_timeline.so python profiles/timeline.py $B 2>&1 | grep -v amdgpu.ids | grep -A9 "sizes=bench"
done > gpurun_out/r03tl/tl_small.txt
cat gpurun_out/r03tl/tl_small.txt

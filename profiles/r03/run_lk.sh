mkdir -p gpurun_out/r03lk
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r03lk/tests.txt
L=labelany3d_amd/lib/libla3d.so; N=build/abl/libla3d_nolk.so
timeout 900 python profiles/sweep_variants.py nolk_auto=$N lk_auto=$L nolk_plain=$N,LA3D_RETAIN=0 lk_plain=$L,LA3D_RETAIN=0 --batches 512,1024,2048,8192 --rle --poly --config3 800 > gpurun_out/r03lk/sweep.txt 2>&1
timeout 600 python profiles/sweep_variants.py nolk_plain=$N,LA3D_RETAIN=0 lk_plain=$L,LA3D_RETAIN=0 lk_auto=$L --batches 1024,16384 --config5 > gpurun_out/r03lk/sweep_c5.txt 2>&1
cat gpurun_out/r03lk/tests.txt; cat gpurun_out/r03lk/sweep.txt | tail -30; tail -12 gpurun_out/r03lk/sweep_c5.txt

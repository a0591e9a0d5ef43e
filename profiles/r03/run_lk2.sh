mkdir -p gpurun_out/r03lk
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r03lk/tests2.txt
python profiles/r03/exp_unproject.py > gpurun_out/r03lk/unproject.txt 2>&1
LA3D_LIB=build/abl/libla3d_unt.so python profiles/r03/exp_unproject.py >> gpurun_out/r03lk/unproject.txt 2>&1
L=labelany3d_amd/lib/libla3d.so
timeout 600 python profiles/sweep_variants.py final=$L --batches 1024 --rle --poly > gpurun_out/r03lk/sweep2.txt 2>&1
cat gpurun_out/r03lk/tests2.txt gpurun_out/r03lk/unproject.txt; tail -3 gpurun_out/r03lk/sweep2.txt

mkdir -p gpurun_out/r03lk
for v in product A B C product; do
  if [ $v = product ]; then python profiles/r03/exp_unproject.py 2>&1 | grep unproject; else LA3D_LIB=build/abl/libla3d_u$v.so python profiles/r03/exp_unproject.py 2>&1 | grep unproject; fi
done > gpurun_out/r03lk/unproject4.txt
cat gpurun_out/r03lk/unproject4.txt

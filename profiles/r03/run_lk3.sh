mkdir -p gpurun_out/r03lk
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r03lk/tests3.txt
python profiles/r03/exp_unproject.py > gpurun_out/r03lk/unproject3.txt 2>&1
cat gpurun_out/r03lk/tests3.txt gpurun_out/r03lk/unproject3.txt

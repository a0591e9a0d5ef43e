mkdir -p gpurun_out/r03tl
for B in 512 1024 2048 4096 8192; do
  python bench.py --rle --batch $B --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('rle B=$B', round(d['ms_per_step']*1e3,1),'us', round(d['value']/1e6,2),'M/s', 'per1024', round(d['ms_per_step']*1e3*1024/$B,1))"
done > gpurun_out/r03tl/rle_B.txt
for B in 1024 4096; do
  python bench.py --rle --area-hint --batch $B --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('rle+hint B=$B', round(d['ms_per_step']*1e3,1),'us', round(d['value']/1e6,2),'M/s', 'per1024', round(d['ms_per_step']*1e3*1024/$B,1))"
done >> gpurun_out/r03tl/rle_B.txt
cat gpurun_out/r03tl/rle_B.txt

for i in 1 2 3; do
for v in "" "LA3D_RETAIN=0"; do
  env $v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('${v:-default}', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,1), 'us wall', round(d['roofline']['avg_launch_ms']*1e3,1), 'us events')"
done; done

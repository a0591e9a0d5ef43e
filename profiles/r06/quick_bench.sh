#!/bin/bash
# us per step (HIP events; rotating inputs | same batch) of the main bench modes - quick A/B between builds
cd "$(dirname "$0")/../.."
if [ -z "$MODES" ]; then MODES="|--rle|--poly|--ground|--config5|--batch 256|--batch 8192 --steps 60"; fi
IFS="|" read -ra MODE_LIST <<< "$MODES"
for mode in "${MODE_LIST[@]}"; do
  printf "%-28s" "bench ${mode:-config2}:"
  python bench.py $mode --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['rotation']
print('%8.2f us | same batch %8.2f | pipelined %8.2f | stream ceiling %.0f GB/s' % (d['roofline']['avg_launch_ms']*1e3, (r['same_batch_ms_per_step'] or 0)*1e3, d['pipelined']['ms_per_step']*1e3 if d.get('pipelined') else 0, d['roofline']['measured_stream_GBps']))"
done

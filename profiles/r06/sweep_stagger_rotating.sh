#!/bin/bash
# Stagger period of the resident groups (LA3D_STAGGER_US) re-swept under the round-6 protocol (rotating input batches): the period was
# tuned in rounds 4/5 on one resident batch, where the 256 MB Infinity Cache lifted the stream to ~6.4 TB/s.
cd "$(dirname "$0")/../.."
for us in ${PERIODS:-default 0 9 10.5 12.3 13.5 15 17}; do
  if [ "$us" = default ]; then unset LA3D_STAGGER_US; else export LA3D_STAGGER_US=$us; fi
  for mode in "" "--config5" "--ground"; do
    printf "stagger %-8s %-10s" "$us" "${mode:-config2}"
    python bench.py $mode --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['rotation']
print('%8.2f us rotating | same batch %8.2f | stream ceiling %.0f GB/s' % (d['roofline']['avg_launch_ms']*1e3, (r['same_batch_ms_per_step'] or 0)*1e3, d['roofline']['measured_stream_GBps']))"
  done
done

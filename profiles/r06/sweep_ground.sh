#!/bin/bash
# Round 6: cheap knobs on the GROUNDED config-2 call (two-pass form): pass-B culling threshold and the stagger period of the resident
# groups, both read once per process from the environment.  us per step (HIP events, 300 steps, rotating inputs) per setting.
cd "$(dirname "$0")/../.."
run() { env "$@" python bench.py --ground --steps 300 --warmup 30 --no-cpu-baseline --no-steady --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%7.2f us  (same batch %.2f)' % (d['roofline']['avg_launch_ms']*1e3, (d['rotation']['same_batch_ms_per_step'] or 0)*1e3))"; }
echo "default:                $(run X=1)"
for c in 48 64 96 160 224; do echo "LA3D_CULL_MIN_U8=$c:    $(run LA3D_CULL_MIN_U8=$c)"; done
for s in 6 8 10 14 16 18; do echo "LA3D_STAGGER_US=$s:      $(run LA3D_STAGGER_US=$s)"; done
echo "LA3D_BALANCE=0:         $(run LA3D_BALANCE=0)"
echo "ungrounded default:     $(env X=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-steady --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%7.2f us  (same batch %.2f)' % (d['roofline']['avg_launch_ms']*1e3, (d['rotation']['same_batch_ms_per_step'] or 0)*1e3))")"
for s in 8 10 14 16; do echo "ungrounded LA3D_STAGGER_US=$s: $(env LA3D_STAGGER_US=$s python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-steady --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%7.2f us  (same batch %.2f)' % (d['roofline']['avg_launch_ms']*1e3, (d['rotation']['same_batch_ms_per_step'] or 0)*1e3))")"; done

#!/bin/bash
# round 4: whole GPU suite after the dispatch change (band engine for 4 <= B <= 400 u8 planes), the new tools, first bench lines
O=gpurun_out/r04all1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/tests_all.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2>$O/bench_driver_style.err
timeout 300 python bench.py --no-cpu-baseline --config4 2000 > $O/bench_config4.json 2>$O/bench_config4.err
timeout 600 python bench.py --end-to-end 1024 > $O/bench_e2e.json 2>$O/bench_e2e.err
cat $O/tests_all.txt
for n in bench_driver_style bench_config4 bench_e2e; do echo "== $n"; tail -c 1500 $O/$n.json; tail -5 $O/$n.err; done

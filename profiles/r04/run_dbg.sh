#!/bin/bash
O=gpurun_out/r04dbg; mkdir -p $O
timeout 600 python profiles/r04/debug_cull_diff.py > $O/diff.txt 2>&1; tail -80 $O/diff.txt

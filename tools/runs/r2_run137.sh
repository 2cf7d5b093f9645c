#!/bin/bash
# crossover of the paired-rows attention form by context: tools/bench_engine_overlap.py, 54 streams per engine, 96 steps from ctx0
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
for c in 128 256 384 448 512 640 896; do for p in 0 1; do echo "ctx0 $c ETD_AD_PAIR=$p: $(ETD_AD_PAIR=$p timeout -k 10 200 python3 tools/bench_engine_overlap.py 54 $c 96 2>&1 | grep '^E=' | sed -n '1p;4p' | awk '{printf "%s %s ms  ", $1, $2}')"; done; done

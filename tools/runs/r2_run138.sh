#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
for rep in 1 2; do for p in 0 1; do echo "64 rows ctx0 512 ETD_AD_PAIR=$p: $(ETD_AD_PAIR=$p timeout -k 10 200 python3 tools/bench_engine_overlap.py 64 512 96 2>&1 | grep '^E=' | sed -n '1p;2p' | awk '{printf "%s %s ms  ", $1, $2}')"; done; done
for rep in 1 2; do for p in 0 1; do echo "64 rows ctx0 3400 ETD_AD_PAIR=$p: $(ETD_BENCH_MAXCTX=4096 ETD_AD_PAIR=$p timeout -k 10 300 python3 tools/bench_engine_overlap.py 64 3400 64 2>&1 | grep '^E=' | sed -n '1p;2p' | awk '{printf "%s %s ms  ", $1, $2}')"; done; done

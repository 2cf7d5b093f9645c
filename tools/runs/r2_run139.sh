#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
for m in 16 27 32 40 48 54 60 64 72 96 128; do for p in 0 1; do echo "rows $m ctx0 340 ETD_AD_PAIR=$p: $(ETD_AD_PAIR=$p timeout -k 10 200 python3 tools/bench_engine_overlap.py $m 340 64 2>&1 | grep '^E=' | sed -n '1p;4p' | awk '{printf "%s %s ms  ", $1, $2}')"; done; done

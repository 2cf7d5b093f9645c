#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r132; mkdir -p $O
run() { echo "== $1" | tee -a $O/b.txt; shift; timeout -k 10 400 env "$@" python3 bench.py --no-cpu-baseline --no-extras >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"\|"avg_launch_ms": [0-9.]*' $O/b.txt | tail -3 | tr '\n' ' '; echo; }
run "default" X=1
run "two rows per attention workgroup" ETD_AD_PAIR=1
run "default" X=1
run "two rows per attention workgroup" ETD_AD_PAIR=1
for p in 0 1; do echo "== bench_engine_overlap ETD_AD_PAIR=$p"; ETD_AD_PAIR=$p timeout -k 10 300 python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="; done

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r133; mkdir -p $O
ETD_AD_PAIR=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_pair.txt 2>&1; tail -3 $O/pytest_pair.txt
ETD_AD_PAIR=1 PROBE_REPS=2 PROBE_LINES=6 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_trace.py 100 extractor 2>&1 | grep "^rep\|alone\|   step" | head
run() { echo "== $1" | tee -a $O/b.txt; shift; timeout -k 10 400 env "$@" python3 bench.py --no-cpu-baseline --no-extras >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo; }
for i in 1 2 3; do run "default" X=1; run "pair" ETD_AD_PAIR=1; done

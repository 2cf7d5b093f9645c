#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r127; mkdir -p $O
run() { echo "== $1" | tee -a $O/b.txt; shift; timeout -k 10 400 env "$@" python3 bench.py --no-cpu-baseline --no-extras >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo; }
run "default (fused prefill MLP on)" X=1
run "row finish in the attention launch" ETD_ROWFIN=1
run "default" X=1
run "row finish" ETD_ROWFIN=1
run "stagger 2 ms" ETD_ENGINE_STAGGER_MS=2
run "stagger 6 ms" ETD_ENGINE_STAGGER_MS=6

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r61; mkdir -p $O
for i in 1 2 3; do timeout -k 10 200 python3 tools/probe_race.py 12 600 gemm 2>&1 | grep "^(" | sed "s/^/gemm $i /" | tee -a $O/race.txt; done
for i in 1 2 3; do timeout -k 10 200 python3 tools/probe_race.py 12 600 own0 2>&1 | grep "^(a" | sed "s/^/own0 $i /" | tee -a $O/race.txt; done

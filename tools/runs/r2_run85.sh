#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r85; mkdir -p $O
for k in 0 1 2 3; do for i in 1 2 3 4; do PROBE_KERNEL=$k timeout -k 10 400 python3 tools/probe_race.py 12 600 gemm 2>&1 | grep "^(a" | sed "s/^/aggressor kernel $k: run $i /" | tee -a $O/race.txt; done; done

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r97; mkdir -p $O
for cfg in "1,8,8,4" "3,8,8,4" "3,256,1,1" "3,2048,1,1"; do for i in 1 2; do PROBE_EMPTY=$cfg timeout -k 10 400 python3 tools/probe_race.py 12 600 emptykernel 2>&1 | grep "^(a" | sed "s/^/empty kernel (which,grid)=$cfg: run $i /" | tee -a $O/race.txt; done; done

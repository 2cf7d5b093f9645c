#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r94; mkdir -p $O
for m in events alloc emptykernel; do for i in 1 2 3; do timeout -k 10 400 python3 tools/probe_race.py 12 600 $m 2>&1 | grep "^(a" | sed "s/^/aggressor = $m: run $i /" | tee -a $O/race.txt; done; done

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r103; mkdir -p $O
for i in 1 2 3; do PROBE_EMPTY_INLIB=-1,8,4 timeout -k 10 400 python3 tools/probe_race.py 12 600 emptykernel 2>&1 | grep "^(a" | sed "s/^/ext_kernels.hip's empty kernel through etd_debug_empty_launch: run $i /" | tee -a $O/race.txt; done

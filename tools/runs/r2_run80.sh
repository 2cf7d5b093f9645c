#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r80; mkdir -p $O
for i in 1 2 3 4 5 6; do PROBE_PREC=fp32 timeout -k 10 400 python3 tools/probe_race.py 6 200 2>&1 | grep "^(a" | sed "s/^/fp32: run $i /" | tee -a $O/race.txt; done

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r71; mkdir -p $O
for i in 1 2 3 4; do timeout -k 10 200 python3 tools/probe_race.py 12 600 regnoise 2>&1 | grep "^(a" | sed "s/^/register noise: run $i /" | tee -a $O/race.txt; done

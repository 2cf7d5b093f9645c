#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r60; mkdir -p $O
for i in 1 2 3; do timeout -k 10 200 python3 tools/probe_race.py 12 600 hog 2>&1 | grep "^(" | sed "s/^/hog $i /" | tee -a $O/race.txt; done

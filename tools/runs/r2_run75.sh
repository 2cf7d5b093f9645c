#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r75; mkdir -p $O
timeout -k 10 200 python3 tools/probe_race.py 12 600 hog 2>&1 | grep "^(m) engine 2" | sed "s/^/hog  /" | tee -a $O/race.txt
for i in 1 2 3; do timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(m) engine 2\|^(a" | sed "s/^/run$i /" | tee -a $O/race.txt; done

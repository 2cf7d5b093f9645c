#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r68; mkdir -p $O
for i in 1 2 3 4; do PROBE_HOG_ON_ENGINE0=1 timeout -k 10 200 python3 tools/probe_race.py 12 600 hog 2>&1 | grep "^(a" | sed "s/^/hog on engine 0's stream: run $i /" | tee -a $O/race.txt; done
for i in 1 2 3 4; do PROBE_PREFILL_NEW_STREAM=1 timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/prefill on a fresh stream: run $i /" | tee -a $O/race.txt; done
unset GPU_MAX_HW_QUEUES
for i in 1 2 3 4; do timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/default queues: run $i /" | tee -a $O/race.txt; done

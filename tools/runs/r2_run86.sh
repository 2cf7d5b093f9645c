#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r86; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout -k 10 400 python3 tools/probe_race.py 12 600 burn 2>&1 | grep "^(a" | sed "s/^/aggressor = MFMA burner: run $i /" | tee -a $O/race.txt; done

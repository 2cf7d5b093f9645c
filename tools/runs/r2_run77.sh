#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r77; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do ETD_ROWFIN=1 timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/rowfin: run $i /" | tee -a $O/race.txt; done

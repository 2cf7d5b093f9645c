#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r105; mkdir -p $O
for g in 8,2,4 64,1,1 8,8,4 1,1,1 16,1,1 1024,1,1; do
PROBE_EXT_IDLE=2 PROBE_EXT_IDLE_GRID=$g timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/prof.hip empty kernel, grid $g: /" | tee -a $O/race.txt
done

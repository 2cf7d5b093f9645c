#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r100; mkdir -p $O
for i in 1 2 3; do PROBE_EXT_IDLE=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/extractor created and run once, then idle: run $i /" | tee -a $O/race.txt; done

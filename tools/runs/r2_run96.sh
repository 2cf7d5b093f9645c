#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r96; mkdir -p $O
for i in 1 2 3; do timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a\|^(x" | sed "s/^/extractor, steady state only: run $i /" | tee -a $O/race.txt; done
for i in 1 2; do ETD_EXT_STOP_STAGE=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a\|^(x" | sed "s/^/k_embed only, steady state only: run $i /" | tee -a $O/race.txt; done
for i in 1 2; do PROBE_EXT_EARLY=1 ETD_EXT_STOP_STAGE=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a\|^(x" | sed "s/^/k_embed only, steppers start 3 s into the creation: run $i /" | tee -a $O/race.txt; done

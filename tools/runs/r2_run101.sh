#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
export PROBE_EXT_CAPI=1
O=gpurun_out/r101; mkdir -p $O
for w in 1 2 3; do for i in 1 2; do ETD_EXT_DBG_EMPTY=$w timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/empty kernel $w in the k_embed translation unit: run $i /" | tee -a $O/race.txt; done; done

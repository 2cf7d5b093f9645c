#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r72; mkdir -p $O
for i in 1 2 3 4 5 6; do ETD_NO_GRAPH=1 timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/eager steps: run $i /" | tee -a $O/race.txt; done
for i in 1 2 3 4 5 6; do ETD_NO_ATTN_DOWN=1 ETD_NO_FUSED_STEP=1 timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/plain step kernels: run $i /" | tee -a $O/race.txt; done

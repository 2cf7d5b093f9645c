#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
export ETD_EXT_STOP_STAGE=1
O=gpurun_out/r98; mkdir -p $O
for i in 1 2; do PROBE_EXT_CAPI=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/k_embed through the C API, sync per launch: run $i /" | tee -a $O/race.txt; done
for i in 1 2; do PROBE_EXT_CAPI=1 PROBE_EXT_NOSYNC=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/k_embed through the C API, no sync: run $i /" | tee -a $O/race.txt; done

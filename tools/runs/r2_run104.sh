#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r104; mkdir -p $O
for i in 1 2; do PROBE_EXT_IDLE=2 PROBE_EXT_IDLE_GX=-1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/extractor mode, k_dbg_e1 through etd_debug_empty_launch: run $i /" | tee -a $O/race.txt; done
for i in 1 2; do PROBE_EXT_CAPI=1 ETD_EXT_DBG_EMPTY=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/extractor mode, k_dbg_e1 through etd_transcript_windows: run $i /" | tee -a $O/race.txt; done

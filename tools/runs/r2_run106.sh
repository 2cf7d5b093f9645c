#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r106; mkdir -p $O
echo "== k_dbg_e1 through etd_debug_empty_launch" >> $O/race.txt
PROBE_EXT_IDLE=2 PROBE_EXT_IDLE_GRID=-1,8,4 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor >> $O/race.txt 2>&1
echo "== k_dbg_e1 through etd_transcript_windows" >> $O/race.txt
PROBE_EXT_CAPI=1 ETD_EXT_DBG_EMPTY=1 timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor >> $O/race.txt 2>&1
grep "^==\|^(a\|^(x\|Error\|error" $O/race.txt

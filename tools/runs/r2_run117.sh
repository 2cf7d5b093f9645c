#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r117; mkdir -p $O
PROBE_REPS=4 PROBE_LINES=0 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 400 python3 tools/probe_trace.py 1 extractor > $O/lanes.txt 2>&1
grep -v "ratio b/a\|host model" $O/lanes.txt | tail -70 | cut -c1-260

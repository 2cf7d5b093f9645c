#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r118; mkdir -p $O
PROBE_REPS=4 PROBE_LINES=6 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 400 python3 tools/probe_trace.py 100 extractor > $O/t.txt 2>&1
grep "^rep\|alone\|step" $O/t.txt | tail -30 | cut -c1-200

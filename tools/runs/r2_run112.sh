#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r112; mkdir -p $O
PROBE_REPS=10 PROBE_LINES=0 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 500 python3 tools/probe_trace.py 1 extractor > $O/trace_kv6.txt 2>&1
grep "lost slots\|by j\|exactly" $O/trace_kv6.txt | tail -60

#!/bin/bash
# library built with ETD_EXTRA_FLAGS=-DETD_AD_XCHG=1 (attention core exchanges through DPP / permlane swaps instead of ds_bpermute)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r111; mkdir -p $O
PROBE_REPS=3 PROBE_LINES=14 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 500 python3 tools/probe_trace.py 100 extractor > $O/trace_kv6_xchg1.txt 2>&1
grep -v "layer [1-7]" $O/trace_kv6_xchg1.txt | tail -40

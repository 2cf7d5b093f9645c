#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r108; mkdir -p $O
ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 500 python3 tools/probe_trace.py 300 extractor > $O/trace_kv6.txt 2>&1
tail -40 $O/trace_kv6.txt

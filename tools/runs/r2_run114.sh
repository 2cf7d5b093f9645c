#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r114; mkdir -p $O
ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_xlane.py 3 15 432 40 > $O/xlane.txt 2>&1
cat $O/xlane.txt | tail -12

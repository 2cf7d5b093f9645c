#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r124; mkdir -p $O
ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_merge.py 4 432 20 > $O/merge.txt 2>&1
tail -14 $O/merge.txt

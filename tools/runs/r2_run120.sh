#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r120; mkdir -p $O
ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_pk.py 3 432 40 > $O/pk.txt 2>&1
tail -14 $O/pk.txt

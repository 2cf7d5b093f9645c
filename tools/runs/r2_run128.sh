#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r128; mkdir -p $O
for m in 1 2 3; do PROBE_PK_ASYNC=$m ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_pk.py 3 432 40 >> $O/pk.txt 2>&1; done
grep -v amdgpu.ids $O/pk.txt | cut -c1-250

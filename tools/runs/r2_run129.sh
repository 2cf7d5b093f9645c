#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r129; mkdir -p $O
for m in 16 17 18 19 20 21 22 23; do echo "== mode $m" >> $O/pk.txt; PROBE_PK_ASYNC=$m ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_pk.py 2 432 40 >> $O/pk.txt 2>&1; done
grep "^==\|beside the aggressor:\|lane group" $O/pk.txt | cut -c1-250

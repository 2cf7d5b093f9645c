#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r82; mkdir -p $O
for i in 1 2 3 4 5 6; do PROBE_LOGITS_T=1088 timeout -k 10 400 python3 tools/probe_race.py 150 600 2>&1 | grep "^(a" | sed "s/^/aggressor = prefill_logits(T=1088) x150: run $i /" | tee -a $O/race.txt; done

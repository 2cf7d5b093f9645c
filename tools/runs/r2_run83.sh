#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r83; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/aggressor = extractor: run $i /" | tee -a $O/race.txt; done
for i in 1 2 3 4 5 6; do ETD_NO_MFMA_PREFILL_ATTN=1 PROBE_LOGITS_T=1088 timeout -k 10 400 python3 tools/probe_race.py 150 600 2>&1 | grep "^(a" | sed "s/^/aggressor = prefill_logits without k_attn: run $i /" | tee -a $O/race.txt; done

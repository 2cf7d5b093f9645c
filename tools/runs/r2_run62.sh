#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r62; mkdir -p $O
for np_ in 1 2 8; do for i in 1 2 3; do PROBE_NPROMPTS=$np_ timeout -k 10 200 python3 tools/probe_race.py 200 600 2>&1 | grep "^(a" | sed "s/^/prompts=$np_ run $i /" | tee -a $O/race.txt; done; done
for i in 1 2 3; do ETD_NO_LAST_ONLY=1 ETD_NO_MFMA_PREFILL_ATTN=1 timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/54 prompts, no last-only, no mfma attn: run $i /" | tee -a $O/race.txt; done

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r81; mkdir -p $O
for i in 1 2 3 4 5 6; do PROBE_PREC=fp32 PROBE_PREC0=bf16 timeout -k 10 400 python3 tools/probe_race.py 12 200 own0 2>&1 | grep "^(m) engine 2" | cut -c1-100 | sed "s/^/fp32 steps beside bf16 prefill: run $i /" | tee -a $O/race.txt; done
for i in 1 2 3 4 5 6; do PROBE_PREC=bf16 PROBE_PREC0=fp32 timeout -k 10 400 python3 tools/probe_race.py 6 600 own0 2>&1 | grep "^(a" | sed "s/^/bf16 steps beside fp32 prefill: run $i /" | tee -a $O/race.txt; done

#!/bin/bash
# kernarg preload off: do the stepping engines still flip beside a prefill?
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r69; mkdir -p $O
ETD_KERNARG_PRELOAD=0 python3 -m etude_amd.build --force > /dev/null 2>&1
for i in 1 2 3 4 5 6; do timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/no kernarg preload: run $i /" | tee -a $O/race.txt; done
python3 -m etude_amd.build --force > /dev/null 2>&1

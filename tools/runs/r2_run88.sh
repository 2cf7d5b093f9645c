#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r88; mkdir -p $O
for kib in 32 60 68 82 100; do for i in 1 2 3; do timeout -k 10 400 python3 tools/probe_race.py 12 600 lds$kib 2>&1 | grep "^(a" | sed "s/^/aggressor = ${kib} KiB LDS fill: run $i /" | tee -a $O/race.txt; done; done

#!/bin/bash
# the prefills in ONE process, the stepping engines in ANOTHER: does the interference cross a process boundary?
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r79; mkdir -p $O
for i in 1 2 3 4 5 6; do
  PROBE_ROLE=prefiller timeout -k 10 200 python3 tools/probe_race.py 4000 600 > $O/p.txt 2>&1 &
  sleep 2
  PROBE_ROLE=stepper timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/two processes: run $i /" | tee -a $O/race.txt
  wait; tail -1 $O/p.txt
done

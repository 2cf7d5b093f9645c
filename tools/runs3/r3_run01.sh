#!/bin/bash
# r3_run01: the headline at north_star's batch (64 clips, 1728 decode jobs) with the round-2 tree: engines 4 / 2 / 3, 432..864 rows per launch
set -x
mkdir -p gpurun_out/r3_01
for e in 4 2 3; do
  ETD_SCHED_STATS=1 timeout -k 10 400 python bench.py --clips 64 --streams 1728 --engines $e --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r3_01/e$e.json 2> gpurun_out/r3_01/e$e.err || exit 1
  tail -c 3000 gpurun_out/r3_01/e$e.json
done

#!/bin/bash
# r3_run03: where the decode step stands at the 64-clip batch's launch sizes: rows x ctx sweep, one engine and four, pair / one-row / unfused attention
set -x
mkdir -p gpurun_out/r3_03
O=gpurun_out/r3_03/sweep.jsonl
: > $O
for rc in "432 340" "432 537" "512 537" "216 537" "128 537"; do set -- $rc
  timeout -k 5 120 python tools/bench_step.py --rows $1 --ctx $2 >> $O 2>> gpurun_out/r3_03/err.log || exit 1
done
timeout -k 5 120 python tools/bench_step.py --rows 432 --ctx 537 --pair 0 >> $O 2>> gpurun_out/r3_03/err.log || exit 1
ETD_NO_ATTN_DOWN=1 timeout -k 5 120 python tools/bench_step.py --rows 432 --ctx 537 >> $O 2>> gpurun_out/r3_03/err.log || exit 1
ETD_AD_WAVES=8 timeout -k 5 120 python tools/bench_step.py --rows 432 --ctx 537 >> $O 2>> gpurun_out/r3_03/err.log || exit 1
for e in 2 4; do
  timeout -k 5 180 python tools/bench_step.py --rows 432 --ctx 537 --engines $e >> $O 2>> gpurun_out/r3_03/err.log || exit 1
done
cat $O

#!/bin/bash
# r3_run06: the whole GPU suite with the round-3 tests (long-context / step logits, paired-rows identity, context goldens, pipeline chain, bench smoke),
# then the low-overhead device stamps against the graph-derived attention time (bench_step: one engine 432 / 1728 rows, four engines)
set -x
mkdir -p gpurun_out/r3_06
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_06/pytest.log 2>&1; rc=$?
tail -25 gpurun_out/r3_06/pytest.log
[ $rc -eq 0 ] || exit $rc
O=gpurun_out/r3_06/stamps.jsonl; : > $O
timeout -k 5 200 python tools/bench_step.py --rows 432 --ctx 537 >> $O 2>> gpurun_out/r3_06/err.log || exit 1
timeout -k 5 200 python tools/bench_step.py --rows 1728 --ctx 537 >> $O 2>> gpurun_out/r3_06/err.log || exit 1
timeout -k 5 200 python tools/bench_step.py --rows 432 --ctx 537 --engines 4 >> $O 2>> gpurun_out/r3_06/err.log || exit 1
cat $O

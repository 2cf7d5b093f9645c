#!/bin/bash
# does the stride between (slot, head) regions of the KV caches (= pages touched per launch) set how engines overlap?
export TMPDIR=/tmp
O=gpurun_out/r15; mkdir -p $O
for mc in 256 1024 4160; do
  echo "== max_ctx $mc ctx0 16" >> $O/log.txt
  ETD_BENCH_MAXCTX=$mc timeout 300 python3 tools/bench_engine_overlap.py 54 16 96 >> $O/log.txt 2>&1
done
for mc in 512 1024 4160; do
  echo "== max_ctx $mc ctx0 320" >> $O/log.txt
  ETD_BENCH_MAXCTX=$mc timeout 300 python3 tools/bench_engine_overlap.py 54 320 96 >> $O/log.txt 2>&1
done
cat $O/log.txt

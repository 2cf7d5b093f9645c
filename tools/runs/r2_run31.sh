#!/bin/bash
# configs[3] (128 streams): one engine x 128 vs two x 64 vs four x 32, ctx 512 and 3.5 k
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r31; mkdir -p $O
for ctx in 512 3500; do
  for S in 128 64 32; do
    echo "== ctx $ctx, $S streams per engine" >> $O/log.txt
    ETD_BENCH_MAXCTX=4160 timeout 600 python3 tools/bench_engine_overlap.py $S $ctx 64 2>&1 | grep "^E=" >> $O/log.txt
  done
done
cat $O/log.txt

#!/bin/bash
# r3_run08: rocprofv3 kernel trace of the one-engine job restricted to 12 bars per job (8 of them at steady-state prompt size): per-kernel averages of the
# full-size batched prefill (886 k rows) and of the 1728-row decode step
set -x
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3_08; mkdir -p $OUT
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --engines 1 --max-streams 1728 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --max-bars 12 > $OUT/line.json 2> $OUT/err.log || { tail -20 $OUT/err.log; exit 1; }
cd $ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +20M -delete

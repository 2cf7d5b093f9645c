#!/bin/bash
# kernel durations vs inter-kernel gaps at 1..4 concurrent engines (steps only)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r16; rm -rf $O; mkdir -p $O
python3 tools/bench_engine_overlap.py 54 320 96 > $O/plain.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/bench_engine_overlap.py 54 320 96 > $O/traced.txt 2>&1
cd $R
python3 tools/trace_concurrency.py $O/tr > $O/concurrency.txt 2>&1
cat $O/plain.txt $O/traced.txt $O/concurrency.txt
find $O -name "*.csv" -size +1M -delete

#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace summary of the bench + PMC passes for HBM traffic.
# Results land in gpurun_out/prof_*; tools/summarize_profile.py turns them into the small files kept in profiles/.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --clips ${CLIPS:-8} --bars ${BARS:-92}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 $ROOT/bench.py $ARGS > $OUT/prof_trace.json 2> $OUT/prof_trace.err
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --clips ${CLIPS:-8} --attr-grid 27 --bars ${PMC_BARS:-24}"   # same rows per engine and (after 4 bars) the same contexts as the full run
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $ROOT/bench.py $PARGS > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $ROOT/bench.py $PARGS > $OUT/prof_write.json 2> $OUT/prof_write.err
cd $ROOT
python3 tools/summarize_profile.py $OUT > $OUT/profile_summary.txt 2>&1
python3 tools/trace_overlap.py $OUT/prof_trace >> $OUT/profile_summary.txt 2>&1
tail -40 $OUT/profile_summary.txt
du -sh $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write 2>/dev/null
find $OUT/prof_trace -name "*.csv" | head
# the raw per-dispatch traces are large: keep only the stats + the summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete

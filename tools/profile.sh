#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace summary of the bench command + two PMC passes for HBM traffic.
# Results land in gpurun_out/prof_*; tools/summarize_profile.py turns them into the small files kept in profiles/.
#   trace:  python3 bench.py --steps 1 --warmup 1 (the headline's configuration: 64 clips, one engine x 1728 streams; the line of THIS run is kept next to the stats)
#   PMC:    the same batch restricted to the first 8 bars of every job (counter mode serialises dispatches; rows per launch and, from bar 4 on, contexts as in the full run)
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 $ROOT/bench.py $ARGS > $OUT/prof_trace.json 2> $OUT/prof_trace.err || { tail -20 $OUT/prof_trace.err; exit 1; }
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-stamp --no-serial-pass --max-bars ${PMC_BARS:-8} ${BENCH_ARGS:-}"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $ROOT/bench.py $PARGS > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err || { tail -20 $OUT/prof_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $ROOT/bench.py $PARGS > $OUT/prof_write.json 2> $OUT/prof_write.err || { tail -20 $OUT/prof_write.err; exit 1; }
cd $ROOT
python3 tools/summarize_profile.py $OUT > $OUT/profile_summary.txt 2>&1
tail -45 $OUT/profile_summary.txt
cp $(find $OUT/prof_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
du -sh $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write 2>/dev/null
# the raw per-dispatch traces are large: keep only the stats + the summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete

#!/bin/bash
# r3_run14: (a) PMC passes without the serial event pass (the counters then cover exactly the timed step's launches), (b) k_dstep_qkv_up with K split over 4 waves at 1728 / 432 rows,
# (c) 3 / 4 extractor instances on the 64-clip extract stage
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT/r3_14
rm -rf $OUT/prof_fetch $OUT/prof_write
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-stamp --no-serial-pass --max-bars 8"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $ROOT/bench.py $PARGS > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err || { tail -20 $OUT/prof_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $ROOT/bench.py $PARGS > $OUT/prof_write.json 2> $OUT/prof_write.err || { tail -20 $OUT/prof_write.err; exit 1; }
cd $ROOT
python3 tools/summarize_profile.py $OUT > $OUT/r3_14/pmc_summary.txt 2>&1
grep -A12 "HBM traffic per launch" $OUT/r3_14/pmc_summary.txt | cut -c1-120; tail -1 $OUT/r3_14/pmc_summary.txt
O=$OUT/r3_14/w4.jsonl; : > $O
for lib in libetude_hip lib_w4; do for r in 1728 432; do
  echo "{\"lib\": \"$lib\"}" >> $O
  ETD_ALLOW_STALE_LIB=1 ETD_LIB_PATH=etude_amd/$lib.so timeout -k 5 200 python tools/bench_step.py --rows $r --ctx 537 >> $O 2>> $OUT/r3_14/err.log || { tail -5 $OUT/r3_14/err.log; exit 1; }
done; done
python - <<'P'
import json
lib=None
for l in open('gpurun_out/r3_14/w4.jsonl'):
    d=json.loads(l)
    if len(d)==1: lib=d['lib']; continue
    print(f"{lib:14s} rows {d['rows']:5d}: {d['ms_per_step']:.4f} ms/step qkv_up {d['event_us_per_launch'].get('k_dstep_qkv_up')} us attn {d.get('attn_us')}")
P
for x in 2 3 4; do
  timeout -k 10 200 python bench.py --ext-engines $x --steps 1 --warmup 1 --max-bars 2 --no-extras --no-cpu-baseline --no-stamp --no-serial-pass > $OUT/r3_14/ext$x.json 2> $OUT/r3_14/ext$x.err || { tail -5 $OUT/r3_14/ext$x.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r3_14/ext$x.json'))
print('ext engines $x: extract', d['extract_audio_s_per_s'], 'audio-s/s')"
done

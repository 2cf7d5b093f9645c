#!/bin/bash
# r3_run04: the fused decode step beyond 512 rows (DS_STEP_MAX_ROWS 2048): one engine with 864 / 1728 rows per launch vs four with 432; then the job with 1 / 2 engines
set -x
mkdir -p gpurun_out/r3_04
O=gpurun_out/r3_04/sweep.jsonl
: > $O
for r in 864 1296 1728; do
  timeout -k 5 200 python tools/bench_step.py --rows $r --ctx 537 >> $O 2>> gpurun_out/r3_04/err.log || { tail -5 gpurun_out/r3_04/err.log; exit 1; }
done
timeout -k 5 200 python tools/bench_step.py --rows 864 --ctx 537 --engines 2 >> $O 2>> gpurun_out/r3_04/err.log || exit 1
cat $O
for e in "1 1728" "2 864"; do set -- $e
  timeout -k 10 300 python bench.py --engines $1 --max-streams $2 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-stamp > gpurun_out/r3_04/job_e$1.json 2> gpurun_out/r3_04/job_e$1.err || { tail -20 gpurun_out/r3_04/job_e$1.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r3_04/job_e$1.json'))
print('engines $1', d['value'], d['ms_per_step'], d['decoder_tokens_per_s'], d['roofline'].get('decode_stage',{}).get('frac'), d['tokens_sha256_rank0'])
print(d['kernel_ms_serial_pass'])
"
done

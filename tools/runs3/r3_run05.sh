#!/bin/bash
# r3_run05: the 64-clip job with 1 engine x 1728 streams and 2 x 864 (fused step up to 2048 rows) against 4 x 432
set -x
mkdir -p gpurun_out/r3_05
for e in "1 1728" "2 864"; do set -- $e
  timeout -k 10 300 python bench.py --engines $1 --max-streams $2 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-stamp > gpurun_out/r3_05/job_e$1.json 2> gpurun_out/r3_05/job_e$1.err || { tail -20 gpurun_out/r3_05/job_e$1.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r3_05/job_e$1.json'))
print('engines $1', d['value'], d['ms_per_step'], d['decoder_tokens_per_s'], d['roofline'].get('decode_stage',{}).get('frac'), d['tokens_sha256_rank0'])
print(d['kernel_ms_serial_pass'])
"
done

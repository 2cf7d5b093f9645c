#!/bin/bash
# r3_run07: one engine x 1728 streams: how long is the queue empty at bar boundaries (host-side prompt assembly, read-back, staging)?
set -x
mkdir -p gpurun_out/r3_07
ETD_SCHED_STATS=1 timeout -k 10 300 python bench.py --engines 1 --max-streams 1728 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r3_07/job_e1.json 2> gpurun_out/r3_07/job_e1.err || { tail -20 gpurun_out/r3_07/job_e1.err; exit 1; }
grep -h "sched\|bench " gpurun_out/r3_07/job_e1.err
python -c "
import json
d=json.load(open('gpurun_out/r3_07/job_e1.json'))
print(d['value'], d['ms_per_step'], d['roofline'])
"

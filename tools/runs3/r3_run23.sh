#!/bin/bash
# r3_run23: closing run of the final tree: the 64-clip job on four engines (token digest against one engine), tools/profile.sh, the driver's command
mkdir -p gpurun_out/r3_23
timeout -k 10 300 python bench.py --engines 4 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r3_23/e4.json 2> gpurun_out/r3_23/e4.err || { tail -5 gpurun_out/r3_23/e4.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r3_23/e4.json'))
print('four engines:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['decode_stage']['frac'], d['tokens_sha256_rank0'])"
bash tools/profile.sh > gpurun_out/r3_23/profile.log 2>&1 || { tail -20 gpurun_out/r3_23/profile.log; exit 1; }
tail -1 gpurun_out/r3_23/profile.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_23/bench.json 2> gpurun_out/r3_23/bench.err
echo rc=$?
grep "bench " gpurun_out/r3_23/bench.err | tail -3
python -c "
import json
d=json.load(open('gpurun_out/r3_23/bench.json'))
print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['decode_stage']['frac'], d['tokens_sha256_rank0'])"

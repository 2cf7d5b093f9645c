#!/bin/bash
# r3_run19: GPU suite + smoke() after the prefill-attention tiling change; then the one-engine job for its effect (2 steps)
mkdir -p gpurun_out/r3_19
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3_19/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r3_19/pytest.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r3_19/job.json 2> gpurun_out/r3_19/job.err || { tail -5 gpurun_out/r3_19/job.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r3_19/job.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['decode_stage']['frac'], d['tokens_sha256_rank0'])
print(d['kernel_ms_serial_pass']['ms'])"

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r32; mkdir -p $O
for c in 512 3500; do python3 tools/bench_decoder_streams.py $c 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['workload']); print(d['ms_per_step'], d['tokens_per_s'], d['roofline']['frac'], d['kernel_ms_per_step'])"; done

#!/bin/bash
# ONE engine, every second repetition with NaN-poisoning workgroups on a second stream: does any step kernel read LDS it never wrote?
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r55; mkdir -p $O
run() { echo "== $TAG" | tee -a $O/det.txt; timeout -k 10 300 python3 tools/probe_determinism.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt; }
TAG="E=1 54 jobs x 12 bars, reps 1 and 3 with noise" run 1 54 12 4 bf16 48 noise
TAG="E=1 fp32 32 jobs x 6 bars, noise" run 1 32 6 4 fp32 48 noise

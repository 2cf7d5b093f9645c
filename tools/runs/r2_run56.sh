#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r56; mkdir -p $O
run() { echo "== $TAG" | tee -a $O/det.txt; timeout -k 10 300 python3 tools/probe_determinism.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt; }
export GPU_MAX_HW_QUEUES=8
TAG="E=4 own weights per engine" PROBE_NOCLONE=1 run 4 216 12 3
TAG="E=4 shared weights" run 4 216 12 3
unset GPU_MAX_HW_QUEUES
TAG="E=4 shared weights, default hardware queues (4)" run 4 216 12 3
TAG="E=2 default hardware queues" run 2 216 12 3

#!/bin/bash
# which kernel makes the four-engine token streams differ run to run?  one switch at a time
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r53; mkdir -p $O
run() { echo "== $TAG" | tee -a $O/det.txt; timeout -k 10 300 python3 tools/probe_determinism.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt; }
TAG="E=4 default" run 4 216 24 3
TAG="E=2" run 2 216 24 3
TAG="E=4 ETD_NO_GRAPH" ETD_NO_GRAPH=1 run 4 216 24 3
TAG="E=4 ETD_NO_ATTN_DOWN" ETD_NO_ATTN_DOWN=1 run 4 216 24 3
TAG="E=4 ETD_NO_FUSED_STEP" ETD_NO_FUSED_STEP=1 run 4 216 12 3
TAG="E=4 ETD_NO_LAST_ONLY" ETD_NO_LAST_ONLY=1 run 4 216 24 3
TAG="E=4 ETD_NO_MFMA_PREFILL_ATTN" ETD_NO_MFMA_PREFILL_ATTN=1 run 4 216 12 3
TAG="E=4 fp32" run 4 64 8 3 fp32

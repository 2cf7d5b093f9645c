#!/bin/bash
# prefill-only bars (1 token per bar) vs step-heavy bars under four concurrent engines
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r54; mkdir -p $O
run() { echo "== $TAG" | tee -a $O/det.txt; timeout -k 10 300 python3 tools/probe_determinism.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt; }
TAG="E=4 prefill only (1 token per bar, 92 bars)" run 4 216 92 3 bf16 1
TAG="E=4 prefill only, ETD_NO_MFMA_PREFILL_ATTN" ETD_NO_MFMA_PREFILL_ATTN=1 run 4 216 40 3 bf16 1
TAG="E=4 prefill only, ETD_NO_LAST_ONLY" ETD_NO_LAST_ONLY=1 run 4 216 92 3 bf16 1
TAG="E=4 fp32, 216 jobs x 12 bars" run 4 216 12 3 fp32
TAG="E=4 bf16 tiny prompts: 2 bars x 200 tokens (steps dominate)" run 4 216 2 3 bf16 200

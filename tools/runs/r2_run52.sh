#!/bin/bash
# run-to-run reproducibility of the greedy token streams (the bench's digests differed between identical runs)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r52; mkdir -p $O
timeout -k 10 200 python3 tools/probe_determinism.py 1 54 24 3 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt
timeout -k 10 300 python3 tools/probe_determinism.py 4 216 24 3 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt
ETD_NO_GRAPH=1 timeout -k 10 300 python3 tools/probe_determinism.py 1 54 24 2 2>&1 | grep -v amdgpu.ids | sed 's/^/nograph: /' | tee -a $O/det.txt

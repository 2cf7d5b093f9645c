#!/bin/bash
# r3_run22: which switch does the sampled-run reproducibility failure follow?
mkdir -p gpurun_out/r3_22
for envs in "ETD_QKV_MT_MIN=513" "ETD_QKV_MT=0" "ETD_NO_GRAPH=1" "X=1"; do for i in 1 2; do
  env $envs timeout -k 5 120 python -m pytest tests/test_gpu_sampling.py -m gpu -q -k "degenerate" 2>&1 | tail -1 | sed "s/^/$envs run $i: /"
done; done
git log --oneline | head -3

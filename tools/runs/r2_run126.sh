#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r126; mkdir -p $O
for e in 3 4 5 6 8; do
  echo "== engines $e" | tee -a $O/b.txt; timeout -k 10 400 env ETD_FUSED_PMLP=1 python3 bench.py --no-cpu-baseline --no-extras --engines $e >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo
done
export GPU_MAX_HW_QUEUES=16
for e in 6 8; do
  echo "== engines $e, GPU_MAX_HW_QUEUES=16" | tee -a $O/b.txt; timeout -k 10 400 env ETD_FUSED_PMLP=1 python3 bench.py --no-cpu-baseline --no-extras --engines $e >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo
done

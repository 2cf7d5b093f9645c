#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r131; mkdir -p $O
for i in 1 2 3; do echo "== --pipeline run $i" | tee -a $O/b.txt; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --pipeline >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo; done

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r125; mkdir -p $O
run() { echo "== $1" | tee -a $O/b.txt; shift; timeout -k 10 400 env "$@" python3 bench.py --no-cpu-baseline --no-extras >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo; }
run "default" X=1
run "fused prefill MLP" ETD_FUSED_PMLP=1
run "default again" X=1
run "fused prefill MLP again" ETD_FUSED_PMLP=1
echo "== --pipeline" | tee -a $O/b.txt; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --pipeline >> $O/b.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/b.txt | tail -2 | tr '\n' ' '; echo

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export ETD_ROWFIN=1   # the in-launch row finish is opt-in
export TMPDIR=/tmp
O=gpurun_out/r22; mkdir -p $O
timeout 600 python3 tools/step_stamps.py 54 320 48 > $O/stamps_fin.txt 2>&1
cat $O/stamps_fin.txt

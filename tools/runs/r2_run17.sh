#!/bin/bash
# phase stamps of the decode-step kernels, 1 vs 4 engines, ctx 320 and ctx 16 (needs the -DETD_STEP_STAMP build shipped with the tree)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r17; mkdir -p $O
ETD_STAMP_OUT=$O/stamps320.npz timeout 600 python3 tools/step_stamps.py 54 320 48 > $O/stamps320.txt 2>&1
ETD_STAMP_OUT=$O/stamps16.npz timeout 600 python3 tools/step_stamps.py 54 16 48 > $O/stamps16.txt 2>&1
cat $O/stamps320.txt $O/stamps16.txt

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r92; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_reproducibility.py -x -q -s > $O/tests.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/tests.txt | tail -8

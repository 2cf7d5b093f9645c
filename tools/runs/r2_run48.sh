#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r48; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_decoder.py -x -q > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; echo "tests rc=$rc"

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r19; mkdir -p $O
python3 tools/bench_extractor.py 4 5 > $O/bench_ext.txt 2>&1
tail -15 $O/bench_ext.txt
python3 -m pytest tests/test_gpu_extractor.py -x -q -m gpu 2>&1 | tail -5

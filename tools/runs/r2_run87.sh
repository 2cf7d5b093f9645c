#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r87; mkdir -p $O
run() { for i in 1 2 3; do timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/$TAG: run $i /" | tee -a $O/race.txt; done; }
TAG="extractor: embedding only" ETD_EXT_STOP_STAGE=1 run
TAG="extractor: embedding + fused encoder layers" ETD_EXT_STOP_STAGE=2 run
TAG="extractor: embedding + round-1 encoder layers" ETD_EXT_STOP_STAGE=2 ETD_NO_FUSED_LAYER=1 ETD_NO_FUSED_FFN=1 ETD_NO_FUSED_PROJ=1 ETD_NO_FRAG_ATTN=1 run

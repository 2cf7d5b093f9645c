#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r70; mkdir -p $O
touch etude_amd/csrc/dec_kernels.hip; ETD_FLAGS_DEC_KERNELS="-DETD_KV_NT=0" python3 -m etude_amd.build > /dev/null 2>&1
for i in 1 2 3 4 5 6; do timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/no nt loads at all: run $i /" | tee -a $O/race.txt; done
touch etude_amd/csrc/dec_kernels.hip; python3 -m etude_amd.build > /dev/null 2>&1

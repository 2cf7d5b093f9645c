#!/bin/bash
# does an agent-scope fence behind the stores of every step kernel make the stepping engines reproducible beside a big prefill?
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r63; mkdir -p $O
touch etude_amd/csrc/dec_kernels.hip; ETD_FLAGS_DEC_KERNELS="-DETD_STEP_FENCE" python3 -m etude_amd.build > /dev/null 2>&1
for i in 1 2 3 4; do timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/fence run $i /" | tee -a $O/race.txt; done
touch etude_amd/csrc/dec_kernels.hip; python3 -m etude_amd.build > /dev/null 2>&1
for i in 1 2 3; do timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a" | sed "s/^/plain run $i /" | tee -a $O/race.txt; done

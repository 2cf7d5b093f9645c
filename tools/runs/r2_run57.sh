#!/bin/bash
# are the nontemporal K/V loads what makes concurrent engines irreproducible?  (-DETD_KV_NT=0: plain loads)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r57; mkdir -p $O
run() { echo "== $TAG" | tee -a $O/det.txt; timeout -k 10 300 python3 tools/probe_determinism.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/det.txt; }
touch etude_amd/csrc/dec_kernels.hip; ETD_FLAGS_DEC_KERNELS="-DETD_KV_NT=0" python3 -m etude_amd.build > /dev/null 2>&1
TAG="E=4 plain K/V loads" run 4 216 24 4
touch etude_amd/csrc/dec_kernels.hip; python3 -m etude_amd.build > /dev/null 2>&1
TAG="E=4 nontemporal K/V loads (default)" run 4 216 24 3

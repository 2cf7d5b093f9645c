#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r45; mkdir -p $O
timeout -k 10 400 python3 tools/probe_cu_mask.py 54 320 all,q_contig,half_contig 2>&1 | grep -v amdgpu.ids | tee $O/cu_mask.txt

#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
export ETD_EXT_STOP_STAGE=1
O=gpurun_out/r93; mkdir -p $O
for abl in 4 5; do
  touch etude_amd/csrc/ext_kernels.hip; ETD_FLAGS_EXT_KERNELS="-DETD_EMBED_ABL=$abl" python3 -m etude_amd.build > /dev/null 2>&1
  for i in 1 2 3; do timeout -k 10 400 python3 tools/probe_race.py 12 600 extractor 2>&1 | grep "^(a" | sed "s/^/k_embed ablation $abl: run $i /" | tee -a $O/race.txt; done
done
touch etude_amd/csrc/ext_kernels.hip; python3 -m etude_amd.build > /dev/null 2>&1

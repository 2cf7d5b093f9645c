#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r119; mkdir -p $O
for v in noslp late1 late2 default; do
  if [ $v = default ]; then unset ETD_LIB_PATH; else export ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_$v.so; fi
  echo "== variant $v" >> $O/variants.txt
  PROBE_REPS=4 PROBE_LINES=0 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 400 python3 tools/probe_trace.py 50 extractor >> $O/variants.txt 2>&1 || exit 1
done
grep "^==\|^rep\|alone" $O/variants.txt

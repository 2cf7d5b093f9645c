#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r115; mkdir -p $O
for v in xchg1 uniform; do
  export ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_$v.so
  echo "== variant $v" >> $O/variants.txt
  PROBE_REPS=25 PROBE_LINES=0 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 400 python3 tools/probe_trace.py 1 extractor >> $O/variants.txt 2>&1 || exit 1
done
grep "^==\|lost slots\|by j" $O/variants.txt; grep -c "exactly: \[\]" $O/variants.txt

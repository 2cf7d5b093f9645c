#!/bin/bash
# victim-kernel variants (etude_amd/variants/, built with ETD_FLAGS_DEC_KERNELS) under the k_proj256_kv6 aggressor
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r113; mkdir -p $O
for v in default xchg1 uniform nt0 uni_xchg_nt0; do
  if [ $v = default ]; then unset ETD_LIB_PATH; else export ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_$v.so; fi
  echo "== variant $v" >> $O/variants.txt
  PROBE_REPS=3 PROBE_LINES=10 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 300 python3 tools/probe_trace.py 50 extractor >> $O/variants.txt 2>&1 || exit 1
done
grep "^==\|^rep\|alone\|layer 0 slab" $O/variants.txt

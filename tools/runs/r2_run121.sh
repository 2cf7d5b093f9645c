#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r121; mkdir -p $O
for v in noslp_xchg1 noslp_uniform noslp_late1; do
  export ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_$v.so
  echo "== variant $v" >> $O/variants.txt
  PROBE_REPS=3 PROBE_LINES=0 ETD_EXT_ONLY=k_proj256_kv6 timeout -k 10 400 python3 tools/probe_trace.py 50 extractor >> $O/variants.txt 2>&1 || exit 1
done
export ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_noslp.so
echo "== variant noslp, aggressor: engine 0's batched prefills" >> $O/variants.txt
PROBE_REPS=3 PROBE_LINES=0 timeout -k 10 400 python3 tools/probe_trace.py 100 prefill >> $O/variants.txt 2>&1 || exit 1
echo "== variant noslp, aggressor: the whole Extract stage" >> $O/variants.txt
PROBE_REPS=3 PROBE_LINES=0 timeout -k 10 400 python3 tools/probe_trace.py 100 extractor >> $O/variants.txt 2>&1 || exit 1
grep "^==\|^rep\|alone" $O/variants.txt
for m in extractor "" ; do timeout -k 10 400 python3 tools/probe_race.py 12 600 $m 2>&1 | grep "^(a\|^(x) agg" | sed "s/^/noslp, probe_race mode [$m]: /" | tee -a $O/race.txt; done

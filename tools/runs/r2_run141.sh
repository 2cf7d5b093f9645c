#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
for v in default occ4 occ6 occ7 default; do
  if [ $v = default ]; then unset ETD_LIB_PATH; else export ETD_LIB_PATH=$PWD/etude_amd/variants/libetude_$v.so; fi
  echo "$v: $(ETD_AD_PAIR=1 timeout -k 10 200 python3 tools/bench_engine_overlap.py 54 340 96 2>&1 | grep '^E=' | sed -n '1p;4p' | awk '{printf "%s %s ms  ", $1, $2}')"
done

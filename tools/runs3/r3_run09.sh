#!/bin/bash
# r3_run09: attention core with two iterations ahead in flight (ETD_AD_DEPTH=2) and / or built for 4 waves per SIMD, against the shipped form
set -x
mkdir -p gpurun_out/r3_09
O=gpurun_out/r3_09/sweep.jsonl; : > $O
for lib in libetude_hip d2o4 d2o5 d1o4; do
  [ $lib = libetude_hip ] && L=etude_amd/libetude_hip.so || L=etude_amd/lib_$lib.so
  for rc in "1728 537" "432 537" "128 537" "54 340"; do set -- $rc
    echo "{\"lib\": \"$lib\"}" >> $O
    ETD_ALLOW_STALE_LIB=1 ETD_LIB_PATH=$L timeout -k 5 200 python tools/bench_step.py --rows $1 --ctx $2 >> $O 2>> gpurun_out/r3_09/err.log || { tail -5 gpurun_out/r3_09/err.log; exit 1; }
  done
done
python - <<'P'
import json
lib=None
for l in open('gpurun_out/r3_09/sweep.jsonl'):
    d=json.loads(l)
    if 'lib' in d and len(d)==1: lib=d['lib']; continue
    print(f"{lib:14s} rows {d['rows']:5d} ctx {d['ctx']:4d}: {d['ms_per_step']:.4f} ms/step  attn {d.get('attn_us')} us frac {d.get('attn_frac')}  step_frac {d['step_frac']}")
P

#!/bin/bash
# r3_run15: is the 1728-row attention launch bound by the HBM stream or by everything that goes through L2 (K/V 2.19 GB + dense slices 0.44 GB + down-projection tiles 0.22 GB)?
# ablation builds replace the dense / down weight loads by constants (wrong results on purpose); ETD_NO_ATTN_DOWN=1 = attention alone (k_dattn) + the K-concatenated GEMM
mkdir -p gpurun_out/r3_15
O=gpurun_out/r3_15/abl.jsonl; : > $O
for v in "libetude_hip:" "lib_abl_d:" "lib_abl_dg:" "libetude_hip:ETD_NO_ATTN_DOWN=1"; do lib=${v%%:*}; envs=${v#*:}
  for r in 1728 432; do
    echo "{\"variant\": \"$lib $envs\"}" >> $O
    env $envs ETD_ALLOW_STALE_LIB=1 ETD_LIB_PATH=etude_amd/$lib.so timeout -k 5 200 python tools/bench_step.py --rows $r --ctx 537 >> $O 2>> gpurun_out/r3_15/err.log || { tail -5 gpurun_out/r3_15/err.log; exit 1; }
  done
done
python - <<'P'
import json
v=None
for l in open('gpurun_out/r3_15/abl.jsonl'):
    d=json.loads(l)
    if len(d)==1: v=d['variant']; continue
    print(f"{v:40s} rows {d['rows']:5d}: {d['ms_per_step']:.4f} ms/step attn(stamp) {d.get('attn_us')} us; events {d['event_us_per_launch']}")
P

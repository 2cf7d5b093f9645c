#!/bin/bash
# r3_run21: from how many rows does the 128 x 128 QKV|up form beat the 32 x 32 one?  (ETD_QKV_MT_MIN = 1: always; default: above 512)
mkdir -p gpurun_out/r3_21
O=gpurun_out/r3_21/sweep.jsonl; : > $O
for mn in 100000 1; do for r in 54 128 216 320 432 512; do
  echo "{\"mt_min\": $mn}" >> $O
  ETD_QKV_MT_MIN=$mn timeout -k 5 200 python tools/bench_step.py --rows $r --ctx 537 >> $O 2>> gpurun_out/r3_21/err.log || { tail -5 gpurun_out/r3_21/err.log; exit 1; }
done; done
python - <<'P'
import json
v=None
for l in open('gpurun_out/r3_21/sweep.jsonl'):
    d=json.loads(l)
    if len(d)==1: v=d['mt_min']; continue
    print(f"mt_min {v:6d} rows {d['rows']:5d}: {d['ms_per_step']:.4f} ms/step  qkv_up(events) {d['event_us_per_launch'].get('k_dstep_qkv_up')} us")
P

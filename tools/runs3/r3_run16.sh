#!/bin/bash
# r3_run16: k_dstep_qkv_up_mt (128 x 128 tiles above 512 rows): bit identity (640 rows = 2 x 320 rows), then the step at 1728 / 864 / 640 rows against the 32 x 32 form (ETD_QKV_MT=0)
mkdir -p gpurun_out/r3_16
timeout -k 10 600 python -m pytest tests/test_gpu_decoder_parity.py -m gpu -x -q -k "wide or step_logits" > gpurun_out/r3_16/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r3_16/pytest.log
[ $rc -eq 0 ] || exit $rc
O=gpurun_out/r3_16/sweep.jsonl; : > $O
for mt in 1 0; do for r in 1728 864 640; do
  echo "{\"mt\": $mt}" >> $O
  ETD_QKV_MT=$mt timeout -k 5 200 python tools/bench_step.py --rows $r --ctx 537 >> $O 2>> gpurun_out/r3_16/err.log || { tail -5 gpurun_out/r3_16/err.log; exit 1; }
done; done
python - <<'P'
import json
v=None
for l in open('gpurun_out/r3_16/sweep.jsonl'):
    d=json.loads(l)
    if len(d)==1: v=d['mt']; continue
    print(f"mt {v} rows {d['rows']:5d}: {d['ms_per_step']:.4f} ms/step  qkv_up(events) {d['event_us_per_launch'].get('k_dstep_qkv_up')} us  attn {d.get('attn_us')}  step_frac {d['step_frac']}")
P

#!/bin/bash
# r3_run10: k_dstep_qkv_up with four row tiles per workgroup above 512 rows: decoder tests (goldens, digests), then the step at 1728 / 864 / 432 / 54 rows, rpt 1 / 2 / 4 / 8
set -x
mkdir -p gpurun_out/r3_10
timeout -k 10 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_decoder_parity.py tests/test_gpu_reproducibility.py -m gpu -x -q > gpurun_out/r3_10/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r3_10/pytest.log
[ $rc -eq 0 ] || exit $rc
O=gpurun_out/r3_10/sweep.jsonl; : > $O
for rpt in 1 2 4 8; do
  for rc in "1728 537" "864 537"; do set -- $rc
    echo "{\"rpt\": $rpt}" >> $O
    ETD_QKV_RPT=$rpt timeout -k 5 200 python tools/bench_step.py --rows $1 --ctx $2 >> $O 2>> gpurun_out/r3_10/err.log || { tail -5 gpurun_out/r3_10/err.log; exit 1; }
  done
done
for rpt in 1 2; do for rc in "432 537" "54 340"; do set -- $rc
    echo "{\"rpt\": $rpt}" >> $O
    ETD_QKV_RPT=$rpt timeout -k 5 200 python tools/bench_step.py --rows $1 --ctx $2 >> $O 2>> gpurun_out/r3_10/err.log || exit 1
done; done
python - <<'P'
import json
v=None
for l in open('gpurun_out/r3_10/sweep.jsonl'):
    d=json.loads(l)
    if len(d)==1: v=d['rpt']; continue
    print(f"rpt {v} rows {d['rows']:5d} ctx {d['ctx']:4d}: {d['ms_per_step']:.4f} ms/step  qkv_up(events) {d['event_us_per_launch'].get('k_dstep_qkv_up')} us  attn {d.get('attn_us')}  step_frac {d['step_frac']}")
P

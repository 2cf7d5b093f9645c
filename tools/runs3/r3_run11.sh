#!/bin/bash
# r3_run11: the driver's command, as is: does W = 5 + K = 20 on the 64-clip batch fit the harness budget, and what does the line say?
mkdir -p gpurun_out/r3_11
time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_11/bench.json 2> gpurun_out/r3_11/bench.err
echo rc=$?
grep "bench \|real" gpurun_out/r3_11/bench.err | tail -12
python - <<'P'
import json
d=json.load(open('gpurun_out/r3_11/bench.json'))
print(d['value'], d['ms_per_step'], d['scaling'], d['config']['warmup_step'], d['config']['batch_clips'])
print(json.dumps(d['roofline'])[:1200])
print(json.dumps(d.get('cpu_baseline'))[:600])
print(json.dumps(d.get('extras'))[:1800])
P

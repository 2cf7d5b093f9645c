#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r134; mkdir -p $O
for p in 1 0 1 0; do echo "== ETD_AD_PAIR=$p" >> $O/b.txt; ETD_AD_PAIR=$p timeout -k 10 500 python3 bench.py --no-cpu-baseline >> $O/b.txt 2>&1; done
python3 - <<'PY'
import json
for l in open('gpurun_out/r134/b.txt'):
    if l.startswith('=='): print(l.strip())
    if l.startswith('{"metric"'):
        d=json.loads(l); e=d['extras']
        print(d['value'], d['roofline']['avg_launch_ms'], 'single clip decode', e['single_clip']['decode_s'], 'configs[3]', e['decoder_streams']['ms_per_step'], '4k', e['decoder_streams_4k']['ms_per_step'])
PY

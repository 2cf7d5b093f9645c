#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r49; mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['extras'].items(): print(k, {x:v[x] for x in v if x in ('ms_per_window','ms_per_step','engines','instances','wall_s','error')}, v.get('roofline',{}).get('frac'))"

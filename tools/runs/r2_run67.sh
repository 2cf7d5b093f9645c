#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r67; mkdir -p $O
PROBE_DUMP=$O/ref.npy timeout -k 10 200 python3 tools/probe_race.py 12 600 hog 2>&1 | grep "^(a"
for i in 1 2 3 4 5 6; do PROBE_DUMP=$O/run$i.npy timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a"; done
python3 - <<'PY'
import numpy as np, glob
ref=np.load('gpurun_out/r67/ref.npy')
for f in sorted(glob.glob('gpurun_out/r67/run*.npy')):
    a=np.load(f)
    for e in range(3):
        d=np.argwhere((a[e]!=ref[e]).any(axis=1)).ravel()
        if len(d):
            firsts=[(int(s_), int(np.argmax(a[e,s_]!=ref[e,s_]))) for s_ in d]
            print(f, 'engine', e+1, 'streams that differ:', len(d), 'first (stream, step):', sorted(firsts, key=lambda x:x[1])[:8])
PY

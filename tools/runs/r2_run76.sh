#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r76; mkdir -p $O
PROBE_KVSUMS=$O/kv_ref.npy timeout -k 10 200 python3 tools/probe_race.py 12 600 hog 2>&1 | grep "^(a"
for i in 1 2 3; do PROBE_KVSUMS=$O/kv_run$i.npy timeout -k 10 200 python3 tools/probe_race.py 12 600 2>&1 | grep "^(a"; done
python3 - <<'PY'
import numpy as np, glob
ref=np.load('gpurun_out/r76/kv_ref.npy')
for f in sorted(glob.glob('gpurun_out/r76/kv_run*.npy')):
    a=np.load(f); d=(a!=ref)            # [layer][slot][pos]
    print(f, 'differing (layer, slot, pos) entries:', int(d.sum()))
    # per slot: earliest differing position and at which layers it differs there
    rows=[]
    for s_ in range(a.shape[1]):
        pp=np.argwhere(d[:,s_,:].any(axis=0)).ravel()
        if len(pp):
            p0=int(pp[0]); rows.append((p0, s_, np.argwhere(d[:,s_,p0]).ravel().tolist(), int(d[:,s_,:].any(axis=0).sum())))
    rows.sort()
    print('   slots affected:', len(rows), ' first (pos, slot, layers differing at that pos, #positions differing):', rows[:12])
PY
rm -f $O/*.npy

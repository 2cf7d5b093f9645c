#!/bin/bash
# job-level 2 x 2: row finish in the attention launch (ETD_ROWFIN) x fused prefill MLP (ETD_FUSED_PMLP)
export TMPDIR=/tmp
O=gpurun_out/r51; mkdir -p $O
for rep in 1 2 3; do for rf in 0; do for fm in 0 1; do
  ETD_ROWFIN=$rf ETD_FUSED_PMLP=$fm python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rowfin=$rf fused=$fm', d['value'], d['ms_per_step'], d['decoder_tokens_per_s'], d['tokens_sha256_rank0'])" | tee -a $O/job.txt
done; done; done

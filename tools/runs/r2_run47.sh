#!/bin/bash
# fused prefill MLP: parity (bit-identical A/B + the decoder suite) and job-level A/B
export TMPDIR=/tmp
O=gpurun_out/r47; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_decoder.py -x -q -k "fused_prefill or identical_to_reference or fast_paths" > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt; echo "tests rc=$rc"
[ $rc = 0 ] || exit 1
for off in 1 0 1 0; do export ETD_FUSED_PMLP=$((1-off));
  :
  python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('unfused=$off', d['value'], d['ms_per_step'], d['decoder_tokens_per_s'], {x:k[x] for x in k if 'linear' in x or 'dmlp' in x or 'ln_rows'==x or 'attn_causal' in x})" | tee -a $O/job.txt
done

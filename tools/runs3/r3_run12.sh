#!/bin/bash
# r3_run12: per-rank shares of the 64-clip batch at N = 2 / 4 / 8 (32 / 16 / 8 clips on one GPU) x 1 / 2 / 4 decoder engines: which engine count per batch size
mkdir -p gpurun_out/r3_12
O=gpurun_out/r3_12/sweep.txt; : > $O
for c in 8 16 32; do for e in 1 2 4; do
  timeout -k 10 300 python bench.py --clips $c --engines $e --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-stamp > gpurun_out/r3_12/c${c}_e$e.json 2> gpurun_out/r3_12/c${c}_e$e.err || { tail -5 gpurun_out/r3_12/c${c}_e$e.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r3_12/c${c}_e$e.json'))
print('clips $c engines $e:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms', 'decode', d['roofline']['decode_stage']['stage_s_per_step'], 'digest', d['tokens_sha256_rank0'])
" | tee -a $O
done; done
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-stamp > gpurun_out/r3_12/c64_e1.json 2> gpurun_out/r3_12/c64_e1.err || { tail -5 gpurun_out/r3_12/c64_e1.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r3_12/c64_e1.json'))
print('clips 64 engines 1 (chunked prefill passes):', d['value'], 'audio-s/s', d['ms_per_step'], 'ms', 'decode', d['roofline']['decode_stage']['stage_s_per_step'], 'digest', d['tokens_sha256_rank0'])
" | tee -a gpurun_out/r3_12/sweep.txt

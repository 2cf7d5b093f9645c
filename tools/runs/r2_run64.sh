#!/bin/bash
# cost of the end-of-kernel fences at job level + reproducibility of the whole job with them
export TMPDIR=/tmp
O=gpurun_out/r64; mkdir -p $O
job() { python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], d['decoder_tokens_per_s'], d['tokens_sha256_rank0'])" | tee -a $O/job.txt; }
touch etude_amd/csrc/dec_kernels.hip; ETD_FLAGS_DEC_KERNELS="-DETD_STEP_FENCE" python3 -m etude_amd.build > /dev/null 2>&1
TAG=fence job; TAG=fence job; TAG=fence job
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 tools/probe_determinism.py 4 216 24 3 2>&1 | grep -v amdgpu.ids | sed 's/^/fence: /' | tee -a $O/job.txt
touch etude_amd/csrc/dec_kernels.hip; python3 -m etude_amd.build > /dev/null 2>&1
TAG=plain job; TAG=plain job

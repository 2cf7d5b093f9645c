#!/bin/bash
# engine start stagger (prefills out of phase) + engine-count sweep on the current tree
export TMPDIR=/tmp
O=gpurun_out/r43; mkdir -p $O
run() { python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], d['extract_audio_s_per_s'], d['decoder_tokens_per_s'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/job.txt; }
for sg in 0 3 5.5 8; do TAG="stagger=$sg eng=4" ETD_ENGINE_STAGGER_MS=$sg run; done
for sg in 0 7 13; do TAG="stagger=$sg eng=2" ETD_ENGINE_STAGGER_MS=$sg run --engines 2; done
TAG="stagger=7 eng=3" ETD_ENGINE_STAGGER_MS=7 run --engines 3

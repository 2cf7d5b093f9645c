#!/bin/bash
# EARLY dense fragments in k_dstep_attn_down: parity, step-only and job-level A/B (ETD_AD_EARLY=0 -> old path)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r42; mkdir -p $O
python -m pytest tests/test_gpu_decoder.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt
for ea in 0 96; do
  ETD_AD_EARLY=$ea python3 tools/probe_graph_launch.py 54 320 96 2>&1 | grep engines | sed "s/^/early=$ea /" | tee -a $O/probe.txt
done
for ea in 0 96 0 96; do
  ETD_AD_EARLY=$ea python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early=$ea', d['value'], d['ms_per_step'], d['extract_audio_s_per_s'], d['decoder_tokens_per_s'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/job.txt
done

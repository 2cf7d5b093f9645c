#!/bin/bash
# r3_run02: the rebuilt bench (64-clip batch, the clip's own condition bars, device stamps): real chain, then --synthetic-bars (A/B vs r3_run01's 703),
# then the spawn path (torch.distributed.run + nccl + gather) at N = 1 on a small configuration, extras and CPU baseline included
set -x
mkdir -p gpurun_out/r3_02
timeout -k 10 500 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r3_02/real.json 2> gpurun_out/r3_02/real.err || { tail -30 gpurun_out/r3_02/real.err; exit 1; }
tail -c 2500 gpurun_out/r3_02/real.json
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --synthetic-bars > gpurun_out/r3_02/synth.json 2> gpurun_out/r3_02/synth.err || { tail -30 gpurun_out/r3_02/synth.err; exit 1; }
tail -c 1500 gpurun_out/r3_02/synth.json
ETD_FORCE_SPAWN=1 timeout -k 10 400 python bench.py --gpus 1 --clips 2 --attr-grid 4 --steps 1 --warmup 1 --seconds 30 > gpurun_out/r3_02/spawn.json 2> gpurun_out/r3_02/spawn.err || { tail -30 gpurun_out/r3_02/spawn.err; exit 1; }
tail -c 3000 gpurun_out/r3_02/spawn.json

#!/bin/bash
# r3_run18: the round's closing run on the final tree: GPU suite, tools/profile.sh (trace + PMC), then the driver's command (--steps 20 --warmup 5)
mkdir -p gpurun_out/r3_18
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3_18/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r3_18/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh > gpurun_out/r3_18/profile.log 2>&1 || { tail -20 gpurun_out/r3_18/profile.log; exit 1; }
tail -2 gpurun_out/r3_18/profile.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_18/bench.json 2> gpurun_out/r3_18/bench.err
echo rc=$?
grep "bench " gpurun_out/r3_18/bench.err | tail -4

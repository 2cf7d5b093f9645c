#!/bin/bash
# r3_run13: GPU suite (incl. the 640-row decode-step test), then tools/profile.sh (trace of the headline command + PMC passes) with the steady-state stamps
mkdir -p gpurun_out/r3_13
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_13/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r3_13/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh > gpurun_out/r3_13/profile.log 2>&1 || { tail -20 gpurun_out/r3_13/profile.log; exit 1; }
tail -3 gpurun_out/r3_13/profile.log

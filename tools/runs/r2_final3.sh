#!/bin/bash
# end-of-round evidence run on the final tree: rocprofv3 trace + PMC of the bench, the full GPU suite, the default bench line
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
bash tools/profile.sh > gpurun_out/r02c_profile_log.txt 2>&1
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r02c_pytest_gpu.txt
python3 bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; tail -c 300 gpurun_out/r02c_bench.json
tail -12 gpurun_out/profile_summary.txt

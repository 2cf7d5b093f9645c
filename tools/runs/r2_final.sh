#!/bin/bash
# end-of-round evidence run: PMC of the extractor, rocprofv3 trace + PMC of the bench, the full GPU suite, the default bench line
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
bash tools/pmc_extractor.sh > gpurun_out/r02_pmc_extractor.txt 2>&1
bash tools/profile.sh > gpurun_out/r02_profile_log.txt 2>&1
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r02_pytest_gpu.txt
python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 400 gpurun_out/r02_bench.json
tail -30 gpurun_out/profile_summary.txt

export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r2k
timeout 1700 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r2k/tests.txt
tail -12 gpurun_out/r2k/tests.txt
timeout 900 python bench.py > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err
tail -c 3000 gpurun_out/r2k/bench.json

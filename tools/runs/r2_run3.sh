mkdir -p gpurun_out/r2c
timeout 1500 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_full_configs.py -q -s -x -k "fp32 or config1" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r2c/tests.txt
tail -8 gpurun_out/r2c/tests.txt

mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_hft_wrapper.py -q -x 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r2g/tests.txt
tail -6 gpurun_out/r2g/tests.txt
timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2g/ext_fused.txt 2>&1
cat gpurun_out/r2g/ext_fused.txt

mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_hft_wrapper.py -q -x 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r2e/tests.txt
tail -4 gpurun_out/r2e/tests.txt
timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2e/ext_fused.txt 2>&1
ETD_NO_FUSED_PROJ=1 timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2e/ext_noproj.txt 2>&1
grep -A8 "ms/window" gpurun_out/r2e/ext_fused.txt | head -12

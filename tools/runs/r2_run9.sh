mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_gpu_extractor.py -q -x 2>&1 | tail -2
timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2i/ext_a.txt 2>&1
ETD_FUSED_TIME=1 timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2i/ext_time.txt 2>&1
head -6 gpurun_out/r2i/ext_a.txt; head -6 gpurun_out/r2i/ext_time.txt

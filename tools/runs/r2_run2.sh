export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests/test_gpu_full_configs.py -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r2b/tests.txt
tail -5 gpurun_out/r2b/tests.txt
for cf in 512 128 64; do ETD_CHUNK_FRAMES=$cf timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2b/ext_cf${cf}_wb4.txt 2>&1; done
ETD_CHUNK_FRAMES=64 timeout 200 python tools/bench_extractor.py 16 3 8 > gpurun_out/r2b/ext_cf64_wb8.txt 2>&1
ETD_CHUNK_FRAMES=128 timeout 200 python tools/bench_extractor.py 16 3 8 > gpurun_out/r2b/ext_cf128_wb8.txt 2>&1
ETD_CHUNK_FRAMES=32 timeout 200 python tools/bench_extractor.py 16 3 8 > gpurun_out/r2b/ext_cf32_wb8.txt 2>&1

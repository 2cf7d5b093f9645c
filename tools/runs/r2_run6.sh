mkdir -p gpurun_out/r2f
timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2f/ext_fused.txt 2>&1
cat gpurun_out/r2f/ext_fused.txt

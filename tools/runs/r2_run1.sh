export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r2a/tests.txt
for w in 4 8 16; do
  ETD_AD_WAVES=$w timeout 300 python tools/bench_engine_overlap.py 54 320 96 > gpurun_out/r2a/eo_$w.txt 2>&1
  ETD_AD_WAVES=$w timeout 300 python tools/bench_decoder_streams.py 512 > gpurun_out/r2a/ds512_$w.txt 2>&1
  ETD_AD_WAVES=$w timeout 300 python tools/bench_decoder_streams.py 3500 > gpurun_out/r2a/ds3500_$w.txt 2>&1
done
tail -5 gpurun_out/r2a/tests.txt

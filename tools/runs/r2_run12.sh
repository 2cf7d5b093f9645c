export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r2l
for pf in 0 64 128; do
  ETD_AD_FULLPF=$pf timeout 300 python tools/bench_engine_overlap.py 54 320 96 > gpurun_out/r2l/eo_pf$pf.txt 2>&1
  echo "== FULLPF $pf"; grep "E=" gpurun_out/r2l/eo_pf$pf.txt
done
ETD_AD_FULLPF=128 timeout 300 python tools/bench_decoder_streams.py 512 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('128x512 fullpf', d['ms_per_step'], d['kernel_ms_per_step'])"

mkdir -p gpurun_out/r2j
timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2j/ext_base.txt 2>&1
touch etude_amd/csrc/ext_fused.hip
ETD_FLAGS_EXT_FUSED="-mllvm -amdgpu-sched-strategy=iterative-minreg" python -m etude_amd.build 2>&1 | grep -E "error|built"
ETD_ALLOW_STALE_LIB=1 timeout 200 python tools/bench_extractor.py 16 3 4 > gpurun_out/r2j/ext_minreg.txt 2>&1
head -5 gpurun_out/r2j/ext_base.txt; head -5 gpurun_out/r2j/ext_minreg.txt

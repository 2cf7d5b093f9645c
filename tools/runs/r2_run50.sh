#!/bin/bash
# k_enc_layer: half-chunk tops pipelined into the previous half's last MFMA group (ETD_ENC_PIPE) -- parity + A/B
export TMPDIR=/tmp
O=gpurun_out/r50; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_extractor.py -x -q > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; echo "tests rc=$rc"
[ $rc = 0 ] || exit 1
python3 tools/bench_extractor.py 16 3 4 2>&1 | grep -v amdgpu.ids | head -6 | sed 's/^/pipe=1 /' | tee -a $O/ab.txt
touch etude_amd/csrc/ext_fused.hip; ETD_FLAGS_EXT_FUSED="-DETD_ENC_PIPE=0" python3 -m etude_amd.build > /dev/null 2>&1
python3 tools/bench_extractor.py 16 3 4 2>&1 | grep -v amdgpu.ids | head -6 | sed 's/^/pipe=0 /' | tee -a $O/ab.txt
touch etude_amd/csrc/ext_fused.hip; python3 -m etude_amd.build > /dev/null 2>&1
python3 tools/bench_extractor.py 16 3 4 2>&1 | grep -v amdgpu.ids | head -3 | sed 's/^/pipe=1 /' | tee -a $O/ab.txt

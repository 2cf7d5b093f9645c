#!/bin/bash
# what do the weight streams of the attention launch cost the other engines?  ablation builds (wrong results on purpose)
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r18; mkdir -p $O
run() {   # name, flags
  touch etude_amd/csrc/dec_kernels.hip
  ETD_EXTRA_FLAGS="$2" python3 -m etude_amd.build > $O/build_$1.txt 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.txt; return; }
  echo "== $1 ($2) ctx 320" >> $O/log.txt
  timeout 300 python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E=" >> $O/log.txt
  echo "== $1 ($2) ctx 16" >> $O/log.txt
  timeout 300 python3 tools/bench_engine_overlap.py 54 16 96 2>&1 | grep "^E=" >> $O/log.txt
}
run base ""
run nodense "-DETD_ABL_DENSE"
run nogemmw "-DETD_ABL_GEMMW"
run noqkvw "-DETD_ABL_QKVW"
run noall "-DETD_ABL_DENSE -DETD_ABL_GEMMW -DETD_ABL_QKVW"
cat $O/log.txt

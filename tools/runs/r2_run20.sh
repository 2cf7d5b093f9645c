#!/bin/bash
# A/B on one box: k_enc_layer ring barriers as __syncthreads() (round-2 form, drains the DMAs) vs raw s_barrier
export TMPDIR=/tmp
O=gpurun_out/r20; mkdir -p $O
run() {
  touch etude_amd/csrc/ext_fused.hip
  ETD_EXTRA_FLAGS="$2" python3 -m etude_amd.build > $O/build_$1.txt 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.txt; return; }
  echo "== $1 ($2)" >> $O/log.txt
  for i in 1 2; do python3 tools/bench_extractor.py 16 5 4 2>&1 | grep -E "ms/window" | head -4 >> $O/log.txt; done
}
run sync "-DETD_ENC_SYNCTHREADS"
run raw ""
run sync2 "-DETD_ENC_SYNCTHREADS"
run raw2 ""
cat $O/log.txt

#!/bin/bash
# where does the in-launch row finish lose its time?  (1) write-through stores + drain only, (2) + arrivals, (3) everything
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r23; mkdir -p $O
run() {
  touch etude_amd/csrc/dec_kernels.hip
  ETD_EXTRA_FLAGS="$2" python3 -m etude_amd.build > $O/build_$1.txt 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.txt; return; }
  echo "== $1 ($2) $3" >> $O/log.txt
  env $3 timeout 300 python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E=" >> $O/log.txt
}
run old "" ETD_ROWFIN=0
run stores_only "-DETD_FIN_ABL=1" ETD_ROWFIN=1
run stores_arrive "-DETD_FIN_ABL=2" ETD_ROWFIN=1
run full "" ETD_ROWFIN=1
cat $O/log.txt

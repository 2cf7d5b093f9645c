#!/bin/bash
# the rare path of the row finish: a down-projection unit arrives last and finishes rows (forced by a test build)
export GPU_MAX_HW_QUEUES=8
export ETD_ROWFIN=1   # the in-launch row finish is opt-in
export TMPDIR=/tmp
O=gpurun_out/r25; mkdir -p $O
ETD_ROWFIN=0 timeout 600 python3 tools/ab_tokens.py 54 320 40 2 > $O/a_full.txt 2>&1; grep "^rep" $O/a_full.txt > $O/a.txt; tail -3 $O/a_full.txt
touch etude_amd/csrc/dec_kernels.hip
ETD_EXTRA_FLAGS="-DETD_FIN_TEST_SLOWGEMM" python3 -m etude_amd.build > $O/build.txt 2>&1 || { echo build failed; tail $O/build.txt; exit 1; }
timeout 900 python3 tools/ab_tokens.py 54 320 40 2 2>&1 | grep "^rep" > $O/b.txt
if cmp -s $O/a.txt $O/b.txt; then echo "TOKENS IDENTICAL with units finishing rows ($(wc -l < $O/a.txt) digests)"; else echo "TOKENS DIFFER"; diff $O/a.txt $O/b.txt | head; fi
python3 tools/bench_engine_overlap.py 54 320 24 2>&1 | grep "^E=1" | head -1

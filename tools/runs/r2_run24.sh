#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export ETD_ROWFIN=1   # the in-launch row finish is opt-in
export TMPDIR=/tmp
O=gpurun_out/r24; mkdir -p $O
ETD_ROWFIN=0 timeout 600 python3 tools/ab_tokens.py 54 320 160 4 2>&1 | grep "^rep" > $O/a.txt
timeout 600 python3 tools/ab_tokens.py 54 320 160 4 2>&1 | grep "^rep" > $O/b.txt
if cmp -s $O/a.txt $O/b.txt; then echo "TOKENS IDENTICAL ($(wc -l < $O/a.txt) digests)"; else echo "TOKENS DIFFER"; diff $O/a.txt $O/b.txt | head; fi
for i in 1 2; do
echo "== old"; ETD_ROWFIN=0 python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="
echo "== new"; python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="
done

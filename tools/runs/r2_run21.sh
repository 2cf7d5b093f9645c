#!/bin/bash
# row finish inside k_dstep_attn_down: bit-exactness against the separate row kernel, decoder parity tests, step time
export GPU_MAX_HW_QUEUES=8
export ETD_ROWFIN=1   # the in-launch row finish is opt-in
export TMPDIR=/tmp
O=gpurun_out/r21; mkdir -p $O
ETD_ROWFIN=0 timeout 600 python3 tools/ab_tokens.py 54 320 160 4 > $O/tok_old.txt 2>&1
timeout 600 python3 tools/ab_tokens.py 54 320 160 4 > $O/tok_new.txt 2>&1
timeout 600 python3 tools/ab_tokens.py 54 320 160 4 > $O/tok_new2.txt 2>&1
grep "^rep" $O/tok_old.txt > $O/a.txt; grep "^rep" $O/tok_new.txt > $O/b.txt; grep "^rep" $O/tok_new2.txt > $O/c.txt
if cmp -s $O/a.txt $O/b.txt && cmp -s $O/a.txt $O/c.txt; then echo "TOKENS IDENTICAL ($(wc -l < $O/a.txt) digests)"; else echo "TOKENS DIFFER"; diff $O/a.txt $O/b.txt | head; tail -3 $O/tok_new.txt; fi
echo "== old (separate row kernel)"; ETD_ROWFIN=0 python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="
echo "== new (row finish in launch)"; python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="
echo "== old"; ETD_ROWFIN=0 python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="
echo "== new"; python3 tools/bench_engine_overlap.py 54 320 96 2>&1 | grep "^E="
python3 -m pytest tests/test_gpu_decoder.py -x -q -m gpu 2>&1 | tail -4

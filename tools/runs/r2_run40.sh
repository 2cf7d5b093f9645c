#!/bin/bash
# re-entry check: GPU suite + default bench on the restored tree
export TMPDIR=/tmp
O=gpurun_out/r40; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt; tail -3 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json

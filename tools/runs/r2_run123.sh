#!/bin/bash
# the fixed build (softmax merge and fp32 GEMV kept out of crossed packed-FP32 forms): probes, GPU suite, bench twice
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
O=gpurun_out/r123; mkdir -p $O
for m in "extractor:k_proj256_kv6" "extractor:" "prefill:"; do
  mode=${m%%:*}; only=${m#*:}
  echo "== probe_trace, aggressor $mode $only" >> $O/trace.txt
  PROBE_REPS=3 PROBE_LINES=8 ETD_EXT_ONLY=$only timeout -k 10 400 python3 tools/probe_trace.py 100 $mode >> $O/trace.txt 2>&1 || exit 1
done
grep "^==\|^rep\|alone\|   step" $O/trace.txt
for m in extractor "" ; do timeout -k 10 400 python3 tools/probe_race.py 12 600 $m 2>&1 | grep "^(a" | sed "s/^/probe_race mode [$m]: /" | tee -a $O/race.txt; done
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for i in 1 2; do timeout -k 10 500 python3 bench.py --no-cpu-baseline > $O/bench$i.txt 2>&1; grep -o '"value": [0-9.]*\|"tokens_sha256_rank0": "[0-9a-f]*"' $O/bench$i.txt | tr '\n' ' '; echo; done

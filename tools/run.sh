#!/bin/bash
# The ONE parametrised gpurun payload (rounds 1-3 kept a script per run under tools/runs*/: those are in the history, commits 4ac2f57 and 7d9ba3c).
#
#   gpurun --timeout 900 -- 'tools/run.sh TAG "name=command" ["name=command" ...]'
#
# Every step runs under `timeout -k 10 $STEP_TIMEOUT` (default 600 s), stdout -> gpurun_out/TAG/name.out, stderr -> gpurun_out/TAG/name.err; the first
# failing step ends the run (no further GPU step after a timeout / fault).  The tail of every step's stdout is echoed.  Examples:
#   tools/run.sh t1 "tests=python -m pytest tests -m gpu -x -q"
#   tools/run.sh b1 "bench=python bench.py --steps 1 --warmup 1" "step=python tools/bench_step.py --rows 1728 --ctx 537"
#   A/B of an environment switch: two steps with the variable set in the command ("a=ETD_X=0 python ...", "b=ETD_X=1 python ...").
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for step in "$@"; do
  name=${step%%=*}; cmd=${step#*=}
  echo "== [$TAG/$name] $cmd"
  timeout -k 10 "${STEP_TIMEOUT:-600}" bash -c "$cmd" > "$OUT/$name.out" 2> "$OUT/$name.err"; rc=$?
  tail -n "${TAIL:-25}" "$OUT/$name.out"
  if [ $rc -ne 0 ]; then echo "== [$TAG/$name] exit $rc"; tail -n 30 "$OUT/$name.err"; exit $rc; fi
done

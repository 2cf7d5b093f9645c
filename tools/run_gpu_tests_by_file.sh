#!/bin/bash
# The GPU suite one file per process, in order, stopping at the first file that fails (a GPU fault kills the process: the file it died in is then known and nothing runs after it).
#   gpurun -- 'tools/run_gpu_tests_by_file.sh TAG'
set -u
TAG=${1:-bf}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for f in tests/test_gpu_*.py tests/test_bench*.py; do
  [ -f "$f" ] || continue
  n=$(basename "$f" .py)
  echo "== $f"
  PYTHONUNBUFFERED=1 timeout -k 10 "${STEP_TIMEOUT:-600}" python -m pytest "$f" -m gpu -x -v -p no:cacheprovider > "$OUT/$n.out" 2> "$OUT/$n.err"; rc=$?
  tail -n 3 "$OUT/$n.out"
  if [ $rc -ne 0 ] && [ $rc -ne 5 ]; then echo "== $f exit $rc"; tail -n 15 "$OUT/$n.err"; exit $rc; fi
done

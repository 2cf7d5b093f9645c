#!/bin/bash
# job-level A/B of the in-launch row finish (bench.py, default workload): separate row kernel vs row finish at 6 / 4 waves per SIMD
export TMPDIR=/tmp
O=gpurun_out/r27; mkdir -p $O
val() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", d["value"], "dec tok/s", round(d["decoder_tokens_per_s"]), "attn_down avg us", round(1e3 * d["roofline"]["avg_launch_ms"], 2))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
runb() {  # name flags env
  touch etude_amd/csrc/dec_kernels.hip
  ETD_EXTRA_FLAGS="$2" python3 -m etude_amd.build > $O/build_$1.txt 2>&1 || { echo "build $1 failed"; return; }
  env $3 python3 bench.py --steps 2 --warmup 1 > $O/$1.json 2> $O/$1.err; val $O/$1.json
}
runb old1 "" ETD_ROWFIN=0
runb fin6a "-DETD_FIN_OCC=6 -DETD_FIN_ZB=6" ETD_ROWFIN=1
runb fin4a "-DETD_FIN_OCC=4 -DETD_FIN_ZB=12" ETD_ROWFIN=1
runb old2 "" ETD_ROWFIN=0
runb fin6b "-DETD_FIN_OCC=6 -DETD_FIN_ZB=6" ETD_ROWFIN=1
runb fin5 "-DETD_FIN_OCC=5 -DETD_FIN_ZB=12" ETD_ROWFIN=1

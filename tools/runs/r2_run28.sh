#!/bin/bash
# job-level A/B: register budget of k_dstep_attn_down (waves per SIMD it is built for)
export TMPDIR=/tmp
O=gpurun_out/r28; mkdir -p $O
val() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", d["value"], "dec tok/s", round(d["decoder_tokens_per_s"]), "attn_down avg us", round(1e3 * d["roofline"]["avg_launch_ms"], 2))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
runb() {  # name flags
  touch etude_amd/csrc/dec_kernels.hip
  ETD_EXTRA_FLAGS="$2" python3 -m etude_amd.build > $O/build_$1.txt 2>&1 || { echo "build $1 failed"; return; }
  python3 bench.py --steps 2 --warmup 1 > $O/$1.json 2> $O/$1.err; val $O/$1.json
}
runb occ7a ""
runb occ5a "-DETD_AD_OCC=5"
runb occ4a "-DETD_AD_OCC=4"
runb occ7b ""
runb occ5b "-DETD_AD_OCC=5"
runb occ3 "-DETD_AD_OCC=3"

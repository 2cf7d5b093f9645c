#!/bin/bash
# job-level knob sweep (bench.py --steps 2): skinny-GEMM K split, row-kernel rows per workgroup, attention waves per workgroup
export TMPDIR=/tmp
O=gpurun_out/r29; mkdir -p $O
val() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print(sys.argv[1], "value", d["value"], "dec tok/s", round(d["decoder_tokens_per_s"]), "attn_down us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "qkv_up ms", round(k.get("k_dstep_qkv_up", 0)), "resid ms", round(k.get("k_resid_ln_rows", 0)))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
runb() {  # name flags env
  touch etude_amd/csrc/dec_kernels.hip
  ETD_EXTRA_FLAGS="$2" python3 -m etude_amd.build > $O/build_$1.txt 2>&1 || { echo "build $1 failed"; tail -3 $O/build_$1.txt; return; }
  env $3 python3 bench.py --steps 2 --warmup 1 > $O/$1.json 2> $O/$1.err; val $O/$1.json
}
runb base1 "" X=1
runb dsw4 "-DDS_WAVES=4" X=1
runb rl1 "-DETD_RL_ROWS=1" X=1
runb rl2 "-DETD_RL_ROWS=2" X=1
runb adw8 "" ETD_AD_WAVES=8
runb base2 "" X=1
runb q16 "" GPU_MAX_HW_QUEUES=16

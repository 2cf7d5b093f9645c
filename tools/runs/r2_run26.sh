#!/bin/bash
export ETD_ROWFIN=1   # the in-launch row finish is opt-in
export TMPDIR=/tmp
O=gpurun_out/r26; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r26/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "extract", d["extract_audio_s_per_s"], "dec tok/s", d["decoder_tokens_per_s"])
print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_ms","launches")}, d["roofline"]["decode_stage"]["frac"])
print({k: round(v,1) for k,v in list(d["kernel_ms_per_step"].items())[:8]})
print(d["extras"]["decoder_streams"]["ms_per_step"], d["extras"]["decoder_streams"]["roofline"]["frac"], d["extras"]["decoder_streams_4k"]["ms_per_step"], d["extras"]["decoder_streams_4k"]["roofline"]["frac"])
PY
ETD_ROWFIN=0 python3 bench.py > $O/bench_old.json 2> $O/bench_old.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r26/bench_old.json").read().strip().splitlines()[-1])
print("OLD value", d["value"], "ms_per_step", d["ms_per_step"], "dec tok/s", d["decoder_tokens_per_s"])
PY

mkdir -p gpurun_out/r2h
for E in 4 2 1 3; do
  timeout 400 python bench.py --engines $E --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2h/bench_E$E.json 2> gpurun_out/r2h/bench_E$E.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2h/bench_E$E.json").read().strip().splitlines()[-1])
print("E=$E", d["value"], d["ms_per_step"], d["extract_audio_s_per_s"], d["decoder_tokens_per_s"])
PY
done

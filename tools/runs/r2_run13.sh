export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r2m
timeout 300 python tools/diag_error_budget.py 0 > gpurun_out/r2m/error_budget.txt 2>&1
cat gpurun_out/r2m/error_budget.txt | grep -v amdgpu.ids
bash tools/pmc_extractor.sh > gpurun_out/r02_pmc_extractor.txt 2>&1
grep -A12 "HBM traffic of the extractor" gpurun_out/r02_pmc_extractor.txt
timeout 900 python bench.py > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2m/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['extract_audio_s_per_s'], d['decoder_tokens_per_s'], d['roofline']['frac'], d['extras']['extractor_only']['ms_per_window'])
PY

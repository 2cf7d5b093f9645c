#!/bin/bash
# r3_run17: extract stage of the 64-clip batch by windows per extractor batch (ETD_WB) and extractor instances
mkdir -p gpurun_out/r3_17
for cfg in "4 3" "8 3" "8 2" "11 2" "2 4"; do set -- $cfg
  ETD_WB=$1 timeout -k 10 200 python bench.py --ext-engines $2 --steps 1 --warmup 1 --max-bars 2 --no-extras --no-cpu-baseline --no-stamp --no-serial-pass > gpurun_out/r3_17/wb$1_e$2.json 2> gpurun_out/r3_17/wb$1_e$2.err || { tail -5 gpurun_out/r3_17/wb$1_e$2.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r3_17/wb$1_e$2.json'))
print('windows per batch $1, extractor instances $2: extract', d['extract_audio_s_per_s'], 'audio-s/s')"
done

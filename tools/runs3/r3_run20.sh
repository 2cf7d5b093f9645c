#!/bin/bash
# r3_run20: one stream on the fused step kernels (ETD_FUSED_M1=1) instead of the GEMV path: decoder tests, then configs[1] (one clip, one job)
mkdir -p gpurun_out/r3_20
ETD_FUSED_M1=1 timeout -k 10 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_decoder_parity.py tests/test_gpu_full_configs.py -m gpu -q > gpurun_out/r3_20/pytest.log 2>&1
tail -15 gpurun_out/r3_20/pytest.log | cut -c1-300
for v in 0 1; do
ETD_FUSED_M1=$v python - <<'P'
import os, time, numpy as np, torch, sys
sys.path.insert(0, '.')
import bench
from etude_amd import synth
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
dev = torch.device('cuda:0')
dec = EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()), synth.decoder_state_dict(1, {}), dev, precision='bf16', max_streams=1)
v = bench.make_vocab()
bars = synth.song_bars(seed=1234, n_bars=92)
at = [synth.attrs(1, 1, 1, 2)] * 92
st = {}
dec.generate_many([(bars, at)], v, stats=st, force_bar_tokens=48)
torch.cuda.synchronize()
t = time.perf_counter(); dec.generate_many([(bars, at)], v, stats=st, force_bar_tokens=48); torch.cuda.synchronize(); dt = time.perf_counter() - t
print('ETD_FUSED_M1', os.environ.get('ETD_FUSED_M1'), 'one job 92 bars x 48 tokens:', round(dt, 4), 's', round(st['tokens'] / dt, 1), 'tokens/s')
P
done

#!/usr/bin/env python3
"""Do a job's greedy tokens depend on the batch it is decoded in?  bf16: yes for small batches -- a batched prefill of <= 512 rows and a step
of one stream take other kernel paths (skinny split-K GEMMs, GEMV) than the big-tile / fused-step paths, with other (equally valid) bf16
roundings; from 16 jobs on the paths and the tokens are the same as in the batch of 54.  fp32: no.  (This is why bench.py --pipeline, where
admission timing decides the batch compositions, prints a different token digest from run to run; the default mode does not.)"""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from test_gpu_reproducibility import _jobs, _vocab
from etude_amd import synth
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines
cfg = EtudeDecoderConfig(**synth.decoder_dims())
dec = EtudeDecoder(cfg, synth.decoder_state_dict(1, {}), "cuda", precision="f16", max_streams=54)
jobs, v = _jobs(54, 6), _vocab()
full = run_engines([dec], jobs, v, force_bar_tokens=24)()[0]
for n in (1, 2, 4, 8, 16):
    part = run_engines([dec], jobs[:n], v, force_bar_tokens=24)()[0]
    d = sum(1 for a, b in zip(part, full[:n]) if a != b)
    print(f"first {n} job(s) alone vs the same jobs inside the batch of 54: {d} differ", flush=True)
decf = EtudeDecoder(cfg, synth.decoder_state_dict(1, {}), "cuda", precision="fp32", max_streams=54)
fullf = run_engines([decf], jobs[:16], v, force_bar_tokens=24)()[0]
for n in (1, 2, 4):
    part = run_engines([decf], jobs[:n], v, force_bar_tokens=24)()[0]
    print(f"fp32: first {n} job(s) alone vs inside a batch of 16: {sum(1 for a, b in zip(part, fullf[:n]) if a != b)} differ", flush=True)

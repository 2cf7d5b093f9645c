#!/usr/bin/env python3
"""BASELINE configs[2]-style extractor-only run (N windows of 512 frames) -- for rocprofv3 / PMC runs."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import synth  # noqa: E402
from etude_amd.config import ExtractorConfig  # noqa: E402
from etude_amd.extractor import AMTAPC_Extractor  # noqa: E402

if __name__ == "__main__":
    nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    wb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = torch.device("cuda:0")
    ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(7), "cuda", max_windows=wb)
    xs = torch.from_numpy(synth.window_features(5, nwin)).to(dev)
    ex.transcript_windows(xs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ex.transcript_windows(xs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / (max(reps, 1) * nwin)
    if reps > 0:
        print(f"{dt*1e3:.3f} ms/window  {ex.window_flops/dt/1e12:.1f} TFLOP/s  {8.192/dt:.0f} audio-s/s")
    from etude_amd import _lib
    _lib.prof_reset(); _lib.prof_enable(True)
    ex.transcript_windows(xs)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    for k, v in sorted(_lib.prof_report().items(), key=lambda kv: -kv[1]["ms"]):
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        print(f"   {k:14s} {v['ms']/nwin:8.3f} ms/window  {v['launches']//nwin:4d} launches/window  {tf:7.1f} TFLOP/s")

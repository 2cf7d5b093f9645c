#!/usr/bin/env python3
"""Where does the wall time of one extract() go? (GPU box)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import synth  # noqa: E402
from etude_amd.config import ExtractorConfig  # noqa: E402
from etude_amd.extractor import AMTAPC_Extractor  # noqa: E402

dev = torch.device("cuda:0")
cfg = ExtractorConfig()
ex = AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), "cuda")
wav = torch.from_numpy(synth.clip_audio(seed=1234, seconds=180.0)).to(dev)
inf = cfg.infer
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feat = ex._front(44100)(wav); torch.cuda.synchronize(); t1 = time.perf_counter()
    on, off, mpe, vel = ex.transcript(feat); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = [t.cpu().numpy() for t in (on, off, mpe, vel)]; t3 = time.perf_counter()
    arr = ex._mpe2note_array(*a, inf.onset_threshold, inf.offset_threshold, inf.frame_threshold); t4 = time.perf_counter()
    arr = arr[~((arr["offset"] - arr["onset"]) < inf.min_duration)]
    notes = ex._notes_from_array(arr); t5 = time.perf_counter()
    print(f"front {1e3*(t1-t0):.1f} ms | model {1e3*(t2-t1):.1f} ms | D2H {1e3*(t3-t2):.1f} ms | mpe2note C++ {1e3*(t4-t3):.1f} ms ({len(arr)} kept) | dicts {1e3*(t5-t4):.1f} ms")
    t6 = time.perf_counter()
    arr2 = ex.mpe2note_device(on, off, mpe, vel, inf.onset_threshold, inf.offset_threshold, inf.frame_threshold); t7 = time.perf_counter()
    arr2 = arr2[~((arr2["offset"] - arr2["onset"]) < inf.min_duration)]
    assert np.array_equal(arr, arr2)
    t8 = time.perf_counter()
    n3 = ex.extract_notes(wav, 44100, inf.min_duration); t9 = time.perf_counter()
    print(f"   device mpe2note {1e3*(t7-t6):.2f} ms (identical notes) | extract_notes end to end {1e3*(t9-t8):.1f} ms")

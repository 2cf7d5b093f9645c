#!/usr/bin/env python3
"""Where does the bf16 mode's error come from?  One full-size window (n_frame = 512) through the bf16 path and through the fp32
parity mode (csrc/ext_fp32.hip, itself within 4e-6 of the reference), activations tapped after every stage."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import synth  # noqa: E402
from etude_amd.config import ExtractorConfig  # noqa: E402
from etude_amd.extractor import AMTAPC_Extractor  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    sd = synth.extractor_state_dict(seed)
    x = torch.from_numpy(synth.window_features(5, 1)).to(dev)
    names = ["embed", "enc0", "enc1", "enc2", "dec0", "dec1", "dec2", "time_in", "time0", "time1", "time2"]
    rows = [512 * 256] * 4 + [512 * 88] * 7
    res = {}
    for prec, dt in (("fp32", torch.float32), ("f16", torch.float16)):
        ex = AMTAPC_Extractor(ExtractorConfig(), sd, "cuda", max_windows=1, precision=prec)
        bufs = [torch.zeros((r, 256), dtype=dt, device=dev) for r in rows]
        for s, b in enumerate(bufs):
            ex.debug_tap(s, b)
        vl = torch.zeros((512, 88, 128), dtype=torch.float32, device=dev)
        ex.debug_velocity_logits(vl)
        out = [t.float().cpu().numpy() for t in ex.transcript_windows(x)]
        torch.cuda.synchronize()
        res[prec] = ([b.float().cpu().numpy() for b in bufs], out, vl.cpu().numpy())
        ex.close()
    print(f"{'stage':8s} {'max|err|/max|ref|':>18s} {'rms err / rms ref':>18s}")
    for i, n in enumerate(names):
        a, b = res["fp32"][0][i], res["f16"][0][i]
        print(f"{n:8s} {np.abs(a - b).max() / np.abs(a).max():18.4e} {np.sqrt(((a - b) ** 2).mean()) / np.sqrt((a ** 2).mean()):18.4e}")
    for j, n in enumerate(("onset", "offset", "mpe")):
        a, b = res["fp32"][1][j], res["f16"][1][j]
        print(f"p_{n:6s} max {np.abs(a - b).max():.3e}  mean {np.abs(a - b).mean():.3e}   fraction of frames on opposite sides of 0.5: {float(((a >= 0.5) != (b >= 0.5)).mean()):.4f}")
    lg32, lg16 = res["fp32"][2], res["f16"][2]
    print(f"velocity logits: max {np.abs(lg32 - lg16).max():.3e}, argmax agreement {float((res['fp32'][1][3] == res['f16'][1][3]).mean()):.4f}")

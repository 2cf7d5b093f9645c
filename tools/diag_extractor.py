#!/usr/bin/env python3
"""Bring-up diagnostic (GPU box): per-stage error of the HIP extractor against the CPU oracle.
Prints statistics instead of asserting; the pytest -m gpu tests carry the tolerances."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from etude_amd import synth  # noqa: E402
from etude_amd.config import ExtractorConfig  # noqa: E402
from etude_amd.extractor import AMTAPC_Extractor  # noqa: E402
from oracle import hft, mel  # noqa: E402


def stats(name, got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    rel = d.max() / (np.abs(ref).max() + 1e-12)
    print(f"  {name:14s} max|d|={d.max():.4e} mean|d|={d.mean():.4e} ref_absmax={np.abs(ref).max():.3f} rel={rel:.3e} "
          f"nan={int(np.isnan(got).sum())}", flush=True)


def main():
    nf = int(os.environ.get("NF", "64"))
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    print("device", torch.cuda.get_device_name(0), flush=True)
    # ---- front end
    from etude_amd.frontend import FrontEnd
    wav = synth.clip_audio(seed=3, seconds=2.0)
    fe = FrontEnd(44100)
    t0 = time.time()
    feat = fe(torch.from_numpy(wav).to(dev)).cpu()
    ref = mel.wav2feature(torch.from_numpy(wav), 44100)
    print("frontend", tuple(feat.shape), tuple(ref.shape), f"{time.time()-t0:.2f}s")
    res_ref = mel.resample(torch.mean(torch.from_numpy(wav), 0), 44100, 16000)
    stats("resample", fe.last_resampled.cpu().numpy(), res_ref.numpy())
    stats("logmel", feat.numpy(), ref.numpy())
    big = ref.numpy() > -10
    stats("logmel(>-10)", feat.numpy()[big], ref.numpy()[big])

    # ---- model, one window, per-stage taps
    cfg = ExtractorConfig()
    cfg.input.num_frame = nf
    sdn = synth.extractor_state_dict(7, dict(n_frame=nf))
    ex = AMTAPC_Extractor(cfg, sdn, "cuda", max_windows=1, chunk_frames=nf)
    d = hft.HftDims(n_frame=nf)
    x = synth.window_features(5, 1, 256, nf + 64)
    taps = {}
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    t0 = time.time()
    r = hft.model_forward(sd, torch.from_numpy(x), d, taps)
    print(f"oracle forward {time.time()-t0:.1f}s")
    nn = 88
    bufs = {}
    shapes = {0: nf * 256, 1: nf * 256, 2: nf * 256, 3: nf * 256, 4: nf * nn, 5: nf * nn, 6: nf * nn, 7: nn * nf, 8: nn * nf, 9: nn * nf, 10: nn * nf}
    for s, rows in shapes.items():
        bufs[s] = torch.zeros((rows, 256), dtype=torch.bfloat16, device=dev)
        ex.debug_tap(s, bufs[s])
    vl = torch.zeros((nf, nn, 128), dtype=torch.float32, device=dev)
    ex.debug_velocity_logits(vl)
    out = ex.transcript_windows(torch.from_numpy(x).to(dev), want_A=True)
    torch.cuda.synchronize()
    names = {0: "embed", 1: "enc0", 2: "enc1", 3: "enc2", 4: "dec0", 5: "dec1", 6: "dec2", 7: "time_in", 8: "time0", 9: "time1", 10: "time2"}
    for s, nm in names.items():
        stats(nm, bufs[s].float().cpu().numpy().reshape(-1), taps[nm].numpy().reshape(-1))
    on = ["onset_A", "offset_A", "mpe_A", "vel_A", "onset_B", "offset_B", "mpe_B", "vel_B"]
    refs = [r[0], r[1], r[2], r[3].argmax(-1), r[5], r[6], r[7], r[8].argmax(-1)]
    for nme, g, rf in zip(on, out, refs):
        g = g.cpu().numpy(); rf = rf[0].numpy()
        if g.dtype == np.int8:
            print(f"  {nme:14s} argmax agreement {(g == rf).mean():.4f}")
        else:
            stats(nme, g, rf)
    stats("vel_logits_B", vl.cpu().numpy(), r[8][0].numpy())
    lg = r[8][0].numpy()
    gv = out[7].cpu().numpy().astype(np.int64)
    chosen = np.take_along_axis(lg, gv[..., None], -1)[..., 0]
    print("  velocity: max(oracle_max - oracle_logit[kernel_argmax]) =", float((lg.max(-1) - chosen).max()))
    # ---- timing of the full-size window path
    if os.environ.get("TIME", "1") == "1":
        cfg2 = ExtractorConfig()
        ex2 = AMTAPC_Extractor(cfg2, synth.extractor_state_dict(7), "cuda", max_windows=1)
        xs = torch.from_numpy(synth.window_features(5, 4)).to(dev)
        for _ in range(2):
            ex2.transcript_windows(xs)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            ex2.transcript_windows(xs)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 12
        print(f"full-size window: {dt*1e3:.2f} ms/window  -> {ex2.window_flops/dt/1e12:.1f} TFLOP/s, {8.192/dt:.0f} audio-s/s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The headline's chain in the EXACT-PARITY mode (fp32 extractor + fp32 decoder) on C clips x 27 attribute tuples: what `extras.parity_mode` of bench.py runs, at any
batch size / engine layout, with an optional eager HIP-event breakdown per kernel.

    python tools/bench_parity.py [--clips 8] [--engines 0] [--max-bars 0] [--events] [--json out.json]

Prints ONE JSON line: extract / decode / notes seconds, audio-s/s, decoder tokens/s, the token digest (sha256 of every job's ids in job order), the decode-step
byte counts of the library (`etd_decoder_stats`) and, with --stamp, the device-stamped roofline fraction of the fp32 attention launches."""
import argparse
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.config import ExtractorConfig  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402
from etude_amd.extractor import AMTAPC_Extractor  # noqa: E402
from etude_amd.pipeline import ClipBatchPipeline, attr_grid, synthetic_tempo  # noqa: E402
from etude_amd.vocab import Vocab  # noqa: E402


def make_vocab():
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--engines", type=int, default=0, help="0 = one engine from 512 jobs up, four below")
    ap.add_argument("--ext-engines", type=int, default=1)
    ap.add_argument("--max-bars", type=int, default=0)
    ap.add_argument("--bar-tokens", type=int, default=48)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--ext-precision", default="")
    ap.add_argument("--prefill-rows", type=int, default=0)
    ap.add_argument("--events", action="store_true", help="one more 2-bar pass with HIP events around every launch (serial)")
    ap.add_argument("--stamp", action="store_true", help="one more decode stage with the attention launches stamped on the device")
    ap.add_argument("--skip-extract", action="store_true", help="condition bars from the bf16 extractor (decoder-only A/B runs)")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    grid = attr_grid(27)
    vocab = make_vocab()
    n_jobs = a.clips * len(grid)
    n_eng = a.engines or (1 if n_jobs >= 512 else 4)
    per_eng = (n_jobs + n_eng - 1) // n_eng
    extp = a.ext_precision or ("f16" if a.skip_extract else a.precision)
    exs = [AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(0), dev, max_windows=4, precision=extp) for _ in range(a.ext_engines)]
    rows = a.prefill_rows or min(262144 if a.precision == "f16" else 131072, per_eng * 520)
    decs = [EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()), synth.decoder_state_dict(1, {}), dev, precision=a.precision, max_streams=per_eng, max_prefill_rows=rows)]
    decs += [decs[0].clone() for _ in range(n_eng - 1)]
    pipe = ClipBatchPipeline(exs, decs, vocab, synthetic_tempo(), grid, 44100, force_bar_tokens=a.bar_tokens)
    wavs = [synth.clip_audio_device(seed=ci, seconds=180.0, device=dev) for ci in range(a.clips)]      # bench.py's batch: seeds 0 .. clips - 1
    conds = pipe.extract_stage(wavs)
    pipe.decode_stage(conds, max_bars=2)
    torch.cuda.synchronize(dev)
    for d in decs:
        d.stats_reset()
    t0 = time.perf_counter()
    conds = pipe.extract_stage(wavs)
    t1 = time.perf_counter()
    res, st = pipe.decode_stage(conds, max_bars=a.max_bars)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    pipe.notes_stage(conds, res)
    t3 = time.perf_counter()
    stats = [d.stats() for d in decs]
    ntok = sum(s["tokens"] for s in st)
    nb = float(np.mean([len(cd.bars) for cd in conds]))
    frac = 1.0 if not a.max_bars else min(1.0, a.max_bars / nb)
    dec_bytes = sum(s["kv_bytes"] + s["steps"] * s["weight_bytes_per_step"] for s in stats)
    out = {"clips": a.clips, "jobs": n_jobs, "engines": n_eng, "streams_per_engine": per_eng, "precision": a.precision, "ext_precision": extp, "max_bars": a.max_bars,
           "extract_s": round(t1 - t0, 3), "decode_s": round(t2 - t1, 3), "notes_s": round(t3 - t2, 3),
           "audio_s_per_s": round(180.0 * a.clips / ((t1 - t0) + (t2 - t1) / frac + (t3 - t2) / frac), 2),
           "extract_audio_s_per_s": round(180.0 * a.clips / (t1 - t0), 1), "decoder_tokens_per_s": round(ntok / (t2 - t1), 1),
           "decode_stage_alg_bytes": dec_bytes, "decode_stage_frac": round(dec_bytes / (t2 - t1) / 8e12, 4),
           "tokens_sha256": hashlib.sha256(np.concatenate([r[0] for r in res]).astype(np.int32).tobytes()).hexdigest()[:16],
           "build_id": _lib.lib().etd_build_id().decode()}
    if a.stamp:
        for d in decs:
            d.stamp(True, skip_steps=4 * (a.bar_tokens - 1))
            d.stats_reset()
        pipe.decode_stage(conds, max_bars=a.max_bars or 12)
        torch.cuda.synchronize(dev)
        s2 = [d.stats() for d in decs]
        for d in decs:
            d.stamp(False)
        launches = sum(s["stamped_launches"] for s in s2); secs = sum(s["stamped_seconds"] for s in s2); byts = sum(s["stamped_alg_bytes"] for s in s2)
        if launches and secs > 0:
            out["roofline"] = {"kernel": "k_dattn<float> (fp32 K/V rows)", "bound": "hbm", "achieved": round(byts / secs / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(byts / secs / 8e12, 4), "launches": int(launches), "avg_launch_ms": round(1e3 * secs / launches, 5), "alg_bytes_per_launch": byts / launches}
    if a.events:
        _lib.prof_reset(); _lib.prof_enable(True)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(pipe.ex_streams[0]):
            pipe.conditions_of(wavs[0], 0, 0)
        pipe.ex_streams[0].synchronize()
        pipe.decode_stage(conds, max_bars=6, one_at_a_time=True)
        torch.cuda.synchronize(dev)
        _lib.prof_enable(False)
        prof = _lib.prof_report()
        out["events"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"], "us_per_launch": round(1e3 * v["ms"] / max(1, v["launches"]), 2)}
                         for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    for d in reversed(decs):
        d.close()
    pipe.close()
    for e in exs:
        e.close()
    s = json.dumps(out)
    print(s)
    if a.json:
        Path(a.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

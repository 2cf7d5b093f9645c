#!/usr/bin/env python3
"""Bring-up diagnostic (GPU box): HIP decoder vs oracle logits and golden greedy ids; rough timings."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from etude_amd import synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402
from etude_amd.vocab import Vocab  # noqa: E402
from tests._util import TINY_DEC, TINY_DEC_KW  # noqa: E402

G = ROOT / "tests" / "golden"


def vocab():
    v = Vocab()
    vj = synth.vocab_json()
    v.token_to_id = vj["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def run(name, dims, seed, kw, limit):
    g = np.load(G / f"{name}.npz")
    cfg = EtudeDecoderConfig(**synth.decoder_dims(**dims))
    sd = synth.decoder_state_dict(seed, dims, **kw)
    v = vocab()
    for prec in ("fp32", "f16"):
        dec = EtudeDecoder(cfg, sd, "cuda", precision=prec, max_streams=4)
        a4 = np.stack([g["prompt_overlap"][0], g["prompt_polyphony"][0], g["prompt_sustain"][0], g["prompt_rhythm"][0]])
        lg = dec.prefill_logits(g["prompt_ids"][0], g["prompt_cls"][0], a4)
        d = np.abs(lg - g["logits"])
        print(f"{name} {prec}: logits max|d|={d.max():.3e} mean|d|={d.mean():.3e} ref_absmax={np.abs(g['logits']).max():.2f} "
              f"argmax agree={(lg.argmax(-1) == g['logits'].argmax(-1)).mean():.3f}", flush=True)
        n_bars = int(g["n_bars"])
        bars = synth.song_bars(seed=3, n_bars=n_bars)
        j = 0
        while f"gen_ids_{j}" in g:
            a = synth.attrs(*[int(x) for x in g[f"gen_attrs_{j}"]])
            t0 = time.time()
            out = dec.generate_ids(v, bars, [a] * n_bars, max_bar_token_limit=limit, temperature=0.0)
            dt = time.time() - t0
            flat = [t for b in out for t in b]
            ref = g[f"gen_ids_{j}"].tolist()
            same = flat == ref
            first = next((i for i, (x, y) in enumerate(zip(flat, ref)) if x != y), min(len(flat), len(ref)))
            print(f"   attrs {a['polyphony_bin'],a['rhythm_intensity_bin'],a['sustain_bin']}: ids equal={same} "
                  f"(len {len(flat)} vs {len(ref)}, first diff at {first}) {len(flat)/dt:.0f} tok/s", flush=True)
            j += 1
        dec.close()


def throughput():
    cfg = EtudeDecoderConfig(**synth.decoder_dims())
    sd = synth.decoder_state_dict(1, {})
    v = vocab()
    for prec, S in (("fp32", 1), ("f16", 1), ("f16", 128), ("fp32", 128)):
        dec = EtudeDecoder(cfg, sd, "cuda", precision=prec, max_streams=S)
        jobs = []
        for s in range(S):
            bars = synth.song_bars(seed=100 + s, n_bars=3)
            jobs.append((bars, [synth.attrs(s % 3, (s // 3) % 3, (s // 9) % 3, 2)] * 3))
        st = {}
        torch.cuda.synchronize()
        t0 = time.time()
        dec.generate_many(jobs, v, max_bar_token_limit=64, stats=st)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"throughput {prec} streams={S}: {st['tokens']} tokens in {dt:.2f}s -> {st['tokens']/dt:.0f} tok/s ({st['steps']} steps)", flush=True)
        dec.close()


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    # tiny config (head_dim 16) is for pinning the oracle only; the kernels are built for head_dim 64
    run("decoder_full", {}, 1, {}, 48)
    throughput()

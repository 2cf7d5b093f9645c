#!/usr/bin/env python3
"""CPU only: what a two-term f16 split of every GEMM operand (hi = f16(s x), lo = f16(s x - hi); products hi*hi + hi*lo + lo*hi [+ lo*lo], fp32 accumulate --
the arithmetic of three v_mfma_f32_32x32x16_f16 per fp32 product tile) does to the decoder oracle's logits and greedy ids, next to the same split in bf16.
`oracle.neox`'s F.linear and its two attention matmuls are replaced by the emulation; everything else (LayerNorm, softmax, GELU, RoPE, residuals) stays fp32.

    python tools/diag_split_f16.py            # decoder_ctx.npz logits (T = 1024 / 3500, both weight sets), decoder_full / clip_ctx greedy ids (a few bars)
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import neox  # noqa: E402

MODE = {"dtype": None, "terms": 3, "scale": True}
_real_linear = F.linear
_real_matmul = torch.matmul


def _split(x, dtype, scale):
    if scale and dtype == torch.float16:
        m = float(x.abs().max())
        s = 2.0 ** (14 - int(np.floor(np.log2(m)))) if m > 0 else 1.0      # max |s x| in [2^14, 2^15)
    else:
        s = 1.0
    xs = x * s
    hi = xs.to(dtype).float()
    lo = (xs - hi).to(dtype).float()
    return hi, lo, s


def _mm(a, b):          # a [.., M, K] @ b [.., K, N] in the emulated arithmetic
    dt = MODE["dtype"]
    if dt is None:
        return _real_matmul(a, b)
    ah, al, sa = _split(a, dt, MODE["scale"])
    bh, bl, sb = _split(b, dt, MODE["scale"])
    acc = _real_matmul(ah, bh)
    if MODE["terms"] >= 2:
        acc = acc + _real_matmul(ah, bl)
    if MODE["terms"] >= 3:
        acc = acc + _real_matmul(al, bh)
    if MODE["terms"] >= 4:
        acc = acc + _real_matmul(al, bl)
    return acc / (sa * sb)


def linear(x, w, b=None):
    y = _mm(x, w.t())
    return y if b is None else y + b


class _TorchProxy:
    """torch with matmul replaced (neox.transformer calls torch.matmul for QK^T and PV)"""
    def __getattr__(self, k):
        return _mm if k == "matmul" else getattr(torch, k)


class _Torch64Proxy:
    """torch for the fp64 reference pass: the oracle's softmax pins dtype=torch.float32"""
    def __getattr__(self, k):
        return torch.float64 if k == "float32" else getattr(torch, k)


class _FProxy:
    def __getattr__(self, k):
        return linear if k == "linear" else getattr(F, k)


def main():
    neox.F = _FProxy()
    neox.torch = _TorchProxy()
    G = ROOT / "tests" / "golden"
    from etude_amd import synth
    d = neox.NeoxDims()
    z = np.load(G / "decoder_ctx.npz")
    print("decoder_ctx.npz keys:", [k for k in z.files][:20])
    modes = [("fp32 (oracle)", None, 1), ("f16 x3", torch.float16, 3), ("f16 x4", torch.float16, 4), ("f16 x3 unscaled", "f16ns", 3), ("bf16 x3", torch.bfloat16, 3), ("bf16 x4", torch.bfloat16, 4), ("f16 x1", torch.float16, 1), ("f16 x1 unscaled", "f16ns1", 1), ("bf16 x1", torch.bfloat16, 1)]
    for wname, sdf in (("benchmark weights", lambda: synth.decoder_state_dict(1, {})), ("context weights", lambda: synth.decoder_state_dict_ctx(1))):
        sd = {k: torch.from_numpy(v) for k, v in sdf().items()}
        for T in (1024,):
            rng = np.random.default_rng(11)
            ids = torch.from_numpy(rng.integers(6, 154, T))[None]; cls = torch.from_numpy(rng.integers(1, 3, T))[None]
            at = {k: torch.from_numpy(rng.integers(0, 3, T))[None] for k in ("pitch_overlap", "polyphony", "note_sustain", "rhythm_intensity")}
            ref = None
            for name, dt, terms in modes:
                MODE.update(dtype=(torch.float16 if dt in ("f16ns", "f16ns1") else dt), terms=terms, scale=(dt not in ("f16ns", "f16ns1")))
                lg, _ = neox.forward_logits(sd, d, ids, cls, at)
                lg = lg[0].double().numpy()
                if ref is None:
                    # fp64 reference of the same forward
                    MODE.update(dtype=None)
                    sd64 = {k: v.double() for k, v in sd.items()}
                    neox.F = F; neox.torch = _Torch64Proxy()
                    lg64, _ = neox.forward_logits(sd64, d, ids, cls, at)
                    neox.F = _FProxy(); neox.torch = _TorchProxy()
                    ref = lg64[0].numpy()
                    top2 = np.sort(ref, -1)[:, -2:]
                    print(f"[{wname}, T = {T}] logits max |.| {np.abs(ref).max():.2f}; top-2 gap: min {np.min(top2[:, 1] - top2[:, 0]):.2e}, 1st percentile {np.percentile(top2[:, 1] - top2[:, 0], 1):.2e}")
                e = np.abs(lg - ref)
                flips = int((lg.argmax(-1) != ref.argmax(-1)).sum())
                print(f"   {name:18s} max err vs fp64 {e.max():.3e}  rms {np.sqrt((e ** 2).mean()):.3e}  argmax flips {flips} / {T}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Which bf16 rounding of the extractor's EncoderLayer owns its error?  (CPU, torch fp32 + emulated bf16 roundings; no GPU, no library.)

The bf16 extractor sits 1.2-1.3 % rms off the fp32 reference after EVERY encoder layer (profiles/r02_error_budget.txt); the layer's LayerNorms renormalise, so
the figure does not grow from layer to layer -- it is made inside one layer.  This script runs encoder layer 0 of the oracle (oracle/hft.py, amt_apc.py:244-259)
on real embedded frames of a synthetic window and rounds to bf16 exactly where csrc/ext_fused.hip's k_enc_layer does -- one site at a time, then cumulatively:

    W      weights                                   X     the layer input (operand AND residual)
    Q, K   projected queries (pre-scaled) / keys     V     projected values
    P      softmax numerators exp(s - max)           O     normalised attention output (operand of fc_o)
    X1     LayerNorm output (operand of the feed-forward block AND its residual)
    H      hidden activations relu(fc_1)             Y     the layer output

Prints rms(err) / rms(ref) and max|err| of the layer output per site.  usage: python tools/diag_rounding_budget.py [n_frames=16]"""
import math
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import synth  # noqa: E402
from oracle import hft  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def layer(sd, pfx, x, sites, n_heads=4):
    r = lambda name, v: bf(v) if name in sites else v      # noqa: E731
    w = lambda k: r("W", sd[k])                            # noqa: E731
    lin = lambda p, v: F.linear(v, w(p + ".weight"), sd[p + ".bias"])      # noqa: E731
    ln = lambda v: F.layer_norm(v, (v.shape[-1],), sd[pfx + ".layer_norm.weight"], sd[pfx + ".layer_norm.bias"], 1e-5)      # noqa: E731
    x = r("X", x)
    B, hid = x.shape[0], x.shape[-1]
    hd = hid // n_heads
    sa = pfx + ".self_attention"
    split = lambda v: v.view(B, -1, n_heads, hd).permute(0, 2, 1, 3)      # noqa: E731
    Q = r("Q", split(lin(sa + ".fc_q", x)) / math.sqrt(hd))
    K = r("K", split(lin(sa + ".fc_k", x)))
    V = r("V", split(lin(sa + ".fc_v", x)))
    s = torch.matmul(Q, K.permute(0, 1, 3, 2))
    p = torch.exp(s - s.max(-1, keepdim=True).values)
    den = p.sum(-1, keepdim=True)                     # the kernel sums the fp32 numerators, then rounds them for the PV product
    o = torch.matmul(r("P", p), V) / den
    o = r("O", o).permute(0, 2, 1, 3).contiguous().view(B, -1, hid)
    x1 = r("X1", ln(x + lin(sa + ".fc_o", o)))
    ff = pfx + ".positionwise_feedforward"
    hdn = r("H", torch.relu(lin(ff + ".fc_1", x1)))
    return r("Y", ln(x1 + lin(ff + ".fc_2", hdn)))


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(8)
    sd = {k: torch.from_numpy(v) for k, v in synth.extractor_state_dict(0).items()}
    d = hft.HftDims()
    x = torch.from_numpy(synth.window_features(5, 1))
    taps = {}
    with torch.no_grad():
        # the embedded frames (amt_apc.py:74-99) of the first nfr frames of the window
        sub = x[:, :, : nfr + 2 * d.n_margin]
        d2 = hft.HftDims(n_frame=nfr)
        hft.encoder_forward(sd, sub, d2, taps)
        emb = taps["embed"]                                  # [nfr, 256, 256]
        pfx = "encoder.layers_freq.0"
        ref = layer(sd, pfx, emb, set())
        assert torch.allclose(ref, taps["enc0"], atol=2e-4), float((ref - taps["enc0"]).abs().max())
        rms = lambda e: float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())      # noqa: E731
        sites = ["W", "X", "Q", "K", "V", "P", "O", "X1", "H", "Y"]
        print(f"encoder layer 0 on {nfr} frames x 256 bins; output rms {float(ref.pow(2).mean().sqrt()):.3f}")
        print("site            rms err / rms ref    max |err|")
        for sname in sites:
            y = layer(sd, pfx, emb, {sname})
            print(f"only {sname:3s}        {rms(y - ref):.4e}          {float((y - ref).abs().max()):.3e}")
        cum = set()
        for sname in sites:
            cum.add(sname)
            y = layer(sd, pfx, emb, cum)
            print(f"+ {sname:3s} (cum.)     {rms(y - ref):.4e}          {float((y - ref).abs().max()):.3e}")
        for drop in (["Q", "K"], ["P"], ["X", "X1"], ["Q", "K", "P"], ["W"]):
            y = layer(sd, pfx, emb, set(sites) - set(drop))
            print(f"all but {'+'.join(drop):8s} {rms(y - ref):.4e}          {float((y - ref).abs().max()):.3e}")
        # the same for layer 1, whose input is a LayerNorm output (unit scale): exact (fp32) input, all roundings on
        pf1 = "encoder.layers_freq.1"
        ref1 = layer(sd, pf1, taps["enc0"], set())
        y1 = layer(sd, pf1, taps["enc0"], set(sites))
        Q1 = F.linear(taps["enc0"], sd[pf1 + ".self_attention.fc_q.weight"], sd[pf1 + ".self_attention.fc_q.bias"]).view(nfr, -1, 4, 64).permute(0, 2, 1, 3) / 8.0
        K1 = F.linear(taps["enc0"], sd[pf1 + ".self_attention.fc_k.weight"], sd[pf1 + ".self_attention.fc_k.bias"]).view(nfr, -1, 4, 64).permute(0, 2, 1, 3)
        print(f"layer 1 from its EXACT input, every rounding on: rms {float((y1 - ref1).pow(2).mean().sqrt() / ref1.pow(2).mean().sqrt()):.4e}, max {float((y1 - ref1).abs().max()):.3e}"
              f" (its scores: std {float(torch.matmul(Q1, K1.permute(0, 1, 3, 2)).std()):.2f})")
        # statistics that explain the Q / K figure: the spread of the scores a rounding error is exponentiated through
        sa = pfx + ".self_attention"
        Q = F.linear(emb, sd[sa + ".fc_q.weight"], sd[sa + ".fc_q.bias"]).view(nfr, -1, 4, 64).permute(0, 2, 1, 3) / 8.0
        K = F.linear(emb, sd[sa + ".fc_k.weight"], sd[sa + ".fc_k.bias"]).view(nfr, -1, 4, 64).permute(0, 2, 1, 3)
        s = torch.matmul(Q, K.permute(0, 1, 3, 2))
        sb = torch.matmul(bf(Q), bf(K).permute(0, 1, 3, 2))
        print(f"scores: std {float(s.std()):.2f}, |max| {float(s.abs().max()):.1f}; bf16 Q, K move a score by rms {float((sb - s).pow(2).mean().sqrt()):.4f} (max {float((sb - s).abs().max()):.3f})"
              " -> that much RELATIVE error in every softmax numerator")


if __name__ == "__main__":
    main()

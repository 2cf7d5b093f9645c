"""Oracle: hFT-Transformer (AMT-APC) forward pass, torch-CPU fp32, functional form.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference line by line, but takes a flat state dict with the reference's
own key names (``encoder.*`` / ``decoder.*`` as produced by ``_Spec2MIDI``,
etude/data/extractor.py:34-42):

  * encoder           etude/models/amt_apc.py:74-120
  * encoder layer     etude/models/amt_apc.py:236-259 (post-LN, ONE shared LayerNorm)
  * decoder (freq)    etude/models/amt_apc.py:159-197, layers :261-320
  * decoder (time)    etude/models/amt_apc.py:199-230
  * MHA               etude/models/amt_apc.py:322-374 (energy / sqrt(head_dim), softmax)
  * FFN               etude/models/amt_apc.py:376-392 (ReLU)
  * _transcript       etude/data/extractor.py:199-253
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class HftDims:
    """Shape parameters (defaults = etude/config/schema.py:68-112)."""
    n_margin: int = 32
    n_frame: int = 512
    n_bin: int = 256
    cnn_channel: int = 4
    cnn_kernel: int = 5
    hid_dim: int = 256
    pf_dim: int = 512
    n_heads: int = 4
    n_layers_enc: int = 3
    n_layers_dec: int = 3
    n_note: int = 88
    n_velocity: int = 128

    @property
    def n_proc(self) -> int:
        return 2 * self.n_margin + 1

    @property
    def cnn_dim(self) -> int:
        return self.cnn_channel * (self.n_proc - (self.cnn_kernel - 1))


def _lin(sd, pfx, x):
    return F.linear(x, sd[pfx + ".weight"], sd[pfx + ".bias"])


def _ln(sd, pfx, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + ".weight"], sd[pfx + ".bias"], 1e-5)


def mha(sd, pfx, q_in, k_in, v_in, n_heads):
    """amt_apc.py:336-374."""
    B = q_in.shape[0]
    hid = q_in.shape[-1]
    hd = hid // n_heads
    Q = _lin(sd, pfx + ".fc_q", q_in).view(B, -1, n_heads, hd).permute(0, 2, 1, 3)
    K = _lin(sd, pfx + ".fc_k", k_in).view(B, -1, n_heads, hd).permute(0, 2, 1, 3)
    V = _lin(sd, pfx + ".fc_v", v_in).view(B, -1, n_heads, hd).permute(0, 2, 1, 3)
    energy = torch.matmul(Q, K.permute(0, 1, 3, 2)) / math.sqrt(hd)
    att = torch.softmax(energy, dim=-1)
    x = torch.matmul(att, V).permute(0, 2, 1, 3).contiguous().view(B, -1, hid)
    return _lin(sd, pfx + ".fc_o", x), att


def ffn(sd, pfx, x):
    """amt_apc.py:383-392."""
    return _lin(sd, pfx + ".fc_2", torch.relu(_lin(sd, pfx + ".fc_1", x)))


def encoder_layer(sd, pfx, x, n_heads):
    """amt_apc.py:244-259 — note the single shared layer_norm."""
    a, _ = mha(sd, pfx + ".self_attention", x, x, x, n_heads)
    x = _ln(sd, pfx + ".layer_norm", x + a)
    x = _ln(sd, pfx + ".layer_norm", x + ffn(sd, pfx + ".positionwise_feedforward", x))
    return x


def decoder_layer_zero(sd, pfx, enc, trg, n_heads):
    """amt_apc.py:269-286."""
    a, att = mha(sd, pfx + ".encoder_attention", trg, enc, enc, n_heads)
    trg = _ln(sd, pfx + ".layer_norm", trg + a)
    trg = _ln(sd, pfx + ".layer_norm", trg + ffn(sd, pfx + ".positionwise_feedforward", trg))
    return trg, att


def decoder_layer(sd, pfx, enc, trg, n_heads):
    """amt_apc.py:297-320."""
    a, _ = mha(sd, pfx + ".self_attention", trg, trg, trg, n_heads)
    trg = _ln(sd, pfx + ".layer_norm", trg + a)
    a, att = mha(sd, pfx + ".encoder_attention", trg, enc, enc, n_heads)
    trg = _ln(sd, pfx + ".layer_norm", trg + a)
    trg = _ln(sd, pfx + ".layer_norm", trg + ffn(sd, pfx + ".positionwise_feedforward", trg))
    return trg, att


def encoder_forward(sd: Dict[str, torch.Tensor], spec_in: torch.Tensor, d: HftDims,
                    taps: dict | None = None) -> torch.Tensor:
    """amt_apc.py:74-120.  spec_in [B, n_bin, n_frame + 2*n_margin] -> [B, n_frame, n_bin, hid]."""
    B = spec_in.shape[0]
    spec = spec_in.unfold(2, d.n_proc, 1).permute(0, 2, 1, 3).contiguous()
    spec_cnn = spec.reshape(B * d.n_frame, d.n_bin, d.n_proc).unsqueeze(1)
    spec_cnn = F.conv2d(spec_cnn, sd["encoder.conv.weight"], sd["encoder.conv.bias"]).permute(0, 2, 1, 3).contiguous()
    spec_cnn_freq = spec_cnn.reshape(B * d.n_frame, d.n_bin, d.cnn_dim)
    emb = _lin(sd, "encoder.tok_embedding_freq", spec_cnn_freq)
    x = emb * math.sqrt(d.hid_dim) + sd["encoder.pos_embedding_freq.weight"][None, :, :]
    if taps is not None:
        taps["embed"] = x
    for i in range(d.n_layers_enc):
        x = encoder_layer(sd, f"encoder.layers_freq.{i}", x, d.n_heads)
        if taps is not None:
            taps[f"enc{i}"] = x
    return x.reshape(B, d.n_frame, d.n_bin, d.hid_dim)


def decoder_forward(sd, enc_spec, d: HftDims, taps: dict | None = None, want_attention: bool = False):
    """amt_apc.py:159-230.  Returns the reference's 9-tuple (attention None unless asked)."""
    B = enc_spec.shape[0]
    enc = enc_spec.reshape(B * d.n_frame, d.n_bin, d.hid_dim)
    midi = sd["decoder.pos_embedding_freq.weight"][None].expand(B * d.n_frame, -1, -1)
    midi, att = decoder_layer_zero(sd, "decoder.layer_zero_freq", enc, midi, d.n_heads)
    if taps is not None:
        taps["dec0"] = midi
    for i in range(d.n_layers_dec - 1):
        midi, att = decoder_layer(sd, f"decoder.layers_freq.{i}", enc, midi, d.n_heads)
        if taps is not None:
            taps[f"dec{i + 1}"] = midi
    attention = att.reshape(B, d.n_frame, *att.shape[1:]) if want_attention else None

    shp = [B, d.n_frame, d.n_note]
    on_f = torch.sigmoid(_lin(sd, "decoder.fc_onset_freq", midi).reshape(shp))
    off_f = torch.sigmoid(_lin(sd, "decoder.fc_offset_freq", midi).reshape(shp))
    mpe_f = torch.sigmoid(_lin(sd, "decoder.fc_mpe_freq", midi).reshape(shp))
    vel_f = _lin(sd, "decoder.fc_velocity_freq", midi).reshape(shp + [d.n_velocity])

    t = midi.reshape(B, d.n_frame, d.n_note, d.hid_dim).permute(0, 2, 1, 3).contiguous()
    t = t.reshape(B * d.n_note, d.n_frame, d.hid_dim)
    t = t * math.sqrt(d.hid_dim) + sd["decoder.pos_embedding_time.weight"][None, :, :]
    if taps is not None:
        taps["time_in"] = t
    for i in range(d.n_layers_dec):
        t = encoder_layer(sd, f"decoder.layers_time.{i}", t, d.n_heads)
        if taps is not None:
            taps[f"time{i}"] = t

    shp_t = [B, d.n_note, d.n_frame]
    on_logit = _lin(sd, "decoder.fc_onset_time", t).reshape(shp_t).permute(0, 2, 1).contiguous()
    off_logit = _lin(sd, "decoder.fc_offset_time", t).reshape(shp_t).permute(0, 2, 1).contiguous()
    mpe_logit = _lin(sd, "decoder.fc_mpe_time", t).reshape(shp_t).permute(0, 2, 1).contiguous()
    if taps is not None:
        taps["onset_logit"], taps["offset_logit"], taps["mpe_logit"] = on_logit, off_logit, mpe_logit
    vel_t = _lin(sd, "decoder.fc_velocity_time", t).reshape(shp_t + [d.n_velocity]).permute(0, 2, 1, 3).contiguous()
    return (on_f, off_f, mpe_f, vel_f, attention,
            torch.sigmoid(on_logit), torch.sigmoid(off_logit), torch.sigmoid(mpe_logit), vel_t)


@torch.no_grad()
def model_forward(sd, spec_in, d: HftDims, taps: dict | None = None, want_attention: bool = False):
    """Model_SPEC2MIDI.forward (amt_apc.py:29-49) via _Spec2MIDI.forward (extractor.py:53-56)."""
    return decoder_forward(sd, encoder_forward(sd, spec_in, d, taps), d, taps, want_attention)


@torch.no_grad()
def transcript(sd, a_feature: np.ndarray, d: HftDims, min_value: float = -18.0,
               return_vel_logits: bool = False):
    """extractor.py:199-253, mode="combination".  a_feature [T, n_bin] fp32.

    Returns the 8 arrays (onset/offset/mpe/velocity x A,B); optionally also the B velocity
    logits [T+len_s, n_note, n_velocity] so that a test can accept an argmax that differs
    only on a numerical near-tie.
    """
    a_feature = np.array(a_feature, dtype=np.float32)
    T = a_feature.shape[0]
    nf = d.n_frame
    len_s = int(np.ceil(T / nf) * nf) - T
    a_in = torch.from_numpy(np.concatenate([
        np.full([d.n_margin, d.n_bin], min_value, np.float32), a_feature,
        np.full([len_s + d.n_margin, d.n_bin], min_value, np.float32)], axis=0))
    outs = [np.zeros((T + len_s, d.n_note), np.float32) for _ in range(3)] + [np.zeros((T + len_s, d.n_note), np.int8)]
    outs = outs + [np.zeros((T + len_s, d.n_note), np.float32) for _ in range(3)] + [np.zeros((T + len_s, d.n_note), np.int8)]
    vel_logits = np.zeros((T + len_s, d.n_note, d.n_velocity), np.float32) if return_vel_logits else None
    for i in range(0, T, nf):
        spec = a_in[i:i + 2 * d.n_margin + nf].T.unsqueeze(0)
        r = model_forward(sd, spec, d)
        outs[0][i:i + nf] = r[0][0].numpy()
        outs[1][i:i + nf] = r[1][0].numpy()
        outs[2][i:i + nf] = r[2][0].numpy()
        outs[3][i:i + nf] = r[3][0].argmax(2).numpy()
        outs[4][i:i + nf] = r[5][0].numpy()
        outs[5][i:i + nf] = r[6][0].numpy()
        outs[6][i:i + nf] = r[7][0].numpy()
        outs[7][i:i + nf] = r[8][0].argmax(2).numpy()
        if vel_logits is not None:
            vel_logits[i:i + nf] = r[8][0].numpy()
    if return_vel_logits:
        return tuple(outs), vel_logits
    return tuple(outs)


@torch.no_grad()
def transcript_stride(sd, a_feature: np.ndarray, d: HftDims, n_offset: int, min_value: float = -80.0):
    """HFT_Transformer._transcript_stride, etude/models/hft_transformer.py:282-460, mode="combination": windows start every
    n_frame/2 frames; rows [n_offset, n_offset + n_frame/2) of each window's outputs are kept.  Returns the 8 arrays in the
    reference's order (onset/offset/mpe/velocity A, then B)."""
    a_feature = np.array(a_feature, dtype=np.float32)
    T = a_feature.shape[0]
    nf, half = d.n_frame, d.n_frame // 2
    tmp_len = T + 2 * d.n_margin + half
    len_s = int(np.ceil(tmp_len / half) * half) - tmp_len
    a_in = torch.from_numpy(np.concatenate([
        np.full([d.n_margin + n_offset, d.n_bin], min_value, np.float32), a_feature,
        np.full([len_s + d.n_margin + (half - n_offset), d.n_bin], min_value, np.float32)], axis=0))
    mk = lambda dt: np.zeros((T + len_s, d.n_note), dt)            # noqa: E731
    outs = [mk(np.float32), mk(np.float32), mk(np.float32), mk(np.int8), mk(np.float32), mk(np.float32), mk(np.float32), mk(np.int8)]
    for i in range(0, T, half):
        spec = a_in[i:i + 2 * d.n_margin + nf].T.unsqueeze(0)
        r = model_forward(sd, spec, d)
        sl = slice(n_offset, n_offset + half)
        outs[0][i:i + half] = r[0][0][sl].numpy()
        outs[1][i:i + half] = r[1][0][sl].numpy()
        outs[2][i:i + half] = r[2][0][sl].numpy()
        outs[3][i:i + half] = r[3][0][sl].argmax(2).numpy()
        outs[4][i:i + half] = r[5][0][sl].numpy()
        outs[5][i:i + half] = r[6][0][sl].numpy()
        outs[6][i:i + half] = r[7][0][sl].numpy()
        outs[7][i:i + half] = r[8][0][sl].argmax(2).numpy()
    return tuple(outs)

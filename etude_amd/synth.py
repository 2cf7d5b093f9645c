"""Deterministic synthetic checkpoints, vocabularies and workloads.

The reference's checkpoints are release downloads (README.md:86-94) and there is no network, so
every benchmark, smoke test and golden vector in this repo runs on seeded synthetic weights laid
out exactly like the reference's files:

  * extractor: flat state dict with the ``encoder.*`` / ``decoder.*`` keys that
    ``_Spec2MIDI`` produces (etude/data/extractor.py:34-42, 165 tensors at the default config);
  * decoder:   ``EtudeDecoder.state_dict()`` keys (etude/models/etude_decoder.py:98-122) incl. the
    unused ``transformer.embed_in.weight`` so that the reference's strict load accepts it;
  * vocab:     ``{"token_to_id": ..., "special_tokens": ...}`` (etude/data/vocab.py:148-157).

Only numpy's ``default_rng`` (PCG64, stable across versions) is used, so the same seed gives the
same tensors in the build container and on the GPU box.
"""
from __future__ import annotations

import json
import math
from typing import Dict, List, Tuple

import numpy as np

# ----------------------------------------------------------------------------- extractor


def extractor_dims(**over) -> Dict[str, int]:
    """Default shape parameters = etude/config/schema.py:68-112."""
    d = dict(n_margin=32, n_frame=512, n_bin=256, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512,
             n_heads=4, n_layers_enc=3, n_layers_dec=3, n_note=88, n_velocity=128)
    d.update(over)
    return d


def _lin(rng, out_f, in_f, gain=1.0):
    b = 1.0 / math.sqrt(in_f)
    return (rng.uniform(-b, b, (out_f, in_f)) * gain).astype(np.float32), (rng.uniform(-b, b, (out_f,)) * gain).astype(np.float32)


def _ln(rng, n):
    return (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32), (0.1 * rng.standard_normal(n)).astype(np.float32)


def _mha(rng, sd, pfx, hid):
    for n in ("fc_q", "fc_k", "fc_v", "fc_o"):
        sd[f"{pfx}.{n}.weight"], sd[f"{pfx}.{n}.bias"] = _lin(rng, hid, hid, gain=2.0 if n in ("fc_q", "fc_k") else 1.0)


def _ffn(rng, sd, pfx, hid, pf):
    sd[f"{pfx}.fc_1.weight"], sd[f"{pfx}.fc_1.bias"] = _lin(rng, pf, hid)
    sd[f"{pfx}.fc_2.weight"], sd[f"{pfx}.fc_2.bias"] = _lin(rng, hid, pf)


def extractor_state_dict(seed: int = 0, dims: Dict[str, int] | None = None, head_gain: float = 4.0, emb_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """emb_gain scales encoder.tok_embedding_freq AFTER every tensor has been drawn (the other tensors do not depend on it).  With emb_gain = 1 (every golden
    of rounds 1-3, the benchmark) the first encoder layer sees x = 16 emb + pos with |x| ~ 75: its attention scores have a standard deviation of ~3 700, the
    softmax is a hard argmax, and ANY rounding of the layer input or of K flips winners -- that one layer makes the 1.2 % rms the bf16 extractor's ENCODER taps
    show on these weights (tools/diag_rounding_budget.py, profiles/r04_rounding_budget.txt); a layer fed a LayerNorm output adds 0.3 %.  (The final probabilities
    are another matter: they are as close to the reference with either checkpoint, tests/test_gpu_extractor.py.)
    `extractor_state_dict_cal` (emb_gain = 1 / 48: scores of layer 0 with sigma ~ 3, like the layers behind it) is the well-conditioned variant of the same weights."""
    d = extractor_dims(**(dims or {}))
    rng = np.random.default_rng(seed)
    hid, pf = d["hid_dim"], d["pf_dim"]
    n_proc = 2 * d["n_margin"] + 1
    cnn_dim = d["cnn_channel"] * (n_proc - (d["cnn_kernel"] - 1))
    sd: Dict[str, np.ndarray] = {}
    b = 1.0 / math.sqrt(d["cnn_kernel"])
    sd["encoder.conv.weight"] = rng.uniform(-b, b, (d["cnn_channel"], 1, 1, d["cnn_kernel"])).astype(np.float32)
    sd["encoder.conv.bias"] = rng.uniform(-b, b, (d["cnn_channel"],)).astype(np.float32)
    sd["encoder.tok_embedding_freq.weight"], sd["encoder.tok_embedding_freq.bias"] = _lin(rng, hid, cnn_dim)
    sd["encoder.pos_embedding_freq.weight"] = rng.standard_normal((d["n_bin"], hid)).astype(np.float32)
    for i in range(d["n_layers_enc"]):
        p = f"encoder.layers_freq.{i}"
        sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"] = _ln(rng, hid)
        _mha(rng, sd, p + ".self_attention", hid)
        _ffn(rng, sd, p + ".positionwise_feedforward", hid, pf)
    sd["decoder.pos_embedding_freq.weight"] = rng.standard_normal((d["n_note"], hid)).astype(np.float32)
    p = "decoder.layer_zero_freq"
    sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"] = _ln(rng, hid)
    _mha(rng, sd, p + ".encoder_attention", hid)
    _ffn(rng, sd, p + ".positionwise_feedforward", hid, pf)
    for i in range(d["n_layers_dec"] - 1):
        p = f"decoder.layers_freq.{i}"
        sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"] = _ln(rng, hid)
        _mha(rng, sd, p + ".self_attention", hid)
        _mha(rng, sd, p + ".encoder_attention", hid)
        _ffn(rng, sd, p + ".positionwise_feedforward", hid, pf)
    for n, o in (("onset", 1), ("offset", 1), ("mpe", 1), ("velocity", d["n_velocity"])):
        sd[f"decoder.fc_{n}_freq.weight"], sd[f"decoder.fc_{n}_freq.bias"] = _lin(rng, o, hid, gain=head_gain)
    sd["decoder.pos_embedding_time.weight"] = rng.standard_normal((d["n_frame"], hid)).astype(np.float32)
    for i in range(d["n_layers_dec"]):
        p = f"decoder.layers_time.{i}"
        sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"] = _ln(rng, hid)
        _mha(rng, sd, p + ".self_attention", hid)
        _ffn(rng, sd, p + ".positionwise_feedforward", hid, pf)
    for n, o in (("onset", 1), ("offset", 1), ("mpe", 1), ("velocity", d["n_velocity"])):
        sd[f"decoder.fc_{n}_time.weight"], sd[f"decoder.fc_{n}_time.bias"] = _lin(rng, o, hid, gain=head_gain)
    if emb_gain != 1.0:
        sd["encoder.tok_embedding_freq.weight"] = (sd["encoder.tok_embedding_freq.weight"] * np.float32(emb_gain)).astype(np.float32)
        sd["encoder.tok_embedding_freq.bias"] = (sd["encoder.tok_embedding_freq.bias"] * np.float32(emb_gain)).astype(np.float32)
    return sd


def extractor_state_dict_cal(seed: int = 0, dims: Dict[str, int] | None = None, head_gain: float = 4.0) -> Dict[str, np.ndarray]:
    """The extractor checkpoint with a well-conditioned first layer (see extractor_state_dict): tests/golden/hft_full_cal.npz, tests/test_gpu_extractor.py."""
    return extractor_state_dict(seed, dims, head_gain, emb_gain=1.0 / 48.0)


def window_features(seed: int, n_windows: int, n_bin: int = 256, n_in: int = 576) -> np.ndarray:
    """BASELINE config 3 input: features ~ N(-8, 2^2) clipped to [-18, 5], [B, n_bin, n_in] fp32."""
    rng = np.random.default_rng(seed)
    return np.clip(rng.normal(-8.0, 2.0, (n_windows, n_bin, n_in)), -18.0, 5.0).astype(np.float32)


def clip_audio(seed: int = 1234, seconds: float = 180.0, sr: int = 44100) -> np.ndarray:
    """BASELINE config 2 audio: stereo fp32 [2, L]; 8 random sinusoids (MIDI 40-90) per 0.25 s
    segment + -30 dB white noise, peak 0.5."""
    rng = np.random.default_rng(seed)
    L = int(round(seconds * sr))
    seg = int(0.25 * sr)
    t = np.arange(seg, dtype=np.float64) / sr
    out = np.zeros((2, L), np.float32)
    for s in range(0, L, seg):
        n = min(seg, L - s)
        midi = rng.integers(40, 91, size=8)
        f = 440.0 * 2.0 ** ((midi - 69) / 12.0)
        ph = rng.uniform(0, 2 * np.pi, size=8)
        env = np.exp(-3.0 * t[:n])
        x = (np.sin(2 * np.pi * f[:, None] * t[None, :n] + ph[:, None]) * env[None]).sum(0) / 8.0
        pan = rng.uniform(0.3, 0.7)
        out[0, s:s + n] = (x * pan * 2).astype(np.float32)
        out[1, s:s + n] = (x * (1 - pan) * 2).astype(np.float32)
    out += (10 ** (-30 / 20) * rng.standard_normal(out.shape)).astype(np.float32) * 0.5
    out *= 0.5 / max(1e-9, float(np.abs(out).max()))
    return out


def clip_audio_device(seed: int, seconds: float = 180.0, sr: int = 44100, device="cuda"):
    """`clip_audio`'s construction evaluated on the GPU (64 distinct 3-minute clips in a second instead of minutes of numpy: bench.py's batch, SURVEY 8(d) config 5:
    "64 clips as config 2 with seeds 0..63"): the same per-segment draws from the same numpy generator (8 MIDI pitches 40-90, phases, pan per 0.25 s segment, decaying
    envelope), sinusoids in float64 on the device, the -30 dB white noise from a seeded device generator (so the samples are NOT bit-equal to `clip_audio(seed)`;
    the goldens keep using that one).  Returns a float32 [2, L] tensor on `device`, peak 0.5."""
    import torch
    dev = torch.device(device)
    rng = np.random.default_rng(seed)
    L = int(round(seconds * sr))
    seg = int(0.25 * sr)
    nseg = (L + seg - 1) // seg
    midi = np.empty((nseg, 8)); ph = np.empty((nseg, 8)); pan = np.empty(nseg)
    for i in range(nseg):                        # (the draw order of clip_audio: pitches, phases, pan per segment)
        midi[i] = rng.integers(40, 91, size=8)
        ph[i] = rng.uniform(0, 2 * np.pi, size=8)
        pan[i] = rng.uniform(0.3, 0.7)
    f = torch.from_numpy(440.0 * 2.0 ** ((midi - 69) / 12.0)).to(dev)                  # [nseg, 8] float64
    pht = torch.from_numpy(ph).to(dev)
    pant = torch.from_numpy(pan).to(dev)
    t = torch.arange(seg, dtype=torch.float64, device=dev) / sr
    env = torch.exp(-3.0 * t)
    x = torch.zeros((nseg, seg), dtype=torch.float64, device=dev)
    for k in range(8):                           # one [nseg, seg] float64 plane at a time (a [nseg, 8, seg] tensor would be 0.5 GB)
        x += torch.sin(2 * np.pi * f[:, k:k + 1] * t[None, :] + pht[:, k:k + 1])
    x = (x * env[None, :] / 8.0)
    left = (x * (pant[:, None] * 2)).reshape(-1)[:L].float()
    right = (x * ((1 - pant[:, None]) * 2)).reshape(-1)[:L].float()
    out = torch.stack([left, right])
    g = torch.Generator(device=dev); g.manual_seed(int(seed))
    out += (10 ** (-30 / 20) * 0.5) * torch.randn(out.shape, generator=g, device=dev, dtype=torch.float32)
    out *= 0.5 / max(1e-9, float(out.abs().max()))
    return out.contiguous()


# ----------------------------------------------------------------------------- decoder


def decoder_dims(**over) -> Dict[str, int]:
    """Defaults = etude/config/schema.py:208-219 + the synthetic vocab size."""
    d = dict(vocab_size=154, hidden_size=512, num_hidden_layers=8, num_attention_heads=8, intermediate_size=2048,
             max_position_embeddings=1024, num_classes=3, pad_class_id=0, attribute_pad_id=0, pad_token_id=0,
             context_num_past_xy_pairs=4, num_attribute_bins=3, attribute_emb_dim=64, initializer_range=0.02)
    d.update(over)
    return d


def decoder_config_json(dims: Dict[str, int] | None = None) -> Dict:
    """The dict written to ``etude_decoder_config.json`` (EtudeDecoderConfig fields, etude_decoder.py:32-81)."""
    return dict(decoder_dims(**(dims or {})), model_type="etude_decoder")


def decoder_state_dict(seed: int = 0, dims: Dict[str, int] | None = None, gain: float = 1.0, emb_gain: float = 2.0,
                       follow: float = 1.0, p_eos: float = 0.06) -> Dict[str, np.ndarray]:
    """Seeded weights.  Plain N(0, 0.02) weights make greedy decoding collapse to one repeated token
    (SURVEY.md §7 hard part 6), which would pin nothing.  So the word embeddings are scaled up
    (``emb_gain``) and ``lm_head`` gets, on top of its random part, a random successor map
    ``f: token -> token`` (``W[f(i)] += follow * unit(E[i])``; a fraction ``p_eos`` of tokens map to
    ``Bar_EOS``).  Greedy sequences then follow ``f`` about 60-80 % of the time and are deflected by
    the attention/MLP context the rest of the time: bars have varied lengths, end in Bar_EOS, and
    depend on the attribute conditioning."""
    d = decoder_dims(**(dims or {}))
    rng = np.random.default_rng(seed)
    H, I, V, E = d["hidden_size"], d["intermediate_size"], d["vocab_size"], d["attribute_emb_dim"]
    std = d["initializer_range"] * gain

    def nrm(*shape):
        return (std * rng.standard_normal(shape)).astype(np.float32)

    sd: Dict[str, np.ndarray] = {}
    sd["word_embeddings.weight"] = nrm(V, H) * 4 * emb_gain
    sd["word_embeddings.weight"][d["pad_token_id"]] = 0
    sd["class_embeddings.weight"] = nrm(d["num_classes"], H) * 4
    sd["class_embeddings.weight"][d["pad_class_id"]] = 0
    for n in ("pitch_overlap", "polyphony", "note_sustain", "rhythm_intensity"):
        w = nrm(d["num_attribute_bins"], E) * 8
        w[d["attribute_pad_id"]] = 0
        sd[f"{n}_embeddings.weight"] = w
    sd["attribute_projection.weight"] = nrm(H, 4 * E)
    sd["attribute_projection.bias"] = nrm(H)
    sd["transformer.embed_in.weight"] = nrm(V, H)
    for i in range(d["num_hidden_layers"]):
        p = f"transformer.layers.{i}."
        for ln in ("input_layernorm", "post_attention_layernorm"):
            sd[p + ln + ".weight"], sd[p + ln + ".bias"] = _ln(rng, H)
        sd[p + "attention.query_key_value.weight"] = nrm(3 * H, H)
        sd[p + "attention.query_key_value.bias"] = nrm(3 * H)
        sd[p + "attention.dense.weight"] = nrm(H, H)
        sd[p + "attention.dense.bias"] = nrm(H)
        sd[p + "mlp.dense_h_to_4h.weight"] = nrm(I, H)
        sd[p + "mlp.dense_h_to_4h.bias"] = nrm(I)
        sd[p + "mlp.dense_4h_to_h.weight"] = nrm(H, I)
        sd[p + "mlp.dense_4h_to_h.bias"] = nrm(H)
    sd["transformer.final_layer_norm.weight"], sd["transformer.final_layer_norm.bias"] = _ln(rng, H)
    W = nrm(V, H)
    rng2 = np.random.default_rng(seed + 1000)
    f = rng2.integers(6, V, size=V)
    f[rng2.random(V) < p_eos] = 5                       # id of Bar_EOS in vocab_tokens()
    E = sd["word_embeddings.weight"]
    En = E / (np.linalg.norm(E, axis=1, keepdims=True) + 1e-6)
    sc = follow * float(np.linalg.norm(W, axis=1).mean())
    for i in range(V):
        W[f[i]] += sc * En[i]
    sd["lm_head.weight"] = W.astype(np.float32)
    return sd


def decoder_state_dict_ctx(seed: int = 0, dims: Dict[str, int] | None = None, follow: float = 1.0, p_eos: float = 0.08, qk_gain: float = 6.0,
                           v_gain: float = 2.0, out_gain: float = 2.0, emb_gain: float = 2.0) -> Dict[str, np.ndarray]:
    """Seeded weights whose GREEDY PATH DEPENDS ON THE CONTEXT (the parity goldens `decoder_ctx` / `clip_ctx`).

    `decoder_state_dict`'s successor map is a random function, whose walks fall into a cycle of ~8 tokens, and its deflections always
    land on the same few tokens (the common-mode part of the final hidden state): the greedy ids of the round-1/2 goldens are 94 % two
    alternating tokens, which a fault in the long-range part of the computation would rarely flip.  Here
      * the successor map is ONE long cycle over the non-special tokens (a random permutation), with a fraction `p_eos` of the tokens
        sent to Bar_EOS instead, so an undisturbed walk visits the whole vocabulary and bars end;
      * query / key projections are scaled by `qk_gain` (sharp, content-addressed attention: which past token a head locks onto changes
        from step to step and with every change of the context), values and the two output projections by `v_gain` / `out_gain`, so
        what attention retrieves is as large in the residual stream as the current token's embedding.
    Measured on the configs[1] condition bars (tests/golden/make_golden.py): ~44-55 % of the ids are predictable from the previous one, 60+ distinct
    ids per 16 bars, most bars end in Bar_EOS; dropping a key block, a wrong RoPE position or a wrong row changes the ids within a few tokens."""
    d = decoder_dims(**(dims or {}))
    sd = decoder_state_dict(seed, dims, gain=1.0, emb_gain=emb_gain, follow=0.0)
    H, V, nh = d["hidden_size"], d["vocab_size"], d["num_attention_heads"]
    hd = H // nh
    for i in range(d["num_hidden_layers"]):
        p = f"transformer.layers.{i}."
        w = sd[p + "attention.query_key_value.weight"].reshape(nh, 3, hd, H)          # HF layout: per head [q | k | v]
        b = sd[p + "attention.query_key_value.bias"].reshape(nh, 3, hd)
        w[:, :2] *= qk_gain; b[:, :2] *= qk_gain
        w[:, 2] *= v_gain; b[:, 2] *= v_gain
        sd[p + "attention.dense.weight"] *= out_gain
        sd[p + "mlp.dense_4h_to_h.weight"] *= out_gain
    rng = np.random.default_rng(seed + 2000)
    perm = rng.permutation(np.arange(6, V))
    f = np.zeros(V, np.int64)
    f[perm] = np.roll(perm, -1)
    f[:6] = perm[:6]
    f[rng.random(V) < p_eos] = 5                        # id of Bar_EOS in vocab_tokens()
    W = sd["lm_head.weight"].copy()
    E = sd["word_embeddings.weight"]
    En = E / (np.linalg.norm(E, axis=1, keepdims=True) + 1e-6)
    sc = follow * float(np.linalg.norm(W, axis=1).mean())
    for i in range(V):
        W[f[i]] += sc * En[i]
    sd["lm_head.weight"] = W.astype(np.float32)
    return sd


def vocab_tokens(n_pos: int = 48) -> Tuple[List[str], List[str]]:
    """A REMI-like vocabulary shaped like the reference tokenizer's events (etude/data/tokenizer.py:253-297):
    specials, Bar_BOS/EOS, Pos_k, Note_21..108, Duration_{allowed 16ths}, Grace_{1,-1}.  154 tokens at n_pos=48."""
    special = ["<PAD>", "<UNK>", "<BOS>", "<EOS>"]
    toks = list(special) + ["Bar_BOS", "Bar_EOS"]
    toks += [f"Pos_{i}" for i in range(n_pos)]
    toks += [f"Note_{p}" for p in range(21, 109)]
    toks += [f"Duration_{v}" for v in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)]
    toks += ["Grace_1", "Grace_-1"]
    return toks, special


def vocab_json(n_pos: int = 48) -> Dict:
    toks, special = vocab_tokens(n_pos)
    return {"token_to_id": {t: i for i, t in enumerate(toks)}, "special_tokens": special}


def write_vocab(path: str, n_pos: int = 48) -> None:
    with open(path, "w", encoding="utf-8") as f:
        json.dump(vocab_json(n_pos), f, ensure_ascii=False, indent=2)


def song_bars(seed: int = 0, n_bars: int = 92, notes_per_bar: int = 8, n_pos: int = 48, vocab: Dict | None = None) -> List[List[int]]:
    """Condition bars ``all_x_bars`` as the tokenizer would emit them for a ~8 notes/bar song
    (BASELINE config 1): [Bar_BOS, (Pos_k, Note_p, Duration_d)*, Bar_EOS]."""
    v = (vocab or vocab_json(n_pos))["token_to_id"]
    rng = np.random.default_rng(seed)
    durs = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)
    bars = []
    for _ in range(n_bars):
        n = int(rng.integers(max(1, notes_per_bar - 3), notes_per_bar + 4))
        pos = np.sort(rng.integers(0, 32, size=n))
        bar = [v["Bar_BOS"]]
        last = -1
        for p in pos:
            if p != last:
                bar.append(v[f"Pos_{int(p)}"])
                last = p
            bar.append(v[f"Note_{int(rng.integers(40, 91))}"])
            bar.append(v[f"Duration_{durs[int(rng.integers(0, len(durs)))]}"])
        bar.append(v["Bar_EOS"])
        bars.append(bar)
    return bars


def attrs(polyphony: int = 1, rhythm: int = 1, sustain: int = 1, overlap: int = 2) -> Dict[str, int]:
    """Target-attribute dict as infer.py builds it (infer.py:187, CLI defaults :280-300)."""
    return {"polyphony_bin": polyphony, "rhythm_intensity_bin": rhythm, "sustain_bin": sustain, "pitch_overlap_bin": overlap}

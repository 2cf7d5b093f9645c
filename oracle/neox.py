"""Oracle: EtudeDecoder forward + generate, torch-CPU fp32.  TEST INFRASTRUCTURE ONLY.

Restates
  * ``EtudeDecoder.forward``   etude/models/etude_decoder.py:148-206
  * ``EtudeDecoder.generate``  etude/models/etude_decoder.py:209-354
  * HF ``GPTNeoXModel`` (third-party dependency, transformers==4.51.3 pinned in
    requirements.txt:6; read from transformers 5.15.0's
    models/gpt_neox/modeling_gpt_neox.py): layer :239-281 (parallel residual), attention
    :195-236 (fused QKV laid out [head][q|k|v][head_dim]), partial RoPE :72-151
    (rotary_pct 0.25, theta 10000, rotate-half), erf-GELU MLP, final LayerNorm, eps 1e-5.

Pinned by golden vectors captured from the reference + transformers 5.15.0 in the build
container (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SRC_CLASS_ID = 1   # etude/data/dataset.py:18
TGT_CLASS_ID = 2   # etude/data/dataset.py:19

ATTR_KEY_MAP = {   # etude_decoder.py:238-243
    "polyphony_bin": "polyphony",
    "rhythm_intensity_bin": "rhythm_intensity",
    "sustain_bin": "note_sustain",
    "pitch_overlap_bin": "pitch_overlap",
}


@dataclass
class NeoxDims:
    vocab_size: int = 154
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    intermediate_size: int = 2048
    max_position_embeddings: int = 1024
    attribute_emb_dim: int = 64
    context_num_past_xy_pairs: int = 4
    rotary_pct: float = 0.25
    rope_theta: float = 10000.0
    layer_norm_eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def rotary_ndims(self) -> int:
        return int(self.head_dim * self.rotary_pct)


def embed(sd, ids, cls, attrs: Dict[str, torch.Tensor]):
    """etude_decoder.py:166-179.  attrs keys: pitch_overlap, polyphony, note_sustain, rhythm_intensity."""
    a = torch.cat([
        sd["pitch_overlap_embeddings.weight"][attrs["pitch_overlap"]],
        sd["polyphony_embeddings.weight"][attrs["polyphony"]],
        sd["note_sustain_embeddings.weight"][attrs["note_sustain"]],
        sd["rhythm_intensity_embeddings.weight"][attrs["rhythm_intensity"]],
    ], dim=-1)
    proj = F.linear(a, sd["attribute_projection.weight"], sd["attribute_projection.bias"])
    return sd["word_embeddings.weight"][ids] + sd["class_embeddings.weight"][cls] + proj


def _rope(x, pos, d: NeoxDims):
    """modeling_gpt_neox.py:111-151: rotate the first rotary_ndims dims of each head."""
    rd = d.rotary_ndims
    inv = 1.0 / (d.rope_theta ** (torch.arange(0, rd, 2, dtype=torch.float32) / rd))
    fr = pos.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    xr, xp = x[..., :rd], x[..., rd:]
    x1, x2 = xr[..., : rd // 2], xr[..., rd // 2:]
    rot = torch.cat([-x2, x1], dim=-1)
    return torch.cat([xr * cos + rot * sin, xp], dim=-1)


def transformer(sd, h, d: NeoxDims, kv: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None):
    """GPTNeoXModel.forward with inputs_embeds.  h [1,T,H]; kv = per-layer (K,V) [1,nh,ctx,hd] or None.

    Returns (final-LN hidden [1,T,H], new kv list).
    """
    B, T, H = h.shape
    nh, hd = d.num_attention_heads, d.head_dim
    past = 0 if kv is None else kv[0][0].shape[2]
    pos = torch.arange(T) + past
    new_kv = []
    for i in range(d.num_hidden_layers):
        p = f"transformer.layers.{i}."
        x1 = F.layer_norm(h, (H,), sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"], d.layer_norm_eps)
        qkv = F.linear(x1, sd[p + "attention.query_key_value.weight"], sd[p + "attention.query_key_value.bias"])
        qkv = qkv.view(B, T, nh, 3 * hd).transpose(1, 2)
        q, k, v = qkv.chunk(3, dim=-1)
        q, k = _rope(q, pos, d), _rope(k, pos, d)
        if kv is not None:
            k = torch.cat([kv[i][0], k], dim=2)
            v = torch.cat([kv[i][1], v], dim=2)
        new_kv.append((k, v))
        w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
        ctx = k.shape[2]
        mask = torch.arange(ctx)[None, :] > (torch.arange(T)[:, None] + past)
        w = w.masked_fill(mask[None, None], float("-inf"))
        w = torch.softmax(w, dim=-1, dtype=torch.float32)
        a = torch.matmul(w, v).transpose(1, 2).reshape(B, T, H)
        a = F.linear(a, sd[p + "attention.dense.weight"], sd[p + "attention.dense.bias"])
        x2 = F.layer_norm(h, (H,), sd[p + "post_attention_layernorm.weight"], sd[p + "post_attention_layernorm.bias"], d.layer_norm_eps)
        m = F.linear(x2, sd[p + "mlp.dense_h_to_4h.weight"], sd[p + "mlp.dense_h_to_4h.bias"])
        m = F.gelu(m)
        m = F.linear(m, sd[p + "mlp.dense_4h_to_h.weight"], sd[p + "mlp.dense_4h_to_h.bias"])
        h = m + a + h
    h = F.layer_norm(h, (H,), sd["transformer.final_layer_norm.weight"], sd["transformer.final_layer_norm.bias"], d.layer_norm_eps)
    return h, new_kv


@torch.no_grad()
def forward_logits(sd, d: NeoxDims, ids, cls, attrs, kv=None):
    """EtudeDecoder.forward -> (logits [1,T,V], kv)."""
    h, kv = transformer(sd, embed(sd, ids, cls, attrs), d, kv)
    return F.linear(h, sd["lm_head.weight"]), kv


def build_bar_prompt(history, x_bar, y_attrs, user_keys, bar_bos_id, bar_eos_id, d: NeoxDims,
                     max_bar_token_limit=512, context_overlap_ratio=0.5):
    """etude_decoder.py:257-296: prompt tokens/classes/attr lists for one bar (Bar_BOS appended)."""
    n_ctx = d.context_num_past_xy_pairs
    toks: List[int] = []
    cls: List[int] = []
    al: Dict[str, List[int]] = {k: [] for k in user_keys}
    hist = history[-n_ctx:]
    for _ in range(n_ctx - len(hist)):
        for c in (SRC_CLASS_ID, TGT_CLASS_ID):
            toks += [bar_bos_id, bar_eos_id]
            cls += [c, c]
            for k in user_keys:
                al[k] += [1, 1]
    for xs, ys, at in hist:
        for item, c in ((xs, SRC_CLASS_ID), (ys, TGT_CLASS_ID)):
            toks += list(item)
            cls += [c] * len(item)
            for k in user_keys:
                al[k] += [at[k]] * len(item)
    toks += list(x_bar)
    cls += [SRC_CLASS_ID] * len(x_bar)
    for k in user_keys:
        al[k] += [y_attrs[k]] * len(x_bar)
    if len(toks) > d.max_position_embeddings - max_bar_token_limit:
        keep = int(d.max_position_embeddings * context_overlap_ratio)
        toks, cls = toks[-keep:], cls[-keep:]
        for k in user_keys:
            al[k] = al[k][-keep:]
    toks = toks + [bar_bos_id]
    cls = cls + [TGT_CLASS_ID]
    for k in user_keys:
        al[k] = al[k] + [y_attrs[k]]
    return toks, cls, al


@torch.no_grad()
def generate_ids(sd, d: NeoxDims, bar_bos_id: int, bar_eos_id: int, all_x_bars, target_attributes_per_bar,
                 max_output_tokens=25600, max_bar_token_limit=512, context_overlap_ratio=0.5, force_bar_tokens: int = 0) -> List[List[int]]:
    """Greedy (temperature==0) branch of generate(); returns per-bar id lists ``[Bar_BOS]+tokens``.

    The reference returns ``vocab.decode_sequence_to_events`` of exactly these ids
    (etude_decoder.py:350), so id equality <=> event equality.

    ``force_bar_tokens=n`` is NOT reference behaviour: it mirrors the library's benchmark switch of the same name (Bar_EOS does
    not end a bar, every bar is exactly n tokens long, truncation still by ``max_bar_token_limit``) so that the CPU baseline and
    the parity tests of the benchmark workload run the same thing as the GPU.
    """
    if not all_x_bars or len(all_x_bars) != len(target_attributes_per_bar):
        return []
    user_keys = sorted(target_attributes_per_bar[0].keys())
    total = 0
    history = []
    out: List[List[int]] = []
    for i, x_bar in enumerate(all_x_bars):
        y_attrs = target_attributes_per_bar[i]
        toks, cls, al = build_bar_prompt(history, x_bar, y_attrs, user_keys, bar_bos_id, bar_eos_id, d,
                                         max_bar_token_limit, context_overlap_ratio)
        ids_t = torch.tensor([toks])
        cls_t = torch.tensor([cls])
        at_t = {ATTR_KEY_MAP[k]: torch.tensor([al[k]]) for k in user_keys}
        kv = None
        bar: List[int] = []
        for _ in range(force_bar_tokens if force_bar_tokens > 0 else max_bar_token_limit):
            if total >= max_output_tokens:
                break
            logits, kv = forward_logits(sd, d, ids_t, cls_t, at_t, kv)
            nxt = int(torch.argmax(logits[:, -1, :], dim=-1).item())
            bar.append(nxt)
            total += 1
            if nxt == bar_eos_id and force_bar_tokens <= 0:
                break
            ids_t = torch.tensor([[nxt]])
            cls_t = torch.tensor([[TGT_CLASS_ID]])
            at_t = {ATTR_KEY_MAP[k]: torch.tensor([[y_attrs[k]]]) for k in user_keys}
        history.append((x_bar, [bar_bos_id] + bar, y_attrs))
        if len(history) > d.context_num_past_xy_pairs:
            history.pop(0)
        out.append([bar_bos_id] + bar)
        if total >= max_output_tokens:
            break
    return out


@torch.no_grad()
def sampling_distribution(next_logits: torch.Tensor, temperature: float, top_p: float) -> torch.Tensor:
    """The distribution `generate` draws the next token from when temperature > 0 -- etude/models/etude_decoder.py:321-330,
    everything but the `torch.multinomial` call itself.  next_logits [1, V] fp32 -> probs [1, V]."""
    probs = F.softmax(next_logits / temperature, dim=-1)
    if 0 < top_p < 1.0:
        sorted_probs, sorted_indices = torch.sort(probs, descending=True)
        cum_probs = torch.cumsum(sorted_probs, dim=-1)
        indices_to_remove = cum_probs > top_p
        indices_to_remove[..., 1:] = indices_to_remove[..., :-1].clone()
        indices_to_remove[..., 0] = 0
        probs[0, sorted_indices[0, indices_to_remove[0]]] = 0
        probs = probs / probs.sum()
    return probs

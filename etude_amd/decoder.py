"""Drop-in replacement for the reference's Decode-stage model object.

``load_etude_decoder(config_path, checkpoint_path, device)`` (etude/utils/model_loader.py:12-60) returns an
``EtudeDecoder`` whose ``generate(vocab, all_x_bars, target_attributes_per_bar, ...)`` has the
signature, defaults, error behaviour and return value of etude/models/etude_decoder.py:209-354.
Prompt assembly / history / truncation / token budget stay on the host in Python (they are list
manipulation); every forward pass, the KV cache, the greedy argmax and the token feedback run on the
GPU through libetude_hip.so (etd_decoder_*).

``generate_many`` runs many independent (song, attribute tuple) jobs as concurrent device streams
(continuous batching); each job's result equals what ``generate`` returns for it.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib

SRC_CLASS_ID = 1   # etude/data/dataset.py:18
TGT_CLASS_ID = 2   # etude/data/dataset.py:19
# C-ABI attribute order = concat order of etude_decoder.py:171-176, keyed by generate()'s user keys (:238-243)
ABI_ATTR_KEYS = ("pitch_overlap_bin", "polyphony_bin", "sustain_bin", "rhythm_intensity_bin")


class EtudeDecoderConfig:
    """The fields of etude/models/etude_decoder.py:32-81 that shape the computation (HF boilerplate in the JSON is ignored)."""
    model_type = "etude_decoder"

    def __init__(self, vocab_size: int = 3000, pad_token_id: int = 0, hidden_size: int = 512, num_hidden_layers: int = 8,
                 num_attention_heads: int = 8, intermediate_size: int = 2048, max_position_embeddings: int = 1024,
                 num_classes: int = 3, pad_class_id: int = 0, attribute_pad_id: int = 0, context_num_past_xy_pairs: int = 4,
                 num_attribute_bins: int = 3, attribute_emb_dim: int = 64, initializer_range: float = 0.02, **kwargs):
        self.vocab_size, self.pad_token_id, self.hidden_size = vocab_size, pad_token_id, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.num_classes, self.pad_class_id, self.attribute_pad_id = num_classes, pad_class_id, attribute_pad_id
        self.context_num_past_xy_pairs, self.num_attribute_bins = context_num_past_xy_pairs, num_attribute_bins
        self.attribute_emb_dim, self.initializer_range = attribute_emb_dim, initializer_range
        # GPT-NeoX defaults resolved by HF for this config; overridable from the JSON
        rp = kwargs.get("rope_parameters") or {}
        self.rotary_pct = float(kwargs.get("rotary_pct", rp.get("partial_rotary_factor", 0.25)))
        self.rope_theta = float(kwargs.get("rotary_emb_base", kwargs.get("rope_theta", rp.get("rope_theta", 10000.0))))
        self.layer_norm_eps = float(kwargs.get("layer_norm_eps", 1e-5))
        if kwargs.get("hidden_act", "gelu") != "gelu" or not kwargs.get("use_parallel_residual", True):
            raise _lib.EtudeHipError("only hidden_act='gelu' with use_parallel_residual=True (the reference's resolved defaults) is implemented")

    @classmethod
    def from_json_file(cls, path: Union[str, Path]) -> "EtudeDecoderConfig":
        with open(path, "r", encoding="utf-8") as f:
            return cls(**json.load(f))


def assemble_bar_prompt(history, x_bar, y_attrs, user_keys, bar_bos_id, bar_eos_id, n_ctx_pairs, max_pos,
                        max_bar_token_limit, context_overlap_ratio):
    """Prompt of one bar as etude_decoder.py:257-296 builds it: [empty-pair padding][<=n past (X,Y) pairs][X_i][Bar_BOS]."""
    toks: List[int] = []
    cls: List[int] = []
    at = {k: [] for k in user_keys}
    hist = history[-n_ctx_pairs:] if n_ctx_pairs > 0 else []
    for _ in range(n_ctx_pairs - len(hist)):
        for c in (SRC_CLASS_ID, TGT_CLASS_ID):
            toks.extend((bar_bos_id, bar_eos_id))
            cls.extend((c, c))
            for k in user_keys:
                at[k].extend((1, 1))                      # neutral bin 1 for every key (etude_decoder.py:250)
    for xs, ys, a in hist:
        for item, c in ((xs, SRC_CLASS_ID), (ys, TGT_CLASS_ID)):
            toks.extend(item)
            cls.extend([c] * len(item))
            for k in user_keys:
                at[k].extend([a[k]] * len(item))
    toks.extend(x_bar)
    cls.extend([SRC_CLASS_ID] * len(x_bar))
    for k in user_keys:
        at[k].extend([y_attrs[k]] * len(x_bar))
    if len(toks) > max_pos - max_bar_token_limit:
        keep = int(max_pos * context_overlap_ratio)
        toks, cls = toks[-keep:], cls[-keep:]
        for k in user_keys:
            at[k] = at[k][-keep:]
    toks.append(bar_bos_id)
    cls.append(TGT_CLASS_ID)
    for k in user_keys:
        at[k].append(y_attrs[k])
    return toks, cls, at


def assemble_bar_prompt_np(history, x_bar, y4, bar_bos_id, bar_eos_id, n_ctx_pairs, max_pos, max_bar_token_limit, context_overlap_ratio):
    """numpy twin of ``assemble_bar_prompt`` for the batched engine: history = [(x ndarray, y ndarray, attrs4 ndarray)],
    y4 = target attrs in C-ABI order.  Returns (tokens[T], classes[T], attrs[4][T]) as int32 arrays, Bar_BOS appended."""
    hist = history[-n_ctx_pairs:] if n_ctx_pairs > 0 else []
    segs, seg_cls, seg_at = [], [], []
    empty = np.asarray([bar_bos_id, bar_eos_id], np.int32)
    neutral = np.ones(4, np.int32)
    for _ in range(n_ctx_pairs - len(hist)):
        for c in (SRC_CLASS_ID, TGT_CLASS_ID):
            segs.append(empty); seg_cls.append(c); seg_at.append(neutral)
    for xs, ys, a4 in hist:
        segs.append(xs); seg_cls.append(SRC_CLASS_ID); seg_at.append(a4)
        segs.append(ys); seg_cls.append(TGT_CLASS_ID); seg_at.append(a4)
    segs.append(x_bar); seg_cls.append(SRC_CLASS_ID); seg_at.append(y4)
    lens = np.asarray([len(x) for x in segs], np.int64)
    toks = np.concatenate(segs).astype(np.int32, copy=False)
    cls = np.repeat(np.asarray(seg_cls, np.int32), lens)
    at = np.repeat(np.stack(seg_at).astype(np.int32), lens, axis=0).T          # [4][T]
    if toks.size > max_pos - max_bar_token_limit:
        keep = int(max_pos * context_overlap_ratio)
        toks, cls, at = toks[-keep:], cls[-keep:], at[:, -keep:]
    toks = np.concatenate([toks, np.asarray([bar_bos_id], np.int32)])
    cls = np.concatenate([cls, np.asarray([TGT_CLASS_ID], np.int32)])
    at = np.concatenate([at, np.asarray(y4, np.int32)[:, None]], axis=1)
    return toks, cls, np.ascontiguousarray(at)


class _Job:
    __slots__ = ("x_bars", "attrs", "x_np", "a4", "keys", "max_out", "bar_limit", "overlap", "i", "history", "total", "bars_out", "slot", "limit")

    def __init__(self, x_bars, attrs, max_out, bar_limit, overlap):
        self.x_bars, self.attrs = x_bars, attrs
        self.x_np = [np.asarray(b, np.int32) for b in x_bars]
        self.a4 = [np.asarray([a[k] for k in ABI_ATTR_KEYS], np.int32) for a in attrs]
        self.keys = sorted(attrs[0].keys())
        self.max_out, self.bar_limit, self.overlap = max_out, bar_limit, overlap
        self.i, self.history, self.total, self.bars_out, self.slot, self.limit = 0, [], 0, [], -1, 0


class EtudeDecoder:
    """GPU-resident EtudeDecoder.  ``state`` maps the reference's state-dict keys to fp32 arrays."""

    def __init__(self, config: EtudeDecoderConfig, state: Dict[str, np.ndarray], device: Union[str, torch.device] = "cuda",
                 precision: Optional[str] = None, max_streams: int = 1, max_ctx: Optional[int] = None,
                 max_prefill_rows: Optional[int] = None):
        if device == "auto":
            device = "cuda"
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.EtudeHipError("etude_amd.EtudeDecoder needs a ROCm GPU (device='cuda'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.config = config
        precision = precision or os.environ.get("ETD_DECODER_PRECISION", "fp32")
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' (token-parity mode) or 'bf16'")
        self.precision = precision
        self.max_streams = int(max_streams)
        # the reference never exceeds ~max_position_embeddings + 1 positions per bar (etude_decoder.py:285-300)
        self.max_ctx = int(max_ctx) if max_ctx else int(config.max_position_embeddings) + 64
        # one begin_bars call carries up to this many prompt rows (a prompt is <= max_pos/2 + 1 tokens after truncation)
        self.max_prefill_rows = int(max_prefill_rows) if max_prefill_rows else self.max_streams * (int(config.max_position_embeddings) // 2 + 8)
        self.max_prefill_rows = max(self.max_prefill_rows, self.max_ctx)
        cfg = _lib.DecCfg(vocab_size=config.vocab_size, hidden_size=config.hidden_size, num_hidden_layers=config.num_hidden_layers,
                          num_attention_heads=config.num_attention_heads, intermediate_size=config.intermediate_size,
                          max_position_embeddings=config.max_position_embeddings, num_classes=config.num_classes,
                          num_attribute_bins=config.num_attribute_bins, attribute_emb_dim=config.attribute_emb_dim,
                          rotary_pct=config.rotary_pct, rope_theta=config.rope_theta, layer_norm_eps=config.layer_norm_eps,
                          max_streams=self.max_streams, max_ctx=self.max_ctx, precision=1 if precision == "bf16" else 0,
                          max_prefill_rows=self.max_prefill_rows)
        names, ptrs, numels, n, keep = _lib.weights_arrays(state)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_create(C.byref(cfg), names, ptrs, numels, n, C.byref(h)), "etd_decoder_create")
        self._h = h
        # decode steps are replayed as hipGraphs, which cannot be captured on the default stream
        self._ts = torch.cuda.Stream(device=self.device)

    # ------------------------------------------------------------------ reference surface
    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def _stream(self):
        return C.c_void_p(self._ts.cuda_stream)

    def _validate(self, vocab, all_x_bars, target_attributes_per_bar):
        try:
            bos, eos = vocab.get_bar_bos_id(), vocab.get_bar_eos_id()
            if bos == -1 or eos == -1:
                raise ValueError("Bar tokens not in vocab.")
        except Exception as e:  # etude_decoder.py:225-232
            print(f"[etude_amd] ERROR Accessing vocab/config: {e}")
            return None
        if not all_x_bars or len(all_x_bars) != len(target_attributes_per_bar):
            print("[etude_amd] ERROR Condition bars mismatch with target attributes.")
            return None
        missing = [k for k in ABI_ATTR_KEYS if k not in target_attributes_per_bar[0]]
        if missing:   # the reference's forward() would raise TypeError for the missing positional tensors
            raise TypeError(f"generate() missing attribute keys {missing}")
        return bos, eos

    @torch.no_grad()
    def generate(self, vocab, all_x_bars: List[List[int]], target_attributes_per_bar: List[Dict[str, int]],
                 max_output_tokens: int = 25600, max_bar_token_limit: int = 512, temperature: float = 0.8,
                 top_p: float = 0.9, context_overlap_ratio: float = 0.5) -> List:
        """etude_decoder.py:209-354.  Returns the list of Events (``vocab.decode_sequence_to_events``)."""
        ids = self.generate_ids(vocab, all_x_bars, target_attributes_per_bar, max_output_tokens, max_bar_token_limit,
                                temperature, top_p, context_overlap_ratio)
        events = []
        for bar in ids:
            events.extend(vocab.decode_sequence_to_events(bar))
        return events

    def generate_ids(self, vocab, all_x_bars, target_attributes_per_bar, max_output_tokens=25600, max_bar_token_limit=512,
                     temperature=0.8, top_p=0.9, context_overlap_ratio=0.5) -> List[List[int]]:
        """Same loop, but returns the per-bar id lists ``[Bar_BOS] + tokens`` (what the events are decoded from)."""
        if temperature > 0:
            raise NotImplementedError("etude_amd decodes greedily (temperature == 0, the reference's configured default, "
                                      "schema.py:223); the temperature/top-p sampling branch is not implemented yet")
        r = self.generate_many([(all_x_bars, target_attributes_per_bar)], vocab, max_output_tokens, max_bar_token_limit,
                               context_overlap_ratio, _validate=True)
        return r[0]

    # ------------------------------------------------------------------ multi-stream engine
    def generate_many(self, jobs: Sequence[Tuple[List[List[int]], List[Dict[str, int]]]], vocab, max_output_tokens: int = 25600,
                      max_bar_token_limit: int = 512, context_overlap_ratio: float = 0.5, steps_per_poll: int = 8,
                      _validate: bool = True, stats: Optional[dict] = None, force_bar_tokens: int = 0) -> List[List[List[int]]]:
        """Greedy-decode many independent jobs on up to ``max_streams`` concurrent device streams.

        ``force_bar_tokens=n`` (benchmarks only) suppresses Bar_EOS and makes every bar exactly n tokens long, so that
        throughput does not depend on where synthetic weights happen to emit EOS."""
        lib = _lib.lib()
        cfg = self.config
        results: List[Optional[List[List[int]]]] = [None] * len(jobs)
        todo: List[Tuple[int, _Job]] = []
        bos = eos = -1
        for ji, (x_bars, attrs) in enumerate(jobs):
            v = self._validate(vocab, x_bars, attrs)
            if v is None:
                results[ji] = []
                continue
            bos, eos = v
            todo.append((ji, _Job(x_bars, attrs, max_output_tokens, max_bar_token_limit, context_overlap_ratio)))
        free = list(range(self.max_streams))[::-1]
        active: Dict[int, Tuple[int, _Job]] = {}      # slot -> (job index, job)
        n_steps_total = n_tokens = 0
        st = self._stream()

        def start_bars(batch: List[_Job]) -> List[_Job]:
            """Prefill the next bar of every job in `batch` in as few device passes as possible (all prompts of a call
            go through the model as one batch).  Returns the jobs that are finished instead."""
            finished, pend = [], []
            for job in batch:
                if job.i >= len(job.x_bars):
                    finished.append(job)
                    continue
                y_attrs = job.attrs[job.i]
                job.limit = min(force_bar_tokens or job.bar_limit, job.max_out - job.total)
                if job.limit <= 0:
                    # the reference's inner loop breaks before the first forward: the bar is just [Bar_BOS]
                    job.bars_out.append([bos])
                    finished.append(job)
                    continue
                toks, cls, at = assemble_bar_prompt_np(job.history, job.x_np[job.i], job.a4[job.i], bos, eos,
                                                       cfg.context_num_past_xy_pairs, cfg.max_position_embeddings,
                                                       job.bar_limit, job.overlap)
                pend.append((job, toks, cls, at, job.a4[job.i]))
            while pend:
                take, rows = [], 0
                while pend and (not take or rows + len(pend[0][1]) <= self.max_prefill_rows):
                    rows += len(pend[0][1])
                    take.append(pend.pop(0))
                n = len(take)
                slots = np.asarray([t[0].slot for t in take], np.int32)
                T = np.asarray([len(t[1]) for t in take], np.int32)
                ids = np.ascontiguousarray(np.concatenate([t[1] for t in take]), np.int32)
                cl = np.ascontiguousarray(np.concatenate([t[2] for t in take]), np.int32)
                a4 = np.ascontiguousarray(np.concatenate([t[3] for t in take], axis=1), np.int32)
                tg = np.ascontiguousarray(np.stack([t[4] for t in take]), np.int32)
                eo = np.full(n, -1 if force_bar_tokens else eos, np.int32)
                li = np.asarray([t[0].limit for t in take], np.int32)
                _lib.check(lib.etd_decoder_begin_bars(self._h, n, slots.ctypes.data, T.ctypes.data, ids.ctypes.data, cl.ctypes.data,
                                                      a4.ctypes.data, tg.ctypes.data, eo.ctypes.data, li.ctypes.data, st), "etd_decoder_begin_bars")
            return finished

        def read_done(done_jobs: List[_Job]) -> List[List[int]]:
            n = len(done_jobs)
            sl = np.asarray([j.slot for j in done_jobs], np.int32)
            out = np.zeros((n, 1024), np.int32)
            cnt = np.zeros(n, np.int32)
            _lib.check(lib.etd_decoder_read_many(self._h, n, sl.ctypes.data, out.ctypes.data, 1024, cnt.ctypes.data, st), "etd_decoder_read_many")
            return [out[i, : cnt[i]].tolist() for i in range(n)]

        def finish_bar(job: _Job, toks: List[int]):
            job.total += len(toks)
            job.history.append((job.x_np[job.i], np.asarray([bos] + toks, np.int32), job.a4[job.i]))
            if len(job.history) > cfg.context_num_past_xy_pairs:
                job.history.pop(0)
            job.bars_out.append([bos] + toks)
            job.i += 1
            return len(toks)

        with torch.cuda.device(self.device):
            pending = todo[::-1]
            job_index = {id(job): ji for ji, job in todo}

            def retire(job: _Job):
                results[job_index[id(job)]] = job.bars_out
                if job.slot in active:
                    del active[job.slot]
                free.append(job.slot)

            while pending or active:
                fresh = []
                while pending and free:
                    ji, job = pending.pop()
                    job.slot = free.pop()
                    active[job.slot] = (ji, job)
                    fresh.append(job)
                if fresh:
                    for job in start_bars(fresh):
                        retire(job)
                if not active:
                    continue
                slots = np.asarray(sorted(active.keys()), np.int32)
                dn = np.zeros(len(slots), np.int32)
                no = np.zeros(len(slots), np.int32)
                _lib.check(lib.etd_decoder_poll(self._h, slots.ctypes.data, len(slots), dn.ctypes.data, no.ctypes.data, st), "etd_decoder_poll")
                again = []
                done_jobs = [active[s][1] for s, d in zip(slots.tolist(), dn.tolist()) if d]
                for job, toks in zip(done_jobs, read_done(done_jobs) if done_jobs else []):
                    n_tokens += finish_bar(job, toks)
                    if job.total >= job.max_out:                 # etude_decoder.py:352
                        retire(job)
                    else:
                        again.append(job)
                if again:
                    for job in start_bars(again):
                        retire(job)
                    continue                                      # re-poll / refill before stepping
                if any(dn):
                    continue
                nstep = steps_per_poll
                if force_bar_tokens:      # no early EOS possible: run every stream to the nearest bar end in one call
                    nstep = max(1, min(active[s][1].limit - int(c) for s, c in zip(slots.tolist(), no.tolist())))
                _lib.check(lib.etd_decoder_step(self._h, slots.ctypes.data, len(slots), nstep, st), "etd_decoder_step")
                n_steps_total += nstep
        if stats is not None:
            stats["steps"] = n_steps_total
            stats["tokens"] = n_tokens
        return results  # type: ignore[return-value]

    # ------------------------------------------------------------------ test / bench hooks
    def prefill_logits(self, ids, cls, attrs4, slot: int = 0) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int32)
        cls = np.ascontiguousarray(cls, np.int32)
        attrs4 = np.ascontiguousarray(attrs4, np.int32)
        T = ids.shape[0]
        out = np.zeros((T, self.config.vocab_size), np.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_prefill_logits(self._h, slot, ids.ctypes.data, cls.ctypes.data, attrs4.ctypes.data, T,
                                                             out.ctypes.data, self._stream()), "etd_decoder_prefill_logits")
        return out

    def step_bytes(self, n_streams: int, ctx: int) -> float:
        return float(_lib.lib().etd_decoder_step_bytes(self._h, n_streams, ctx))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().etd_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_decoder_state(checkpoint_path: Union[str, Path]) -> Dict[str, np.ndarray]:
    """Checkpoint -> name->fp32 array: accepts a bare state dict or the training payload
    ``{"model_state_dict": ...}`` and strips ``_orig_mod.`` (model_loader.py:44-53)."""
    sd = torch.load(checkpoint_path, map_location="cpu")
    if "model_state_dict" in sd:
        sd = sd["model_state_dict"]
    out = OrderedDict()
    for k, v in sd.items():
        if torch.is_tensor(v):
            out[k.replace("_orig_mod.", "")] = v.detach().to(torch.float32).cpu().numpy()
    return out


def expected_state_keys(cfg: EtudeDecoderConfig) -> List[str]:
    keys = ["word_embeddings.weight", "class_embeddings.weight", "pitch_overlap_embeddings.weight", "polyphony_embeddings.weight",
            "note_sustain_embeddings.weight", "rhythm_intensity_embeddings.weight", "attribute_projection.weight",
            "attribute_projection.bias", "transformer.embed_in.weight"]
    for i in range(cfg.num_hidden_layers):
        p = f"transformer.layers.{i}."
        for m in ("input_layernorm", "post_attention_layernorm", "attention.query_key_value", "attention.dense",
                  "mlp.dense_h_to_4h", "mlp.dense_4h_to_h"):
            keys += [p + m + ".weight", p + m + ".bias"]
    keys += ["transformer.final_layer_norm.weight", "transformer.final_layer_norm.bias", "lm_head.weight"]
    return keys


def load_etude_decoder(config_path: Union[str, Path], checkpoint_path: Union[str, Path], device: str = "auto",
                       precision: Optional[str] = None, max_streams: int = 1) -> EtudeDecoder:
    """model_loader.py:12-60: JSON config -> model -> checkpoint (strict key match) -> eval."""
    config = EtudeDecoderConfig.from_json_file(str(config_path))
    state = load_decoder_state(checkpoint_path)
    exp = set(expected_state_keys(config))
    got = {k for k in state if not k.endswith("rotary_emb.inv_freq") and not k.endswith("attention.bias") and not k.endswith("attention.masked_bias")}
    missing, unexpected = sorted(exp - got), sorted(got - exp)
    if missing or unexpected:     # load_state_dict(strict=True) semantics (model_loader.py:56)
        raise RuntimeError(f"Error(s) in loading state_dict for EtudeDecoder: missing keys {missing}; unexpected keys {unexpected}")
    return EtudeDecoder(config, state, device=device, precision=precision, max_streams=max_streams)

"""Drop-in replacement for the reference's Decode-stage model object.

``load_etude_decoder(config_path, checkpoint_path, device)`` (etude/utils/model_loader.py:12-60) returns an
``EtudeDecoder`` whose ``generate(vocab, all_x_bars, target_attributes_per_bar, ...)`` has the
signature, defaults, error behaviour and return value of etude/models/etude_decoder.py:209-354.
Prompt assembly / history / truncation / token budget run in the library's native scheduler
(csrc/sched_dec.cpp); every forward pass, the KV cache, the greedy argmax and the token feedback run on the
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


class PackedBars:
    """Condition bars of one song as arrays: ``ids`` int32 (all bars back to back) + ``offsets`` int32 [n_bars + 1].  Shared by every
    attribute-tuple job of the song, so a 27-tuple grid converts its bars once."""

    def __init__(self, ids: np.ndarray, offsets: np.ndarray):
        self.ids = np.ascontiguousarray(ids, np.int32)
        self.offsets = np.ascontiguousarray(offsets, np.int32)
        if self.offsets.ndim != 1 or self.offsets.size < 1 or int(self.offsets[0]) != 0 or int(self.offsets[-1]) != self.ids.size or (np.diff(self.offsets) < 0).any():
            raise ValueError("PackedBars: offsets must rise from 0 to len(ids)")
        self.checked_for = None

    @classmethod
    def from_lists(cls, bars: Sequence[Sequence[int]]) -> "PackedBars":
        lens = np.asarray([len(b) for b in bars], np.int64)
        offs = np.zeros(len(bars) + 1, np.int32)
        offs[1:] = np.cumsum(lens)
        return cls(np.concatenate([np.asarray(b, np.int32) for b in bars]) if len(bars) else np.zeros(0, np.int32), offs)

    def __len__(self) -> int:
        return int(self.offsets.size) - 1

    def bar(self, i: int) -> List[int]:
        return self.ids[self.offsets[i]: self.offsets[i + 1]].tolist()


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
        # "fp32" (default: the reference's token ids) or the 16-bit serving mode "f16" ("bf16" is accepted as its older name: the operands are IEEE half since round 5,
        # bf16 in a -DETD_DEC_BF16 build -- `etd_decoder_operand_type`)
        precision = precision or os.environ.get("ETD_DECODER_PRECISION", "fp32")
        if precision not in ("fp32", "f16", "bf16"):
            raise ValueError("precision must be 'fp32' (token-parity mode) or 'f16' (alias 'bf16': the 16-bit serving mode)")
        precision = "fp32" if precision == "fp32" else "f16"
        self.precision = precision
        # the arithmetic type of the 16-bit mode's operands in THIS library build (etd_decoder_operand_type: 1 = IEEE half, 0 = bf16 of a -DETD_DEC_BF16 build)
        self.operand_dtype = torch.float32 if precision == "fp32" else (torch.float16 if _lib.lib().etd_decoder_operand_type() == 1 else torch.bfloat16)
        self.max_streams = int(max_streams)
        # KV positions per stream.  A bar touches prompt + limit - 1 positions, and the prompt (after the truncation rule of
        # etude_decoder.py:285-289, + Bar_BOS) is at most max(max_pos - limit, int(max_pos * ratio)) + 1 tokens: with the reference's
        # default arguments that is ~max_pos + 1 positions, with context_overlap_ratio up to 1 and max_bar_token_limit up to
        # 1024 (the output ring of a stream) up to 2 * max_pos.  HF's GPT-NeoX has no such bound (dynamic cache, rotary computed
        # on the fly), so the default covers every such call; `ctx_needed` below rejects what does not fit instead of
        # truncating silently.
        self.max_ctx = int(max_ctx) if max_ctx else 2 * int(config.max_position_embeddings) + 64
        # one begin_bars call carries up to this many prompt rows (a prompt is <= max_pos/2 + 1 tokens after truncation)
        self.max_prefill_rows = int(max_prefill_rows) if max_prefill_rows else self.max_streams * (int(config.max_position_embeddings) // 2 + 8)
        self.max_prefill_rows = max(self.max_prefill_rows, self.max_ctx)
        cfg = _lib.DecCfg(vocab_size=config.vocab_size, hidden_size=config.hidden_size, num_hidden_layers=config.num_hidden_layers,
                          num_attention_heads=config.num_attention_heads, intermediate_size=config.intermediate_size,
                          max_position_embeddings=config.max_position_embeddings, num_classes=config.num_classes,
                          num_attribute_bins=config.num_attribute_bins, attribute_emb_dim=config.attribute_emb_dim,
                          rotary_pct=config.rotary_pct, rope_theta=config.rope_theta, layer_norm_eps=config.layer_norm_eps,
                          max_streams=self.max_streams, max_ctx=self.max_ctx, precision=1 if precision == "f16" else 0,
                          max_prefill_rows=self.max_prefill_rows)
        names, ptrs, numels, n, keep = _lib.weights_arrays(state)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_create(C.byref(cfg), names, ptrs, numels, n, C.byref(h)), "etd_decoder_create")
        self._h = h
        # decode steps are replayed as hipGraphs, which cannot be captured on the default stream
        self._ts = torch.cuda.Stream(device=self.device)

    def clone(self) -> "EtudeDecoder":
        """A second engine over the same device weights (own KV cache / workspaces / stream state): what `generate_many`
        callers use to run several engines side by side without one weight copy per engine."""
        other = object.__new__(EtudeDecoder)
        for k in ("device", "config", "precision", "operand_dtype", "max_streams", "max_ctx", "max_prefill_rows"):
            setattr(other, k, getattr(self, k))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_clone(self._h, C.byref(h)), "etd_decoder_clone")
        other._h = h
        other._ts = torch.cuda.Stream(device=self.device)
        return other

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
        if all_x_bars is None or len(all_x_bars) == 0 or target_attributes_per_bar is None or len(all_x_bars) != len(target_attributes_per_bar):
            print("[etude_amd] ERROR Condition bars mismatch with target attributes.")
            return None
        if isinstance(target_attributes_per_bar, np.ndarray):
            if target_attributes_per_bar.ndim != 2 or target_attributes_per_bar.shape[1] != 4:
                raise TypeError("attribute array must be [n_bars, 4] in ABI_ATTR_KEYS order")
            return bos, eos
        missing = [k for k in ABI_ATTR_KEYS if k not in target_attributes_per_bar[0]]
        if missing:   # the reference's forward() would raise TypeError for the missing positional tensors
            raise TypeError(f"generate() missing attribute keys {missing}")
        return bos, eos

    @torch.no_grad()
    def generate(self, vocab, all_x_bars: List[List[int]], target_attributes_per_bar: List[Dict[str, int]],
                 max_output_tokens: int = 25600, max_bar_token_limit: int = 512, temperature: float = 0.8,
                 top_p: float = 0.9, context_overlap_ratio: float = 0.5, seed: Optional[int] = None) -> List:
        """etude_decoder.py:209-354.  Returns the list of Events (``vocab.decode_sequence_to_events``).
        ``temperature == 0``: greedy (token ids identical to the reference).  ``temperature > 0``: the reference's
        softmax(logits / T) + top-p filter + one draw per token (:321-331); draws come from a counter-based generator keyed by
        ``seed`` (default: derived from ``torch.initial_seed()`` and a per-model call counter), so runs are reproducible but not
        bit-equal to torch.multinomial's stream."""
        ids = self.generate_ids(vocab, all_x_bars, target_attributes_per_bar, max_output_tokens, max_bar_token_limit,
                                temperature, top_p, context_overlap_ratio, seed)
        events = []
        for bar in ids:
            events.extend(vocab.decode_sequence_to_events(bar))
        return events

    def generate_ids(self, vocab, all_x_bars, target_attributes_per_bar, max_output_tokens=25600, max_bar_token_limit=512,
                     temperature=0.8, top_p=0.9, context_overlap_ratio=0.5, seed=None) -> List[List[int]]:
        """Same loop, but returns the per-bar id lists ``[Bar_BOS] + tokens`` (what the events are decoded from)."""
        r = self.generate_many([(all_x_bars, target_attributes_per_bar)], vocab, max_output_tokens, max_bar_token_limit,
                               context_overlap_ratio, _validate=True, temperature=temperature, top_p=top_p, seed=seed)
        return r[0]

    # ------------------------------------------------------------------ multi-stream engine
    def generate_many(self, jobs: Sequence[Tuple[List[List[int]], List[Dict[str, int]]]], vocab, max_output_tokens: int = 25600,
                      max_bar_token_limit: int = 512, context_overlap_ratio: float = 0.5, steps_per_poll: int = 8,
                      _validate: bool = True, stats: Optional[dict] = None, force_bar_tokens: int = 0,
                      ready: Optional[Tuple[np.ndarray, Sequence[int]]] = None, temperature: float = 0.0, top_p: float = 0.9,
                      seed: Optional[int] = None, _job_key: Tuple[int, int] = (0, 1), as_arrays: bool = False) -> List[List[List[int]]]:
        """Greedy-decode many independent jobs on up to ``max_streams`` concurrent device streams.

        The bar loop (prompt assembly, history, truncation, token budget, EOS stop) runs in the library's native
        scheduler (csrc/sched_dec.cpp, ``etd_decoder_run_jobs``); this method only marshals the jobs.
        ``force_bar_tokens=n`` (benchmarks only) suppresses Bar_EOS and makes every bar exactly n tokens long, so that
        throughput does not depend on where synthetic weights happen to emit EOS.
        ``ready=(flags, index)`` gates admission on upstream pipeline stages: ``flags`` is an int32 array another thread sets
        non-zero (e.g. one entry per song, set when its extract..tokenize stages are done) and job i waits for
        ``flags[index[i]]``; jobs are admitted in list order.

        Batch fast paths (no per-bar Python objects; what bench.py's 1 728-job batches use): a job's condition bars may be a
        ``PackedBars`` (one int32 id array + bar offsets, shared by all attribute tuples of a song) and its attributes an
        ``int32 [n_bars, 4]`` array in ``ABI_ATTR_KEYS`` order; ``as_arrays=True`` returns each job as ``(flat_ids, bar_lens)``
        instead of a list of lists."""
        lib = _lib.lib()
        if not temperature >= 0:
            raise ValueError("temperature must be >= 0")
        lim = int(force_bar_tokens or max_bar_token_limit)
        need = self.ctx_needed(max_bar_token_limit, context_overlap_ratio, gen_limit=lim)
        if need > self.max_ctx:
            raise _lib.EtudeHipError(f"generate: max_bar_token_limit={max_bar_token_limit} (bars of up to {lim} generated tokens) with context_overlap_ratio={context_overlap_ratio} needs {need} KV positions "
                                     f"per stream, this decoder was created with max_ctx={self.max_ctx}; pass max_ctx>={need} to EtudeDecoder / load_etude_decoder")
        if seed is None:
            self._draw_calls = getattr(self, "_draw_calls", 0) + 1
            seed = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + self._draw_calls) & 0xFFFFFFFFFFFFFFFF
        if ready is not None:
            flags, ridx = ready
            if flags.dtype != np.int32 or not flags.flags["C_CONTIGUOUS"] or len(ridx) != len(jobs):
                raise ValueError("ready: need a contiguous int32 flag array and one index per job")
        cfg = self.config
        results: List[Optional[List[List[int]]]] = [None] * len(jobs)
        live: List[int] = []
        bos = eos = -1
        for ji, (x_bars, attrs) in enumerate(jobs):
            v = self._validate(vocab, x_bars, attrs)
            if v is None:
                results[ji] = (np.zeros(0, np.int32), np.zeros(0, np.int64)) if as_arrays else []
                continue
            bos, eos = v
            live.append(ji)
        if live:
            keep = []                                   # numpy buffers referenced by the job descriptors
            cjobs = (_lib.Job * len(live))()
            cap = 0
            for k, ji in enumerate(live):
                x_bars, attrs = jobs[ji]
                if isinstance(x_bars, PackedBars):
                    xi, offs = x_bars.ids, x_bars.offsets
                    if not x_bars.checked_for == cfg.vocab_size:
                        if xi.size and (xi.min() < 0 or xi.max() >= cfg.vocab_size):
                            raise IndexError("token id out of range in all_x_bars")
                        x_bars.checked_for = cfg.vocab_size
                else:
                    lens = np.asarray([len(b) for b in x_bars], np.int64)
                    offs = np.zeros(len(x_bars) + 1, np.int32)
                    offs[1:] = np.cumsum(lens)
                    xi = np.ascontiguousarray(np.concatenate([np.asarray(b, np.int32) for b in x_bars]) if len(x_bars) else np.zeros(0, np.int32))
                    if xi.size and (xi.min() < 0 or xi.max() >= cfg.vocab_size):
                        raise IndexError("token id out of range in all_x_bars")
                if isinstance(attrs, np.ndarray):
                    a4 = np.ascontiguousarray(attrs, np.int32).reshape(-1, 4)
                else:
                    a4 = np.ascontiguousarray(np.asarray([[a[key] for key in ABI_ATTR_KEYS] for a in attrs], np.int32).reshape(-1, 4))
                keep += [offs, xi, a4]
                gate = None
                if ready is not None:
                    if not 0 <= int(ridx[ji]) < flags.size:
                        raise IndexError("ready: flag index out of range")
                    gate = flags.ctypes.data + 4 * int(ridx[ji])
                cjobs[k] = _lib.Job(xi.ctypes.data, offs.ctypes.data, len(x_bars), a4.ctypes.data, gate)
                per_bar = (force_bar_tokens or max_bar_token_limit) + 1
                cap += 1 + len(x_bars) + min(len(x_bars) * per_bar, max(0, max_output_tokens) + len(x_bars) + per_bar)
            sc = _lib.SchedCfg(bar_bos_id=bos, bar_eos_id=eos, n_ctx_pairs=cfg.context_num_past_xy_pairs,
                               max_position_embeddings=cfg.max_position_embeddings, max_output_tokens=max_output_tokens,
                               max_bar_token_limit=max_bar_token_limit, context_overlap_ratio=context_overlap_ratio,
                               force_bar_tokens=force_bar_tokens, max_streams=self.max_streams,
                               max_prefill_rows=self.max_prefill_rows, steps_per_poll=steps_per_poll,
                               temperature=float(temperature), top_p=float(top_p), seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                               job_key_offset=int(_job_key[0]), job_key_stride=int(_job_key[1]))
            out = np.zeros(cap, np.int32)
            offs_out = np.zeros(len(live) + 1, np.int64)
            nsteps = C.c_longlong()
            with torch.cuda.device(self.device):
                _lib.check(lib.etd_decoder_run_jobs(self._h, C.byref(sc), cjobs, len(live), out.ctypes.data, cap, offs_out.ctypes.data,
                                                    C.byref(nsteps), self._stream()), "etd_decoder_run_jobs")
            n_tokens = 0
            for k, ji in enumerate(live):
                rec = out[offs_out[k]: offs_out[k + 1]]
                nb = int(rec[0])
                lens = rec[1:1 + nb].astype(np.int64)
                flat = rec[1 + nb:]
                ends = np.cumsum(lens)
                results[ji] = (flat.copy(), lens) if as_arrays else [flat[e - l: e].tolist() for l, e in zip(lens.tolist(), ends.tolist())]
                n_tokens += int(lens.sum()) - nb
            if stats is not None:
                stats["steps"] = int(nsteps.value)
                stats["tokens"] = n_tokens
        elif stats is not None:
            stats["steps"] = stats["tokens"] = 0
        return results  # type: ignore[return-value]

    def ctx_needed(self, max_bar_token_limit: int, context_overlap_ratio: float, gen_limit: Optional[int] = None) -> int:
        """KV positions a bar can touch under generate()'s truncation rule (etude_decoder.py:285-300): the prompt bound follows
        ``max_bar_token_limit`` (what the scheduler truncates with), the generated length ``gen_limit`` (= the limit unless a benchmark
        forces longer / shorter bars)."""
        mp = int(self.config.max_position_embeddings)
        prompt = max(mp - int(max_bar_token_limit), int(mp * float(context_overlap_ratio))) + 1
        return prompt + int(gen_limit if gen_limit is not None else max_bar_token_limit) - 1

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

    STAT_KEYS = ("steps", "row_steps", "kv_bytes", "attn_launches", "stamped_launches", "stamped_seconds", "stamped_alg_bytes", "weight_bytes_per_step")

    def stats(self) -> Dict[str, float]:
        """Exact accounting of the decode steps issued since `stats_reset` (etd_decoder_stats): steps, rows x steps, algorithmic
        K/V bytes, attention launches, and -- when `stamp(True)` was on -- the device-measured duration of those launches."""
        out = (C.c_double * 8)()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_stats(self._h, out, 8, self._stream()), "etd_decoder_stats")
        return dict(zip(self.STAT_KEYS, [float(x) for x in out]))

    def stats_reset(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_stats_reset(self._h, self._stream()), "etd_decoder_stats_reset")

    def stamp(self, on: bool, skip_steps: int = 0) -> None:
        """Device-side span measurement of every k_dstep_attn_down launch (measurement runs only; own captured graphs); the first
        ``skip_steps`` decode steps after switching on are left out."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_stamp(self._h, 1 if on else 0, int(skip_steps), self._stream()), "etd_decoder_stamp")

    def stamp_log(self) -> np.ndarray:
        """(start, end) of every stamped attention launch since the last stats_reset, uint64 [n, 2] in 100 MHz device ticks (one clock for the whole chip)"""
        buf = np.zeros((131072 + 2, 2), np.uint64)
        n = C.c_longlong()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_decoder_stamp_log(self._h, buf.ctypes.data, buf.shape[0], C.byref(n), self._stream()), "etd_decoder_stamp_log")
        return buf[: n.value]

    def debug_step_logits(self, on: bool, n_active: int = 0) -> Optional[np.ndarray]:
        """Test hook: switch the per-step logit store on / off; with n_active > 0 also return the LAST step's logits [n_active, V]."""
        out = np.zeros((n_active, self.config.vocab_size), np.float32) if n_active > 0 else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_debug_decoder_step_logits(self._h, 1 if on else 0, out.ctypes.data if out is not None else None, n_active, self._stream()),
                       "etd_debug_decoder_step_logits")
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().etd_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_engines(engines: Sequence["EtudeDecoder"], jobs, vocab, one_at_a_time: bool = False, ready=None, stagger_s: float = 0.0, **generate_many_kwargs):
    """`generate_many` over several engines at once: the job list is dealt round-robin over ``engines`` (normally one
    EtudeDecoder and its ``clone()``s, which share the device weights) and each engine runs its share from its own host
    thread -- ctypes releases the GIL inside the library calls.  On MI355X four engines is the useful maximum: a GPU has
    four compute pipes, and a fifth chain of dependent kernels halves the one it shares a pipe with (LABNOTES.md).

    ``ready=(flags, job_index_to_flag)`` is split per engine like the jobs.  Returns ``join``: calling it waits for the
    engines and returns ``(results in job order, per-engine stats dicts)``; exceptions of the workers are re-raised there.
    ``one_at_a_time`` runs the engines one after the other (profiling passes).  ``stagger_s``: engine i starts i * stagger_s
    seconds after engine 0, so that engines whose bars all take the same time do not prefill (and then step) in phase."""
    import threading
    import time as _time
    n = len(engines)
    seed = generate_many_kwargs.pop("seed", None)
    if seed is None and float(generate_many_kwargs.get("temperature", 0.0) or 0.0) > 0:
        # sampled covers: one seed per call for every engine (each engine deriving its own from its private call counter gave all
        # engines the same seed AND the same engine-local job indices, i.e. identical draw streams for job k of every engine)
        e0 = engines[0]
        e0._draw_calls = getattr(e0, "_draw_calls", 0) + 1
        seed = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + e0._draw_calls) & 0xFFFFFFFFFFFFFFFF
    outs = [None] * n
    stats = [dict() for _ in engines]
    errs: list = []

    def run(i):
        try:
            torch.cuda.set_device(engines[i].device)
            kw = dict(generate_many_kwargs)
            kw["seed"] = seed                         # ONE seed for the whole job list ...
            kw["_job_key"] = (i, n)                   # ... and GLOBAL job indices in the draw keys: engine i holds jobs i, i + n, ...
            if ready is not None:
                kw["ready"] = (ready[0], ready[1][i::n])
            if stagger_s > 0 and i > 0 and not one_at_a_time:
                _time.sleep(i * stagger_s)
            outs[i] = engines[i].generate_many(jobs[i::n], vocab, stats=stats[i], **kw)
        except Exception as e:      # noqa: BLE001 -- surfaced by join()
            errs.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
        if one_at_a_time:
            t.join()

    def join():
        if not one_at_a_time:
            for t in th:
                t.join()
        if errs:
            raise errs[0]
        out = [None] * len(jobs)
        for i in range(n):
            out[i::n] = outs[i]
        return out, stats

    return join


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
                       precision: Optional[str] = None, max_streams: int = 1, max_ctx: Optional[int] = None) -> EtudeDecoder:
    """model_loader.py:12-60: JSON config -> model -> checkpoint (strict key match) -> eval."""
    config = EtudeDecoderConfig.from_json_file(str(config_path))
    state = load_decoder_state(checkpoint_path)
    exp = set(expected_state_keys(config))
    got = {k for k in state if not k.endswith("rotary_emb.inv_freq") and not k.endswith("attention.bias") and not k.endswith("attention.masked_bias")}
    missing, unexpected = sorted(exp - got), sorted(got - exp)
    if missing or unexpected:     # load_state_dict(strict=True) semantics (model_loader.py:56)
        raise RuntimeError(f"Error(s) in loading state_dict for EtudeDecoder: missing keys {missing}; unexpected keys {unexpected}")
    return EtudeDecoder(config, state, device=device, precision=precision, max_streams=max_streams, max_ctx=max_ctx)

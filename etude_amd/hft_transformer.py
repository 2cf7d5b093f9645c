"""HFT_Transformer on MI355X -- drop-in for etude/models/hft_transformer.py:36-460.

prepare.py's transcriber (prepare.py:98-132) wraps the same hFT-Transformer network as the Extract stage, at
``num_frame=128`` with half-overlapping windows (``n_stride=32``), zero STFT padding and ``min_value=-80``.  Every
kernel is the Extract stage's (csrc/ext_kernels.hip, frontend.hip, mpe2note_dev.hip); this module only mirrors the
wrapper's call surface: ``HFT_Transformer(config, model_path, device).transcribe(wav, json)`` plus the underscore methods
the reference exposes (``_wav2feature``, ``_transcript``, ``_transcript_stride``, ``_mpe2note``).
"""
from __future__ import annotations

import io
import json
import pickle
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from . import _lib
from .config import ExtractorConfig, HFTConfig
from .extractor import AMTAPC_Extractor, read_wav


# ---------------------------------------------------------------------------------------------------------------------
# checkpoint: the reference unpickles a whole nn.Module (hft_transformer.py:26-56).  Here the pickle is read WITHOUT
# importing any model code: module classes become inert records, tensors are rebuilt by torch, and the parameter /
# buffer tree is flattened into the state-dict keys the extractor loads (encoder_spec2midi -> encoder, ...).
class _Record:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):     # (dict, slots) form
            state = {**(state[0] or {}), **state[1]}
        self.__dict__.update(state)


_TORCH_OK = {("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_tensor"),
             ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"),
             ("torch.serialization", "_get_layout"), ("collections", "OrderedDict"), ("torch._tensor", "_rebuild_from_type_v2")}


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "torch.storage" and name == "_load_from_bytes":
            return lambda b: torch.load(io.BytesIO(b), map_location="cpu", weights_only=True)
        if (module, name) in _TORCH_OK:
            if module == "collections":
                return OrderedDict
            obj = __import__(module, fromlist=[name])
            return getattr(obj, name)
        if module == "torch" and (name.endswith("Storage") or name in ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int8", "uint8", "bool")):
            return getattr(torch, name)
        if module == "torch.nn.parameter" and name == "Parameter":
            return torch.nn.Parameter
        if module.startswith("model") or module.startswith("etude.models") or module.startswith("torch.nn.modules"):
            return type(name, (_Record,), {})
        raise pickle.UnpicklingError(f"hFT checkpoint refers to {module}.{name}, which this loader does not admit")


def _flatten(mod, prefix: str, out: Dict[str, np.ndarray]):
    for table in ("_parameters", "_buffers"):
        for k, v in (getattr(mod, table, None) or {}).items():
            if v is not None and torch.is_tensor(v):
                out[prefix + k] = v.detach().to(torch.float32).cpu().numpy()
    for k, m in (getattr(mod, "_modules", None) or {}).items():
        if m is not None:
            _flatten(m, prefix + k + ".", out)


def load_hft_state(path_model: Union[str, Path]) -> Dict[str, np.ndarray]:
    """Pickled hFT-Transformer model object (or a flat state dict saved with torch.save) -> extractor state dict."""
    path_model = Path(path_model)
    if not path_model.exists():
        raise FileNotFoundError(path_model)
    try:
        sd = torch.load(path_model, weights_only=True, map_location="cpu")
        flat = {k: v.detach().to(torch.float32).cpu().numpy() for k, v in sd.items() if torch.is_tensor(v)}
    except Exception:
        with open(path_model, "rb") as f:
            root = _ModelUnpickler(f).load()
        flat = {}
        _flatten(root, "", flat)
    ren = {}
    for k, v in flat.items():
        k = k.replace("encoder_spec2midi.", "encoder.", 1) if k.startswith("encoder_spec2midi.") else k
        k = k.replace("decoder_spec2midi.", "decoder.", 1) if k.startswith("decoder_spec2midi.") else k
        ren[k] = v
    return ren


class HFT_Transformer:
    """Signature of etude/models/hft_transformer.py:40 / :75."""

    def __init__(self, config: Optional[HFTConfig], model_path: Union[str, Path, Dict[str, np.ndarray]], device: str = "auto",
                 max_windows: int = 16):
        self.config = config if config is not None else HFTConfig()
        c = self.config
        if c.infer.mode != "combination":
            raise _lib.EtudeHipError("only mode='combination' (the reference's configuration, schema.py:187) is implemented")
        state = model_path if isinstance(model_path, dict) else load_hft_state(model_path)
        nf_ckpt = state["decoder.pos_embedding_time.weight"].shape[0] if "decoder.pos_embedding_time.weight" in state else c.input.num_frame
        if nf_ckpt != c.input.num_frame:
            raise _lib.EtudeHipError(f"checkpoint was built for num_frame={nf_ckpt}, config says {c.input.num_frame}")
        ec = ExtractorConfig()
        ec.feature.sr, ec.feature.hop_sample, ec.feature.n_bins, ec.feature.mel_bins = c.feature.sr, c.feature.hop_sample, c.feature.n_bins, c.feature.mel_bins
        ec.feature.fft_bins, ec.feature.window_length, ec.feature.log_offset = c.feature.fft_bins, c.feature.window_length, c.feature.log_offset
        ec.input.margin_b, ec.input.margin_f, ec.input.num_frame, ec.input.min_value = c.input.margin_b, c.input.margin_f, c.input.num_frame, c.input.min_value
        ec.midi = c.midi
        self._ex = AMTAPC_Extractor(ec, state, device, max_windows=max_windows, stft_pad_mode=c.feature.pad_mode)
        self.device = self._ex.device

    # ------------------------------------------------------------------ reference surface
    def transcribe(self, input_wav_path: Union[str, Path], output_json_path: Union[str, Path]):
        """hft_transformer.py:75-117: wav -> notes JSON (indent=4, no duration filter)."""
        feature = self._wav2feature(input_wav_path)
        n_stride = self.config.infer.n_stride
        inf = self.config.infer
        if n_stride > 0:
            on, off, mpe, vel = self._transcript_stride_dev(feature, n_stride)
        else:
            on, off, mpe, vel = self._ex.transcript(feature)
        arr = self._ex.mpe2note_device(on, off, mpe, vel, inf.thred_onset, inf.thred_offset, inf.thred_mpe)
        notes = self._ex._notes_from_array(arr)
        output_path = Path(output_json_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        with open(output_path, "w", encoding="utf-8") as f:
            json.dump(notes, f, ensure_ascii=False, indent=4)

    def _wav2feature(self, f_wav: Union[str, Path]) -> torch.Tensor:
        """hft_transformer.py:120-138; returns the DEVICE tensor [T, mel_bins]."""
        wave, sr = read_wav(f_wav)
        return self._ex.wav2feature_tensor(wave, sr)

    def _transcript(self, a_feature, mode="combination") -> tuple:
        """hft_transformer.py:140-280 (non-overlapping windows) -> the 8 host arrays."""
        if mode != "combination":
            raise _lib.EtudeHipError("only mode='combination' is implemented")
        feat = torch.as_tensor(np.asarray(a_feature.cpu() if torch.is_tensor(a_feature) else a_feature, dtype=np.float32)).to(self.device)
        return tuple(o.cpu().numpy() for o in self._ex.transcript(feat, want_A=True))

    def _transcript_stride(self, a_feature, n_offset: int, mode="combination") -> tuple:
        """hft_transformer.py:282-460 (half-overlapping windows) -> the 8 host arrays (A outputs first, like the reference)."""
        if mode != "combination":
            raise _lib.EtudeHipError("only mode='combination' is implemented")
        feat = torch.as_tensor(np.asarray(a_feature.cpu() if torch.is_tensor(a_feature) else a_feature, dtype=np.float32)).to(self.device)
        outs = self._transcript_stride_dev(feat, n_offset, want_A=True)
        return tuple(o.cpu().numpy() for o in outs)

    def _mpe2note(self, a_onset=None, a_offset=None, a_mpe=None, a_velocity=None, thred_onset=0.5, thred_offset=0.5, thred_mpe=0.5,
                  mode_velocity="ignore_zero", mode_offset="shorter") -> List[dict]:
        """hft_transformer.py:462-674 (the same algorithm as extractor.py:256-418)."""
        return self._ex._mpe2note(a_onset, a_offset, a_mpe, a_velocity, thred_onset, thred_offset, thred_mpe, mode_velocity, mode_offset)

    # ------------------------------------------------------------------ device path
    def _transcript_stride_dev(self, feat: torch.Tensor, n_offset: int, want_A: bool = False):
        """Windows start every num_frame/2 frames; of each window's num_frame output rows, rows [n_offset, n_offset + half)
        are kept (hft_transformer.py:352-435).  All windows go through the model as one batch sequence."""
        c = self.config
        nf, mb, mf = c.input.num_frame, c.input.margin_b, c.input.margin_f
        half = nf // 2
        if not 0 <= n_offset <= half:
            raise ValueError("n_offset must lie in [0, num_frame/2]")
        T = feat.shape[0]
        tmp_len = T + mb + mf + half
        len_s = int(np.ceil(tmp_len / half) * half) - tmp_len
        pad = lambda n: torch.full((n, feat.shape[1]), float(c.input.min_value), dtype=torch.float32, device=feat.device)   # noqa: E731
        a_in = torch.cat([pad(mb + n_offset), feat.to(torch.float32), pad(len_s + mf + (half - n_offset))], dim=0)
        starts = list(range(0, T, half))
        win = mb + nf + mf
        spec = torch.stack([a_in[i:i + win].T for i in starts], dim=0).contiguous()          # [B, n_bin, win]
        outs = self._ex.transcript_windows(spec, want_A=want_A)                               # each [B * nf, n_note]
        nn = outs[0].shape[1]
        keep = [o.view(len(starts), nf, nn)[:, n_offset:n_offset + half].reshape(len(starts) * half, nn) for o in outs]
        total = T + len_s
        res = []
        for o in keep:                                                                        # arrays are [T + len_s] long, zero beyond the last window
            full = torch.zeros((total, nn), dtype=o.dtype, device=o.device)
            n = min(total, o.shape[0])
            full[:n] = o[:n]
            res.append(full)
        return tuple(res)

    def close(self):
        self._ex.close()

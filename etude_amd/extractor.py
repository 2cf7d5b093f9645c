"""Drop-in replacement for the reference's Extract-stage wrapper.

``AMTAPC_Extractor(config, model_path, device).extract(audio_path, output_json_path, output_midi_path)``
has the reference's signature, file side effects and error behaviour
(etude/data/extractor.py:116-176); the arithmetic runs in libetude_hip.so:

    wav --H2D--> etd_frontend_run (resample + STFT + mel + log)          extractor.py:178-197
        --> etd_transcript (hFT-Transformer, windows batched on device)  extractor.py:199-253
        --D2H (4 arrays)--> etd_mpe2note (host C++)                      extractor.py:256-418
        --> JSON                                                         extractor.py:432-446

Extras the reference does not have: ``transcript(features)`` / ``transcript_windows(spec)`` returning
the ``_transcript`` arrays for parity checks, ``extract_many`` for clip batches, ``wav2feature_tensor``.
"""
from __future__ import annotations

import ctypes as C
import json
import wave as _wave
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .config import ExtractorConfig
from .frontend import FrontEnd


NOTE_DTYPE = np.dtype([("onset", "<f8"), ("offset", "<f8"), ("pitch", "<i4"), ("velocity", "<i4")])   # == etd_note


def read_wav(path: Union[str, Path]):
    """PCM/float WAV -> (float32 [C, L] in [-1, 1), sample_rate); stands in for ``torchaudio.load``
    (extractor.py:180) for the RIFF/WAVE files infer.py feeds it (origin.wav, infer.py:61-80)."""
    with open(path, "rb") as f:
        head = f.read(12)
    if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    data = Path(path).read_bytes()
    pos = 12
    fmt = None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], int.from_bytes(data[pos + 4:pos + 8], "little")
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag = int.from_bytes(body[0:2], "little")
            ch = int.from_bytes(body[2:4], "little")
            sr = int.from_bytes(body[4:8], "little")
            bits = int.from_bytes(body[14:16], "little")
            if tag == 0xFFFE and len(body) >= 26:
                tag = int.from_bytes(body[24:26], "little")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            if fmt is None:
                raise ValueError(f"{path}: data chunk before fmt chunk")
            tag, ch, sr, bits = fmt
            if tag == 3 and bits == 32:
                x = np.frombuffer(body, "<f4").astype(np.float32)
            elif tag == 3 and bits == 64:
                x = np.frombuffer(body, "<f8").astype(np.float32)
            elif tag == 1 and bits == 16:
                x = np.frombuffer(body, "<i2").astype(np.float32) / 32768.0
            elif tag == 1 and bits == 32:
                x = np.frombuffer(body, "<i4").astype(np.float32) / 2147483648.0
            elif tag == 1 and bits == 24:
                b = np.frombuffer(body[: len(body) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
                v = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
                v = np.where(v >= 1 << 23, v - (1 << 24), v)
                x = v.astype(np.float32) / 8388608.0
            elif tag == 1 and bits == 8:
                x = (np.frombuffer(body, np.uint8).astype(np.float32) - 128.0) / 128.0
            else:
                raise ValueError(f"{path}: unsupported WAV format tag={tag} bits={bits}")
            n = (x.size // ch) * ch
            return np.ascontiguousarray(x[:n].reshape(-1, ch).T), sr
        pos += 8 + size + (size & 1)
    raise ValueError(f"{path}: no data chunk")


def write_wav_f32(path: Union[str, Path], wav: np.ndarray, sr: int) -> None:
    """float32 [C, L] -> IEEE-float WAV (test/bench helper)."""
    wav = np.asarray(wav, np.float32)
    ch, n = wav.shape
    body = np.ascontiguousarray(wav.T).astype("<f4").tobytes()
    hdr = b"RIFF" + (36 + len(body)).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") + \
        (3).to_bytes(2, "little") + ch.to_bytes(2, "little") + sr.to_bytes(4, "little") + (sr * ch * 4).to_bytes(4, "little") + \
        (ch * 4).to_bytes(2, "little") + (32).to_bytes(2, "little") + b"data" + len(body).to_bytes(4, "little")
    Path(path).write_bytes(hdr + body)


def load_extractor_state(path_model: Union[str, Path]) -> Dict[str, np.ndarray]:
    """``torch.load(weights_only=True)`` of the flat extractor state dict (extractor.py:108); tensors -> fp32 numpy."""
    sd = torch.load(path_model, weights_only=True, map_location="cpu")
    return {k: v.detach().to(torch.float32).cpu().numpy() for k, v in sd.items() if torch.is_tensor(v)}


class AMTAPC_Extractor:
    """Audio -> notes (JSON/MIDI) on one MI355X.  Signature of etude/data/extractor.py:121-146."""

    def __init__(self, config: Optional[ExtractorConfig], model_path: Union[str, Path, Dict[str, np.ndarray]],
                 device: Union[str, torch.device] = "auto", max_windows: int = 4, chunk_frames: int = 0, stft_pad_mode: str = "reflect",
                 precision: Optional[str] = None):
        if device == "auto":
            device = "cuda"
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.EtudeHipError("etude_amd.AMTAPC_Extractor needs a ROCm GPU (device='cuda'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.config = config if config is not None else ExtractorConfig()
        c = self.config
        # "f16" (default; "bf16" is accepted as its older name): the 16-bit serving path -- IEEE-half operands (bf16 in a -DETD_EXT_BF16 build: `operand_dtype`),
        # fp32 accumulate / LayerNorm / softmax / sigmoid.  "fp32": the exact-parity mode (csrc/ext_fp32.hip), fp32 activations and fp32-grade products like the
        # reference (extractor.py runs the model in fp32), ~4x slower; also selectable with ETD_EXTRACTOR_PRECISION.
        # The 16-bit kernels are built for the reference's default architecture (etude/config/schema.py:100-112); any other architecture the reference can build
        # with head_dim 64 runs on the fp32 engine, which is general: that is the default precision for it (an explicit "f16" is refused by the library).
        import os
        m = c.model
        default_arch = (m.transformer_hid_dim, m.encoder_n_head, m.transformer_pf_dim, c.feature.n_bins, c.input.margin_b, m.cnn_channel, m.cnn_kernel,
                        m.encoder_n_layer, m.decoder_n_layer, c.midi.num_velocity) == (256, 4, 512, 256, 32, 4, 5, 3, 3, 128)
        precision = precision or os.environ.get("ETD_EXTRACTOR_PRECISION") or ("f16" if default_arch else "fp32")
        if precision not in ("f16", "bf16", "fp32"):
            raise ValueError("precision must be 'f16' (alias 'bf16') or 'fp32'")
        self.precision = "fp32" if precision == "fp32" else "f16"
        self.operand_dtype = torch.float32 if precision == "fp32" else (torch.float16 if _lib.lib().etd_extractor_operand_type() == 1 else torch.bfloat16)
        state = model_path if isinstance(model_path, dict) else load_extractor_state(model_path)
        cfg = _lib.ExtCfg(n_margin=c.input.margin_b, n_frame=c.input.num_frame, n_bin=c.feature.n_bins,
                          cnn_channel=c.model.cnn_channel, cnn_kernel=c.model.cnn_kernel, hid_dim=c.model.transformer_hid_dim,
                          pf_dim=c.model.transformer_pf_dim, n_heads=c.model.encoder_n_head,
                          n_layers_enc=c.model.encoder_n_layer, n_layers_dec=c.model.decoder_n_layer,
                          n_note=c.midi.num_note, n_velocity=c.midi.num_velocity, min_value=c.input.min_value,
                          max_windows=max_windows, chunk_frames=chunk_frames, precision=1 if precision == "fp32" else 0)
        if c.input.margin_b != c.input.margin_f:
            raise _lib.EtudeHipError("margin_b != margin_f is not supported")
        if c.model.encoder_n_head != c.model.decoder_n_head:
            raise _lib.EtudeHipError("encoder_n_head != decoder_n_head is not supported")
        names, ptrs, numels, n, keep = _lib.weights_arrays(state)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_extractor_create(C.byref(cfg), names, ptrs, numels, n, C.byref(h)), "etd_extractor_create")
        self._h = h
        self._m2n = None          # device mpe2note handle, created on first use
        self._stft_pad_mode = stft_pad_mode     # "reflect": what extractor.py:186-193 gets from torchaudio's default; "constant": hft_transformer.py:130
        self._fronts: Dict[int, FrontEnd] = {}
        self.n_note = c.midi.num_note
        self.n_frame = c.input.num_frame
        self.window_flops = float(_lib.lib().etd_extractor_window_flops(self._h))

    # ------------------------------------------------------------------ reference surface
    def extract(self, audio_path: str, output_json_path: str, output_midi_path: Optional[str] = None):
        """extractor.py:148-176."""
        wave, sr = read_wav(audio_path)
        min_duration = self.config.infer.min_duration
        notes = self.extract_notes(wave, sr, min_duration)       # filtered exactly as _note2json / _note2midi would
        self._note2json(notes, output_json_path, min_duration)
        if output_midi_path:
            self._note2midi(notes, output_midi_path, min_duration)

    def _wav2feature(self, audio_path: str) -> torch.Tensor:
        """extractor.py:178-197 (returns a CPU tensor like the reference)."""
        wave, sr = read_wav(audio_path)
        return self.wav2feature_tensor(wave, sr).cpu()

    def _transcript(self, a_feature, sv=None, silent=True, mode="combination", ablation_flag=False):
        """extractor.py:199-253: [T, n_mels] -> the 8 (or 4) numpy arrays."""
        if sv is not None:
            raise _lib.EtudeHipError("style vectors are disabled in the reference (sv_dim=0, extractor.py:107)")
        feat = torch.as_tensor(np.asarray(a_feature, dtype=np.float32)).to(self.device)
        out = self.transcript(feat, want_A=True)
        res = tuple(o.cpu().numpy() for o in out)
        self.check_feature_range()
        return res if mode == "combination" else res[:4]

    _M2N_VEL = {"ignore_zero": 0, "org": 1}
    _M2N_OFF = {"shorter": 0, "longer": 1, "offset": 2}

    def _mpe2note_array(self, a_onset, a_offset, a_mpe, a_velocity, thred_onset=0.5, thred_offset=0.5, thred_mpe=0.5,
                        mode_velocity="ignore_zero", mode_offset="shorter") -> np.ndarray:
        """extractor.py:256-418 through the C ABI; returns a structured array (onset f8, offset f8, pitch i4, velocity i4)."""
        on = np.ascontiguousarray(a_onset, np.float32)
        off = np.ascontiguousarray(a_offset, np.float32)
        mp = np.ascontiguousarray(a_mpe, np.float32)
        ve = np.ascontiguousarray(a_velocity, np.int8)
        T, nn = on.shape
        lib = _lib.lib()
        cap = max(4096, T * 8)
        f = self.config.feature
        while True:
            buf = np.empty(cap, dtype=NOTE_DTYPE)
            n = C.c_longlong()
            rc = lib.etd_mpe2note_modes(on.ctypes.data, off.ctypes.data, mp.ctypes.data, ve.ctypes.data, T, nn, thred_onset,
                                        thred_offset, thred_mpe, f.hop_sample, f.sr, self.config.midi.note_min,
                                        self._M2N_VEL[mode_velocity], self._M2N_OFF[mode_offset],
                                        C.cast(buf.ctypes.data, C.POINTER(_lib.Note)), cap, C.byref(n))
            if rc == -12 and n.value > cap:
                cap = int(n.value)
                continue
            _lib.check(rc, "etd_mpe2note")
            return buf[: n.value]

    def mpe2note_device(self, onset: torch.Tensor, offset: torch.Tensor, mpe: torch.Tensor, velocity: torch.Tensor,
                        thred_onset=0.5, thred_offset=0.5, thred_mpe=0.5, mode_velocity="ignore_zero", mode_offset="shorter") -> np.ndarray:
        """extractor.py:256-418 on the device arrays `transcript` returned (fp32 [T, n_note] x3, int8 [T, n_note]); only the
        notes cross PCIe.  Same structured array, bit for bit, as `_mpe2note_array` on the host copies, for every mode of the reference."""
        lib = _lib.lib()
        for t, dt in ((onset, torch.float32), (offset, torch.float32), (mpe, torch.float32), (velocity, torch.int8)):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous() or t.shape != onset.shape:
                raise ValueError("mpe2note_device: need contiguous device tensors [T, n_note] (fp32, fp32, fp32, int8)")
        T, nn = onset.shape
        if self._m2n is None:
            h = C.c_void_p()
            _lib.check(lib.etd_mpe2note_dev_create(nn, C.byref(h)), "etd_mpe2note_dev_create")
            self._m2n = h
        f = self.config.feature
        st = torch.cuda.current_stream(self.device).cuda_stream
        cap = max(4096, T * 2)
        while True:
            buf = np.empty(cap, dtype=NOTE_DTYPE)
            n = C.c_longlong()
            with torch.cuda.device(self.device):
                rc = lib.etd_mpe2note_dev_modes(self._m2n, onset.data_ptr(), offset.data_ptr(), mpe.data_ptr(), velocity.data_ptr(), T,
                                                thred_onset, thred_offset, thred_mpe, f.hop_sample, f.sr, self.config.midi.note_min,
                                                self._M2N_VEL[mode_velocity], self._M2N_OFF[mode_offset],
                                                C.cast(buf.ctypes.data, C.POINTER(_lib.Note)), cap, C.byref(n), st)
            if rc == -12 and n.value > cap:
                cap = int(n.value)
                continue
            _lib.check(rc, "etd_mpe2note_dev")
            return buf[: n.value]

    @staticmethod
    def _notes_from_array(arr: np.ndarray) -> List[dict]:
        return [{"pitch": p, "onset": a, "offset": b, "velocity": v}
                for p, a, b, v in zip(arr["pitch"].tolist(), arr["onset"].tolist(), arr["offset"].tolist(), arr["velocity"].tolist())]

    def _mpe2note(self, a_onset=None, a_offset=None, a_mpe=None, a_velocity=None, thred_onset=0.5, thred_offset=0.5,
                  thred_mpe=0.5, mode_velocity="ignore_zero", mode_offset="shorter") -> List[dict]:
        """extractor.py:256-418 (host C++ through the C ABI)."""
        if mode_velocity not in self._M2N_VEL or mode_offset not in self._M2N_OFF:
            raise ValueError(f"_mpe2note: unknown mode {mode_velocity!r} / {mode_offset!r}")
        return self._notes_from_array(self._mpe2note_array(a_onset, a_offset, a_mpe, a_velocity, thred_onset, thred_offset, thred_mpe,
                                                           mode_velocity, mode_offset))

    def _note2json(self, notes, path_output, min_length=0.0):
        """extractor.py:432-446."""
        filtered = []
        for note in notes:
            if note["offset"] - note["onset"] < min_length:
                continue
            filtered.append({"onset": note["onset"], "offset": note["offset"], "pitch": note["pitch"], "velocity": note["velocity"]})
        with open(path_output, "w", encoding="utf-8") as f:
            json.dump(filtered, f, ensure_ascii=False, indent=2)

    def _note2midi(self, notes, path_output, min_length=0.0):
        """extractor.py:421-429, through the native writer (csrc/midi.cpp: the file pretty_midi would write; no pretty_midi needed)."""
        kept = [n for n in notes if not (n["offset"] - n["onset"] < min_length)]
        arr = np.empty(len(kept), dtype=NOTE_DTYPE)
        for i, n in enumerate(kept):
            arr[i] = (n["onset"], n["offset"], int(n["pitch"]), int(n["velocity"]))
        _lib.check(_lib.lib().etd_midi_write(arr.ctypes.data if arr.size else None, int(arr.size), str(path_output).encode()), "etd_midi_write")

    # ------------------------------------------------------------------ device-level API
    def _front(self, sr: int) -> FrontEnd:
        if sr not in self._fronts:
            f = self.config.feature
            with torch.cuda.device(self.device):
                self._fronts[sr] = FrontEnd(sr, f.sr, f.fft_bins, f.hop_sample, f.mel_bins, f.log_offset, pad_mode=self._stft_pad_mode,
                                            win_length=f.window_length)
        return self._fronts[sr]

    def wav2feature_tensor(self, wave: Union[np.ndarray, torch.Tensor], sr: int) -> torch.Tensor:
        """[C, L] float32 (host or device) -> device [T, n_mels]."""
        w = torch.as_tensor(wave, dtype=torch.float32)
        if w.dim() == 1:
            w = w[None]
        w = w.to(self.device, non_blocking=True).contiguous()
        with torch.cuda.device(self.device):
            return self._front(sr)(w)

    def transcript(self, feat: torch.Tensor, want_A: bool = False):
        """device [T, n_bin] fp32 -> device tensors over T_pad rows (B heads; A heads first if want_A,
        in the reference's return order onset/offset/mpe/velocity A then B)."""
        assert feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous()
        T = feat.shape[0]
        self._feat_absmax = feat.abs().amax() if feat.numel() else None      # (device scalar, no sync: `check_feature_range` reads it once the outputs are on the host)
        tp = ((T + self.n_frame - 1) // self.n_frame) * self.n_frame
        B = self._alloc(tp)
        A = self._alloc(tp) if want_A else [None] * 4
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_transcript(self._h, feat.data_ptr(), T, *[t.data_ptr() for t in B],
                                                 *[(t.data_ptr() if t is not None else None) for t in A], C.c_void_p(st)),
                       "etd_transcript")
        return tuple(A) + tuple(B) if want_A else tuple(B)

    def transcript_windows(self, spec: torch.Tensor, want_A: bool = False):
        """device [B, n_bin, n_frame + 2*margin] fp32 (the model's own input) -> outputs over B*n_frame rows."""
        assert spec.is_cuda and spec.dtype == torch.float32 and spec.is_contiguous()
        nb = spec.shape[0]
        self._feat_absmax = spec.abs().amax() if spec.numel() else None
        B = self._alloc(nb * self.n_frame)
        A = self._alloc(nb * self.n_frame) if want_A else [None] * 4
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().etd_transcript_windows(self._h, spec.data_ptr(), nb, *[t.data_ptr() for t in B],
                                                         *[(t.data_ptr() if t is not None else None) for t in A], C.c_void_p(st)),
                       "etd_transcript_windows")
        return tuple(A) + tuple(B) if want_A else tuple(B)

    def check_feature_range(self) -> None:
        """The library sizes its 16-bit planes for log-mel features in [-F, F], F = max(|min_value|, 32) (include/etude_hip.h: etd_transcript): log(mel + 1e-8) >= -18.4,
        full-scale audio stays below 15, the HFT_Transformer wrapper pads with -80.  Features outside that range (e.g. a spectrogram of int16-scale samples) could overflow an
        IEEE-half plane into Inf / NaN probabilities, so the last `transcript*` call's input maximum is checked here -- by the entry points that bring results to the host
        (one scalar read behind a synchronisation they need anyway); callers of the device-level `transcript*` call it themselves."""
        mx = getattr(self, "_feat_absmax", None)
        if mx is None:
            return
        self._feat_absmax = None
        F = max(abs(float(self.config.input.min_value)), 32.0)
        v = float(mx)
        if not v <= F:
            raise ValueError(f"log-mel features reach |x| = {v:.1f}, outside the [-{F:g}, {F:g}] this extractor's planes are sized for (unnormalised audio?)")

    def _alloc(self, rows: int):
        nn = self.n_note
        return [torch.empty((rows, nn), dtype=torch.float32, device=self.device) for _ in range(3)] + \
               [torch.empty((rows, nn), dtype=torch.int8, device=self.device)]

    def debug_velocity_logits(self, buf: Optional[torch.Tensor]):
        _lib.check(_lib.lib().etd_extractor_debug_vel_logits(self._h, buf.data_ptr() if buf is not None else None), "debug")

    def debug_tap(self, stage: int, buf: Optional[torch.Tensor]):
        _lib.check(_lib.lib().etd_extractor_debug_tap(self._h, stage, buf.data_ptr() if buf is not None else None), "debug_tap")

    def extract_note_array(self, wave: Union[np.ndarray, torch.Tensor], sr: int, min_duration: Optional[float] = None) -> np.ndarray:
        """wav -> NOTE_DTYPE array: everything extract() does except the file I/O, no per-note Python objects.  With
        ``min_duration`` the ``_note2json`` filter (extractor.py:435-437) is applied: the array is then what extract.json lists."""
        feat = self.wav2feature_tensor(wave, sr)
        on, off, mpe, vel = self.transcript(feat)
        inf = self.config.infer
        arr = self.mpe2note_device(on, off, mpe, vel, inf.onset_threshold, inf.offset_threshold, inf.frame_threshold)
        self.check_feature_range()
        if min_duration is not None:
            arr = arr[~((arr["offset"] - arr["onset"]) < min_duration)]
        return arr

    def extract_notes(self, wave: Union[np.ndarray, torch.Tensor], sr: int, min_duration: Optional[float] = None) -> List[dict]:
        """`extract_note_array` as the note dicts the reference handles."""
        return self._notes_from_array(self.extract_note_array(wave, sr, min_duration))

    def extract_many(self, audio_paths: Sequence[str], output_json_paths: Sequence[str]) -> None:
        for a, o in zip(audio_paths, output_json_paths):
            self.extract(a, o)

    def close(self):
        for f in self._fronts.values():
            f.close()
        self._fronts = {}
        if getattr(self, "_m2n", None):
            _lib.lib().etd_mpe2note_dev_destroy(self._m2n)
            self._m2n = None
        if getattr(self, "_h", None):
            _lib.lib().etd_extractor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

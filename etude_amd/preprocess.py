"""Volume contour of stage 1 (infer.py:99-104) -- drop-in for etude/utils/preprocess.py:116-166.

``analyze_volume`` = librosa.load(sr=22050, mono) + librosa.feature.rms(frame = 2 * hop, hop = sr // resolution, zero-padded
centre frames) + min-max normalisation.  Here: the clip is averaged to mono and resampled on the GPU by the Extract stage's
polyphase sinc resampler (csrc/frontend.hip), the frame energies come from ``k_rms_frames``.

PARITY UNPINNED: librosa resamples with the third-party soxr library ("soxr_hq"), which is neither in /root/reference nor in
this image; a different band-limited resampler changes individual samples in the 4th digit and a 2 204-sample RMS far
less.  The oracle (oracle/mel.py: volume_contour) restates the same steps with the torchaudio-style resampler.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Union

import numpy as np
import torch

from . import _lib
from .extractor import read_wav
from .frontend import FrontEnd


class VolumeAnalyzer:
    """`volume_contour_tensor` for many clips of one sample rate: the resampler tables are built once (batch serving)."""

    def __init__(self, sr_in: int, sr: int = 22050, resolution: int = 20, device="cuda"):
        self.dev = torch.device(device)
        self.sr, self.resolution = int(sr), int(resolution)
        with torch.cuda.device(self.dev):
            self.fe = FrontEnd(int(sr_in), sr_out=int(sr), pad_mode="constant")

    def __call__(self, wave: Union[np.ndarray, torch.Tensor]) -> np.ndarray:
        w = torch.as_tensor(wave, dtype=torch.float32)
        if w.dim() == 1:
            w = w[None]
        w = w.to(self.dev).contiguous()
        with torch.cuda.device(self.dev):
            y = self.fe.resample(w)                         # mono mean + resample (no spectrogram: librosa.load does neither)
            hop = self.sr // self.resolution
            T = 1 + y.numel() // hop
            out = torch.empty(T, dtype=torch.float32, device=self.dev)
            st = torch.cuda.current_stream(self.dev).cuda_stream
            _lib.check(_lib.lib().etd_rms_frames(y.data_ptr(), y.numel(), 2 * hop, hop, out.data_ptr(), T, C.c_void_p(st)), "etd_rms_frames")
            rms = out.cpu().numpy()
        if rms.size and rms.max() > rms.min():
            return (rms - rms.min()) / (rms.max() - rms.min())
        return np.zeros_like(rms)

    def close(self):
        self.fe.close()


def volume_contour_tensor(wave: Union[np.ndarray, torch.Tensor], sr_in: int, sr: int = 22050, resolution: int = 20, device="cuda") -> np.ndarray:
    """[C, L] (or [L]) float32 audio -> normalised RMS contour, float32 [1 + L_resampled // hop]."""
    dev = torch.device(device)
    w = torch.as_tensor(wave, dtype=torch.float32)
    if w.dim() == 1:
        w = w[None]
    w = w.to(dev).contiguous()
    with torch.cuda.device(dev):
        fe = FrontEnd(int(sr_in), sr_out=int(sr), pad_mode="constant")
        y = fe.resample(w)                                  # mono mean + resample (no spectrogram)
        hop = int(sr) // int(resolution)
        frame = 2 * hop
        T = 1 + y.numel() // hop
        out = torch.empty(T, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().etd_rms_frames(y.data_ptr(), y.numel(), frame, hop, out.data_ptr(), T, C.c_void_p(st)), "etd_rms_frames")
        rms = out.cpu().numpy()
        fe.close()
    if rms.size and rms.max() > rms.min():
        return (rms - rms.min()) / (rms.max() - rms.min())
    return np.zeros_like(rms)


def analyze_volume(audio_path: Union[str, Path], sr: int = 22050, resolution: int = 20) -> np.ndarray:
    """Signature of etude/utils/preprocess.py:116-120."""
    if not Path(audio_path).exists():
        raise FileNotFoundError(f"Audio file not found at: {audio_path}")
    wave, sr_in = read_wav(audio_path)
    return volume_contour_tensor(wave, sr_in, sr, resolution)


def save_volume_map(volume_map: np.ndarray, output_path: Union[str, Path]):
    """etude/utils/preprocess.py:154-166."""
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    with open(output_path, "w") as f:
        json.dump(np.asarray(volume_map).tolist(), f)

"""Oracle: audio front end (mono mix -> resample -> STFT -> mel -> log).  TEST INFRASTRUCTURE ONLY.

**PARITY UNPINNED.**  The arithmetic of ``AMTAPC_Extractor._wav2feature``
(etude/data/extractor.py:178-197) lives in torchaudio==2.6.0 (requirements.txt:4), which is neither
vendored under /root/reference nor installed in this image, and the reference holds no test or
fixture for it.  This file restates torchaudio 2.6's published algorithm for the exact calls the
reference makes:

  * ``torch.mean(wave, dim=0)``                                            extractor.py:181
  * ``transforms.Resample(sr, 16000)`` -> ``functional._get_sinc_resample_kernel`` +
    ``_apply_sinc_resample_kernel`` (sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99;
    kernel built in float64 then cast to float32; conv1d with stride orig/gcd)   extractor.py:183-184
  * ``transforms.MelSpectrogram(sample_rate, n_fft, win_length, hop_length, n_mels, norm="slaney")``
    = ``torch.stft(center=True, pad_mode="reflect", window=hann_window(periodic), onesided)``,
    ``|X|**2``, ``melscale_fbanks(n_freqs, 0, sr//2, n_mels, sr, "slaney", "htk")``     extractor.py:186-194
  * ``log(mel + log_offset).T``                                            extractor.py:196-197

It is self-checked (tests/test_oracle_mel.py) against torch.stft and analytic signals only.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional._get_sinc_resample_kernel (hann window).  Returns (kernel [new, width*2+orig] fp32, width, orig, new)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=None)[:, None, None] / new + idx
    t = t * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=t.dtype), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32)[:, 0, :].contiguous(), width, orig, new


def resample(wave: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """transforms.Resample.forward on a 1-D waveform."""
    if int(orig_freq) == int(new_freq):
        return wave
    k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    length = wave.shape[-1]
    w = torch.nn.functional.pad(wave[None, None, :], (width, width + orig))
    res = torch.nn.functional.conv1d(w, k[:, None, :], stride=orig)        # [1, new, n]
    res = res.transpose(1, 2).reshape(-1)
    target = int(math.ceil(new * length / orig))
    return res[:target]


def hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="htk") -> [n_freqs, n_mels] fp32."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return fb * enorm.unsqueeze(0)


def log_mel(wave16k: torch.Tensor, sr: int = 16000, n_fft: int = 2048, win_length: int = 2048, hop: int = 256,
            n_mels: int = 256, log_offset: float = 1e-8, pad_mode: str = "reflect") -> torch.Tensor:
    """MelSpectrogram + log, transposed to [T, n_mels].  pad_mode "reflect" = torchaudio's default (extractor.py:186-193);
    "constant" = what HFT_Transformer passes (hft_transformer.py:124-131)."""
    window = torch.hann_window(win_length, periodic=True)
    spec = torch.stft(wave16k, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window, center=True,
                      pad_mode=pad_mode, normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)                                             # [n_freqs, T]
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, float(sr // 2), n_mels, sr)
    mel = torch.matmul(power.transpose(-1, -2), fb)                          # [T, n_mels]
    return torch.log(mel + log_offset)


def wav2feature(wave: torch.Tensor, sr: int, target_sr: int = 16000, **kw) -> torch.Tensor:
    """extractor.py:178-197 minus the file read.  wave [C, L] fp32 -> [T, n_mels] fp32."""
    mono = torch.mean(wave, dim=0)
    return log_mel(resample(mono, sr, target_sr), sr=target_sr, **kw)


def feature_frames(n_samples_in: int, sr: int, target_sr: int = 16000, hop: int = 256) -> int:
    g = math.gcd(sr, target_sr)
    n16 = int(math.ceil((target_sr // g) * n_samples_in / (sr // g))) if sr != target_sr else n_samples_in
    return 1 + n16 // hop


def volume_contour(wave: torch.Tensor, sr_in: int, sr: int = 22050, resolution: int = 20) -> torch.Tensor:
    """analyze_volume, etude/utils/preprocess.py:116-152, on an in-memory clip [C, L]: mono mean, resample to `sr`,
    librosa.feature.rms(frame_length = 2 * hop, hop_length = sr // resolution, center=True with zero padding), min-max
    normalisation.  PARITY UNPINNED at the resampler: librosa.load uses soxr ("soxr_hq", third-party, absent here); this
    restatement uses the torchaudio-style sinc resampler of `resample` above."""
    y = resample(torch.mean(wave, dim=0), sr_in, sr) if sr_in != sr else torch.mean(wave, dim=0)
    hop = sr // resolution
    frame = 2 * hop
    yp = torch.nn.functional.pad(y, (frame // 2, frame // 2))
    T = 1 + y.numel() // hop
    frames = yp.unfold(0, frame, hop)[:T]
    rms = torch.sqrt(torch.mean(frames * frames, dim=1))
    if rms.numel() and float(rms.max()) > float(rms.min()):
        return (rms - rms.min()) / (rms.max() - rms.min())
    return torch.zeros_like(rms)

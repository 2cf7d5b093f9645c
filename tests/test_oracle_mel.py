"""Self-checks of the front-end oracle (parity UNPINNED: torchaudio is absent, see oracle/mel.py)."""
import math

import numpy as np
import torch

from oracle import mel


def test_resample_length_and_tone_frequency():
    sr, f0 = 44100, 1000.0
    n = 44100
    x = torch.sin(2 * math.pi * f0 * torch.arange(n) / sr).float()
    y = mel.resample(x, sr, 16000)
    assert y.numel() == math.ceil(160 * n / 441) == 16000
    spec = torch.fft.rfft(y[2000:2000 + 8192] * torch.hann_window(8192))
    assert abs(int(spec.abs().argmax()) * 16000 / 8192 - f0) < 2.0
    assert abs(float(y[1000:15000].abs().max()) - 1.0) < 0.02          # passband gain ~ 1
    z = mel.resample(x, 16000, 16000)
    assert z is x                                                       # identity when rates match


def test_resample_rejects_above_nyquist():
    sr = 44100
    x = torch.sin(2 * math.pi * 12000.0 * torch.arange(sr) / sr).float()
    y = mel.resample(x, sr, 16000)
    assert float(y[1000:15000].abs().max()) < 0.02


def test_logmel_shape_floor_and_peak_bin():
    x = 0.5 * torch.sin(2 * math.pi * 440.0 * torch.arange(32000) / 16000).float()
    f = mel.log_mel(x)
    assert f.shape == (1 + 32000 // 256, 256)
    assert mel.feature_frames(88200, 44100) == 1 + 32000 // 256
    # 440 Hz -> HTK mel bin index m with f_pts[m+1] closest to 440
    m_pts = torch.linspace(0, mel.hz_to_mel_htk(8000.0), 258)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    want = int((f_pts[1:-1] - 440.0).abs().argmin())
    got = int(f[60].argmax())
    assert abs(got - want) <= 1
    silent = mel.log_mel(torch.zeros(16000))
    assert torch.allclose(silent, torch.full_like(silent, math.log(1e-8)))     # the -18.42 floor


def test_melbank_slaney_normalisation():
    fb = mel.melscale_fbanks(1025, 0.0, 8000.0, 256, 16000)
    assert fb.shape == (1025, 256) and float(fb.min()) >= 0.0
    # Slaney norm: each triangle integrates to ~1 over frequency (bin spacing 8000/1024 Hz) where it spans >= 3 bins
    area = fb.sum(0) * (8000.0 / 1024)
    wide = (fb > 0).sum(0) >= 6
    assert torch.allclose(area[wide], torch.ones_like(area[wide]), atol=0.08)

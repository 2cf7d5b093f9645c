"""Host side of the audio front end: builds torchaudio's tables, drives etd_frontend_run.

Mirrors ``AMTAPC_Extractor._wav2feature`` (etude/data/extractor.py:178-197).  The tables are built
with the same torch-CPU operations torchaudio 2.6 uses (``functional._get_sinc_resample_kernel``,
``torch.hann_window``, ``functional.melscale_fbanks(norm="slaney", mel_scale="htk")``), so the constants
are the ones the reference's dependency would have produced; all per-sample arithmetic then runs in
the HIP kernels (csrc/frontend.hip).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def _resample_table(sr_in: int, sr_out: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=None)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.tensor(1.0, dtype=t.dtype), t.sin() / t) * window * (base / orig)
    k = k.to(torch.float32)[:, 0, :]                                   # [new, 2*width + orig]
    return np.ascontiguousarray(k.t().numpy()), width, orig, new       # k-major [K][new]


def _mel_csr(n_freqs: int, f_max: float, n_mels: int, sample_rate: int):
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)     # slaney
    fb = fb.numpy()                                                             # [n_freqs, n_mels]
    start = np.zeros(n_mels, np.int32)
    length = np.zeros(n_mels, np.int32)
    w = []
    for m in range(n_mels):
        nz = np.flatnonzero(fb[:, m])
        if nz.size:
            start[m], length[m] = nz[0], nz[-1] - nz[0] + 1
            w.append(fb[nz[0]:nz[-1] + 1, m])
    wcat = np.concatenate(w).astype(np.float32) if w else np.zeros(1, np.float32)
    return start, length, wcat


class FrontEnd:
    """wav (device, planar [C][L] fp32) -> log-mel features (device, [T][n_mels] fp32)."""

    def __init__(self, sr_in: int, sr_out: int = 16000, n_fft: int = 2048, hop: int = 256, n_mels: int = 256,
                 log_offset: float = 1e-8, pad_mode: str = "reflect", win_length: int | None = None):
        self.sr_in, self.sr_out, self.hop, self.n_mels = int(sr_in), int(sr_out), hop, n_mels
        win_length = int(win_length) if win_length else int(n_fft)
        if win_length < 1 or win_length > n_fft:
            raise _lib.EtudeHipError(f"FrontEnd: win_length {win_length} must be in [1, n_fft={n_fft}] (torch.stft's rule)")
        lib = _lib.lib()
        if self.sr_in != self.sr_out:
            kt, width, orig, new = _resample_table(self.sr_in, self.sr_out)
        else:
            kt, width, orig, new = np.zeros((1, 1), np.float32), 0, 1, 1
        # MelSpectrogram(win_length=w) -> torch.stft: a periodic Hann window of w samples, zero-padded on both sides to n_fft
        # (extractor.py:186-193 passes win_length=feature.window_length)
        win = np.zeros(n_fft, np.float32)
        lpad = (n_fft - win_length) // 2
        win[lpad:lpad + win_length] = torch.hann_window(win_length, periodic=True).numpy().astype(np.float32)
        ms, ml, mw = _mel_csr(n_fft // 2 + 1, float(self.sr_out // 2), n_mels, self.sr_out)
        h = C.c_void_p()
        _lib.check(lib.etd_frontend_create(self.sr_in, self.sr_out, orig, new, kt.shape[0], width, kt.ctypes.data, n_fft, hop,
                                           win.ctypes.data, n_mels, ms.ctypes.data, ml.ctypes.data, mw.ctypes.data,
                                           log_offset, C.byref(h)), "etd_frontend_create")
        self._h = h
        if pad_mode not in ("reflect", "constant"):
            raise ValueError(f"pad_mode {pad_mode!r}: only 'reflect' and 'constant' are implemented")
        _lib.check(lib.etd_frontend_set_pad_mode(h, 1 if pad_mode == "constant" else 0), "etd_frontend_set_pad_mode")

    def num_frames(self, n_in: int) -> int:
        return int(_lib.lib().etd_frontend_num_frames(self._h, n_in))

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        lib = _lib.lib()
        c, n = wav.shape
        n16 = int(lib.etd_frontend_resampled_len(self._h, n))
        T = int(lib.etd_frontend_num_frames(self._h, n))
        res = torch.empty(n16, dtype=torch.float32, device=wav.device)
        feat = torch.empty((T, self.n_mels), dtype=torch.float32, device=wav.device)
        out_t = C.c_longlong()
        st = torch.cuda.current_stream(wav.device).cuda_stream
        _lib.check(lib.etd_frontend_run(self._h, wav.data_ptr(), c, n, res.data_ptr(), feat.data_ptr(), T, C.byref(out_t),
                                        C.c_void_p(st)), "etd_frontend_run")
        self.last_resampled = res
        return feat

    def resample(self, wav: torch.Tensor) -> torch.Tensor:
        """channel mean + resample only: wav (device, [C][L] fp32) -> mono [L'] fp32 at sr_out (no STFT / mel)"""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        lib = _lib.lib()
        c, n = wav.shape
        res = torch.empty(int(lib.etd_frontend_resampled_len(self._h, n)), dtype=torch.float32, device=wav.device)
        st = torch.cuda.current_stream(wav.device).cuda_stream
        _lib.check(lib.etd_frontend_run(self._h, wav.data_ptr(), c, n, res.data_ptr(), None, 0, None, C.c_void_p(st)), "etd_frontend_run")
        return res

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().etd_frontend_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""GPU front end (resample + STFT + mel + log) against the CPU oracle of torchaudio's algorithm."""
import numpy as np
import pytest
import torch

from etude_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _run(wav, sr, dev):
    from etude_amd.frontend import FrontEnd
    fe = FrontEnd(sr)
    feat = fe(torch.from_numpy(wav).to(dev))
    res = fe.last_resampled
    torch.cuda.synchronize()
    return feat.cpu().numpy(), res.cpu().numpy()


@pytest.mark.parametrize("sr,seconds,channels", [(44100, 2.0, 2), (48000, 1.3, 1), (16000, 1.0, 2), (22050, 0.7, 2)])
def test_frontend_matches_oracle(dev, sr, seconds, channels):
    from oracle import mel
    wav = synth.clip_audio(seed=sr % 97, seconds=seconds, sr=sr)[:channels]
    wav = np.ascontiguousarray(wav)
    feat, res = _run(wav, sr, dev)
    ref_res = mel.resample(torch.mean(torch.from_numpy(wav), 0), sr, 16000).numpy()
    ref = mel.wav2feature(torch.from_numpy(wav), sr).numpy()
    assert res.shape == ref_res.shape and feat.shape == ref.shape == (mel.feature_frames(wav.shape[1], sr), 256)
    # fp32 arithmetic, different summation order: resampler 475-tap dot products, 2048-point FFT
    np.testing.assert_allclose(res, ref_res, rtol=0, atol=2e-6)
    # log-mel: relative error of the power ~1e-5 -> abs error in the log; bins at the 1e-8 floor are exact
    assert np.abs(feat - ref).max() < 2e-3
    assert np.abs(feat - ref).mean() < 2e-5


def test_frontend_silence_hits_log_floor(dev):
    feat, _ = _run(np.zeros((2, 44100), np.float32), 44100, dev)
    assert np.allclose(feat, np.log(np.float32(1e-8)))


def test_frontend_rejects_too_short_clip(dev):
    from etude_amd import _lib
    from etude_amd.frontend import FrontEnd
    fe = FrontEnd(44100)
    with pytest.raises(_lib.EtudeHipError):
        fe(torch.zeros((1, 1000), device=dev))           # < n_fft/2 samples after resampling: reflect pad undefined

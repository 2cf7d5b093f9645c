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


def test_volume_contour_matches_oracle(dev):
    """analyze_volume (stage 1's volume map): GPU mono + resample + frame RMS vs the oracle's restatement"""
    from etude_amd.preprocess import volume_contour_tensor
    from oracle import mel
    wav = synth.clip_audio(seed=9, seconds=4.0)
    wav = wav * np.linspace(0.1, 1.0, wav.shape[1], dtype=np.float32)[None]          # a crescendo: the contour must rise
    got = volume_contour_tensor(wav, 44100)
    want = mel.volume_contour(torch.from_numpy(wav), 44100).numpy()
    assert got.shape == want.shape == (1 + (wav.shape[1] // 2) // 1102,)
    assert got.dtype == np.float32 and got.min() == 0.0 and got.max() == 1.0
    assert np.abs(got - want).max() < 1e-4
    assert got[-20:].mean() > got[:20].mean() + 0.3
    assert not volume_contour_tensor(np.zeros((2, 44100), np.float32), 44100).any()   # silence -> zeros (preprocess.py:147-149)


def test_analyze_volume_file_surface(dev, tmp_path):
    from etude_amd.extractor import write_wav_f32
    from etude_amd.preprocess import analyze_volume, save_volume_map
    import json
    wav = synth.clip_audio(seed=2, seconds=1.0)
    write_wav_f32(tmp_path / "a.wav", wav, 44100)
    v = analyze_volume(tmp_path / "a.wav")
    save_volume_map(v, tmp_path / "out" / "volume.json")
    assert json.loads((tmp_path / "out" / "volume.json").read_text()) == v.tolist()
    with pytest.raises(FileNotFoundError):
        analyze_volume(tmp_path / "missing.wav")


def test_frontend_against_real_torchaudio_when_present(dev):
    """The oracle's front end restates torchaudio's published algorithm because torchaudio is neither vendored by the reference
    nor installed in the build image (parity unpinned).  Where a box does have it, pin the HIP front end to the
    real thing: the exact calls of etude/data/extractor.py:181-197."""
    torchaudio = pytest.importorskip("torchaudio")
    wav = synth.clip_audio(seed=3, seconds=2.0)
    w = torch.mean(torch.from_numpy(wav), dim=0)
    w = torchaudio.transforms.Resample(44100, 16000)(w)
    ms = torchaudio.transforms.MelSpectrogram(sample_rate=16000, n_fft=2048, win_length=2048, hop_length=256, n_mels=256, norm="slaney")(w)
    want = torch.log(ms + 1e-8).T.numpy()
    feat, res = _run(wav, 44100, dev)
    np.testing.assert_allclose(res, w.numpy(), rtol=0, atol=2e-6)
    assert feat.shape == want.shape and np.abs(feat - want).max() < 2e-3


def test_frontend_window_shorter_than_fft(dev):
    """feature.window_length != feature.fft_bins: MelSpectrogram(win_length=w) centres a w-sample Hann window in the n_fft frame
    (torch.stft); extractor.py:186-193 passes the config's value."""
    from etude_amd import _lib
    from etude_amd.frontend import FrontEnd
    from oracle import mel
    wav = synth.clip_audio(seed=12, seconds=1.0, sr=16000)
    fe = FrontEnd(16000, win_length=1024)
    feat = fe(torch.from_numpy(wav).to(dev)).cpu().numpy()
    ref = mel.log_mel(torch.mean(torch.from_numpy(wav), 0), win_length=1024).numpy()
    if ref.shape != feat.shape:
        ref = ref.T
    assert feat.shape == ref.shape and np.abs(feat - ref).max() < 2e-3
    with pytest.raises(_lib.EtudeHipError):
        FrontEnd(16000, win_length=4096)

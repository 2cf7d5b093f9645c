"""HFT_Transformer wrapper on the GPU: num_frame=128, half-overlapping windows, min_value=-80, zero
STFT padding -- against vectors captured from the reference class and against the oracle.  Tolerances as in
test_gpu_extractor.py (bf16 compute vs the fp32 reference: 8e-2 max / 6e-3 mean), except that frames whose receptive field
contains the wrapper's -80 padding rows get 1e-1: the padding value lies 70 units outside the data range the embedding's
bf16 operands are centred on (measured: 0.086 on 2 of 192 rows at the clip end, 0.059 elsewhere)."""
import json

import numpy as np
import pytest
import torch

from etude_amd import synth
from etude_amd.config import HFTConfig

from tests._util import EXT_P_MEAN as P_MEAN, EXT_P_TOL as P_TOL, EXT_P_TOL_PAD as P_TOL_PAD, close_to

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _wrapper(seed=7, **over):
    from etude_amd.hft_transformer import HFT_Transformer
    cfg = HFTConfig()
    for k, v in over.items():
        setattr(cfg.input, k, v)
    return HFT_Transformer(cfg, synth.extractor_state_dict(seed, dict(n_frame=cfg.input.num_frame)), "cuda")


def test_transcript_stride_against_reference_golden(dev, golden_dir):
    g = np.load(golden_dir / "hft_wrapper_full.npz")
    tr = _wrapper(7)
    out = tr._transcript_stride(g["feature"], 32)
    assert out[0].shape == g["onset_B"].shape == (192, 88)        # 150 frames -> 3 half-windows of 64
    inner = slice(32, 118)                                         # frames >= 32 away from both clip ends (margin = 32 frames)
    for name, i in (("onset_B", 4), ("offset_B", 5), ("mpe_B", 6)):
        close_to(out[i][inner], g[name][inner], P_TOL, None, "wrapper inner frames " + name)
        close_to(out[i], g[name], P_TOL_PAD, P_MEAN, "wrapper all frames " + name)
    close_to(out[0], g["onset_A"].astype(np.float32), P_TOL_PAD, None, "wrapper onset_A")
    close_to(out[2], g["mpe_A"].astype(np.float32), P_TOL_PAD, None, "wrapper mpe_A")
    assert (out[7] == g["velocity_B"]).mean() > 0.9


def test_plain_transcript_against_oracle(dev):
    from oracle import hft
    tr = _wrapper(3)
    rng = np.random.default_rng(2)
    feat = np.clip(rng.normal(-8, 2, (200, 256)), -18, 5).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in synth.extractor_state_dict(3, dict(n_frame=128)).items()}
    want = hft.transcript(sd, feat, hft.HftDims(n_frame=128), min_value=-80.0)
    got = tr._transcript(feat)
    assert got[4].shape == want[4].shape == (256, 88)
    for i in (4, 5, 6):
        close_to(got[i], want[i], P_TOL_PAD, P_MEAN, f"wrapper plain transcript output {i}")


def test_constant_pad_front_end_matches_oracle(dev):
    from etude_amd.frontend import FrontEnd
    from oracle import mel
    wav = synth.clip_audio(seed=3, seconds=1.5)
    fe = FrontEnd(44100, pad_mode="constant")
    feat = fe(torch.from_numpy(wav).to(dev)).cpu().numpy()
    ref = mel.wav2feature(torch.from_numpy(wav), 44100, pad_mode="constant").numpy()
    assert feat.shape == ref.shape
    assert np.abs(feat - ref).max() < 2e-3 and np.abs(feat - ref).mean() < 2e-5
    refl = mel.wav2feature(torch.from_numpy(wav), 44100).numpy()
    assert np.abs(ref[0] - refl[0]).max() > 1e-2               # the two padding modes do differ at the clip edge
    short = fe(torch.zeros((1, 500), device=dev))              # no reflect restriction on very short clips
    assert short.shape[1] == 256


def test_transcribe_writes_the_reference_json(dev, tmp_path):
    from etude_amd.extractor import write_wav_f32
    from oracle import mpe2note
    tr = _wrapper(9)
    wav = synth.clip_audio(seed=6, seconds=2.5)
    write_wav_f32(tmp_path / "in.wav", wav, 44100)
    tr.transcribe(tmp_path / "in.wav", tmp_path / "out" / "notes.json")
    text = (tmp_path / "out" / "notes.json").read_text()
    notes = json.loads(text)
    assert text.startswith("[\n    {") or notes == []          # indent=4 (hft_transformer.py:116)
    feat = tr._wav2feature(tmp_path / "in.wav")
    outs = tr._transcript_stride(feat, 32)
    assert notes == mpe2note.mpe2note(outs[4], outs[5], outs[6], outs[7], 0.75, 0.5, 0.5)     # no duration filter here
    assert all(set(n) == {"onset", "offset", "pitch", "velocity"} for n in notes)

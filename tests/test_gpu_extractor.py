"""HIP hFT-Transformer against the oracle / the reference's golden vectors, through the C ABI.

Tolerance (stated per north_star; the constants and the measured values live in tests/_util.py): the 16-bit serving mode computes with
IEEE-half operands (fp32 accumulate, fp32 LayerNorm / softmax / sigmoid); the reference is fp32.  Measured on MI355X over the seeds used
here: probabilities differ by <= 1.94e-2 (mean <= 4.1e-4), velocity logits by <= 1.8e-2 (rounds 1-4, bf16 operands: 5.5e-2 / 3e-3 / 0.15).
The tests allow 3e-2 max / 1e-3 mean on probabilities, 0.06 on logits, and require every velocity argmax to be a near-maximum of the
ORACLE's logits (within 0.06)."""
import json

import numpy as np
import pytest
import torch

from etude_amd import synth
from etude_amd.config import ExtractorConfig

from tests._util import EXT_L_TOL as L_TOL, EXT_P_MEAN as P_MEAN, EXT_P_TOL as P_TOL, close_to

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _extractor(nf, seed=7, **kw):
    from etude_amd.extractor import AMTAPC_Extractor
    cfg = ExtractorConfig()
    cfg.input.num_frame = nf
    return AMTAPC_Extractor(cfg, synth.extractor_state_dict(seed, dict(n_frame=nf)), "cuda", **kw)


def _oracle(nf, seed=7):
    from oracle import hft
    sd = {k: torch.from_numpy(v) for k, v in synth.extractor_state_dict(seed, dict(n_frame=nf)).items()}
    return sd, hft.HftDims(n_frame=nf)


def test_full_size_window_against_reference_golden(dev, golden_dir):
    g = np.load(golden_dir / "hft_full.npz")
    ex = _extractor(512)
    x = torch.from_numpy(synth.window_features(5, 1)).to(dev)
    vl = torch.zeros((512, 88, 128), dtype=torch.float32, device=dev)
    ex.debug_velocity_logits(vl)
    oA, fA, mA, vA, on, off, mpe, vel = [t.cpu().numpy() for t in ex.transcript_windows(x, want_A=True)]
    ex.debug_velocity_logits(None)
    for name, got in (("onset_B", on), ("offset_B", off), ("mpe_B", mpe)):
        close_to(got, g[name], P_TOL, P_MEAN, "hft_full " + name)
    close_to(oA, g["onset_A"].astype(np.float32), P_TOL, None, "hft_full onset_A")
    close_to(mA, g["mpe_A"].astype(np.float32), P_TOL, None, "hft_full mpe_A")
    close_to(vl.cpu().numpy()[::64], g["velocity_B_rows"], L_TOL, None, "hft_full velocity logits")
    clear = g["velocity_B_top2gap"].astype(np.float32) > 2 * L_TOL
    assert (vel == g["velocity_B_argmax"])[clear].all()
    assert (vel == g["velocity_B_argmax"]).mean() > 0.97
    assert on.min() >= 0 and on.max() <= 1 and np.isfinite(off).all()


P_TOL_CAL, P_MEAN_CAL = 1e-2, 1e-3


def test_full_size_window_calibrated_weights_16bit(dev, golden_dir):
    """(Numbers in this paragraph: rounds 1-4, bf16 operands.  With IEEE-half operands the same window sits at 4.3e-3 max, 0.06 % of the cells across 0.5,
    velocity argmax agreement 0.998.)  A second reference golden for the 16-bit extractor, on a checkpoint whose FIRST encoder layer is well conditioned.  synth.extractor_state_dict feeds that
    layer x = 16 emb + pos with |x| ~ 75: its attention scores have sigma ~ 3 700 -- a hard argmax whose winner any rounding of X or K flips, which is what makes
    the 1.2 % rms the encoder taps show (tools/diag_rounding_budget.py reproduces it with emulated roundings on the oracle; a layer fed a LayerNorm output adds
    0.3 %).  synth.extractor_state_dict_cal scales the embedding so that layer 0's scores look like the other layers' (sigma ~ 3: a soft attention, the regime a
    trained checkpoint is in).  Measured: the FINAL probabilities are no closer for it (2.7e-2 max on one window, 0.4 % of the cells across 0.5, velocity argmax
    agreement 0.989 -- the benchmark checkpoint: 2.5e-2 / 0.2-0.5 % / 0.986): the end-to-end error is made behind the encoder, by the 0.5-0.8 % rms of the
    frequency / time decoder activations through output heads of gain 4 (profiles/r02_error_budget.txt), not by the first layer's artefact."""
    from etude_amd.extractor import AMTAPC_Extractor
    g = np.load(golden_dir / "hft_full_cal.npz")
    ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict_cal(7, {}), "cuda")
    x = torch.from_numpy(synth.window_features(5, 1)).to(dev)
    oA, fA, mA, vA, on, off, mpe, vel = [t.cpu().numpy() for t in ex.transcript_windows(x, want_A=True)]
    worst = flips = 0.0
    for name, got in (("onset_B", on), ("offset_B", off), ("mpe_B", mpe)):
        err = np.abs(got - g[name])
        worst = max(worst, float(err.max()))
        flips = max(flips, float(((got > 0.5) != (g[name] > 0.5)).mean()))
        assert err.max() < P_TOL_CAL and err.mean() < P_MEAN_CAL, (name, float(err.max()), float(err.mean()))
    agree = float((vel == g["velocity_B_argmax"]).mean())
    print(f"16-bit mode, calibrated checkpoint, one 512-frame window: max |p - reference| = {worst:.2e}, cells across 0.5: {flips:.5f}, velocity argmax agreement {agree:.4f}")
    assert flips < 1.5e-3
    assert np.abs(oA - g["onset_A"].astype(np.float32)).max() < P_TOL_CAL and np.abs(mA - g["mpe_A"].astype(np.float32)).max() < P_TOL_CAL
    clear = g["velocity_B_top2gap"].astype(np.float32) > 0.1
    assert (vel == g["velocity_B_argmax"])[clear].all() and agree > 0.995
    ex.close()


def test_per_stage_taps_against_oracle(dev):
    from oracle import hft
    nf = 64
    ex = _extractor(nf, chunk_frames=nf)
    sd, d = _oracle(nf)
    x = synth.window_features(11, 1, 256, nf + 64)
    taps = {}
    hft.model_forward(sd, torch.from_numpy(x), d, taps)
    rows = {0: nf * 256, 3: nf * 256, 6: nf * 88, 7: 88 * nf, 10: 88 * nf}
    names = {0: "embed", 3: "enc2", 6: "dec2", 7: "time_in", 10: "time2"}
    bufs = {s: torch.zeros((r, 256), dtype=ex.operand_dtype, device=dev) for s, r in rows.items()}
    for s, b in bufs.items():
        ex.debug_tap(s, b)
    ex.transcript_windows(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    for s, b in bufs.items():
        ref = taps[names[s]].numpy().reshape(-1, 256)
        got = b.float().cpu().numpy()
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        relm = np.abs(got - ref).mean() / np.abs(ref).mean()
        print(f"[measured] tap {names[s]}: max-abs error / max-abs value {rel:.3e}, mean error / mean value {relm:.3e}")
        assert rel < 0.12, (names[s], rel)                 # LayerNorm'ed activations, 16-bit storage; the first encoder layer's hard-argmax attention
        assert relm < 0.02, names[s]


def test_transcript_ragged_matches_oracle_and_is_chunk_invariant(dev):
    """T=150 frames at n_frame=64 -> 3 windows, last one ragged (extractor.py:210-228 padding)."""
    from oracle import hft
    nf = 64
    sd, d = _oracle(nf)
    rng = np.random.default_rng(3)
    feat = np.clip(rng.normal(-8, 2, (150, 256)), -18, 5).astype(np.float32)
    ref, ref_vl = hft.transcript(sd, feat, d, return_vel_logits=True)
    outs = {}
    for kw in (dict(max_windows=1, chunk_frames=32), dict(max_windows=2, chunk_frames=64), dict(max_windows=4, chunk_frames=64)):
        ex = _extractor(nf, **kw)
        got = ex._transcript(feat)                       # the reference's own method name/signature
        assert len(got) == 8 and got[0].shape == (192, 88) and got[3].dtype == np.int8
        outs[tuple(kw.values())] = got
        for i in (0, 1, 2, 4, 5, 6):
            close_to(got[i], ref[i], P_TOL, P_MEAN, f"ragged transcript {kw} output {i}")
        chosen = np.take_along_axis(ref_vl, got[7].astype(np.int64)[..., None], -1)[..., 0]
        assert (ref_vl.max(-1) - chosen).max() < L_TOL
        ex.close()
    a, b, c = outs.values()
    for i in range(8):     # batching / chunking must not change a single bit
        assert np.array_equal(a[i], b[i]) and np.array_equal(a[i], c[i]), i


def test_window_batch_invariance_and_determinism(dev):
    nf = 64
    ex = _extractor(nf, max_windows=3)
    x = torch.from_numpy(synth.window_features(2, 3, 256, nf + 64)).to(dev)
    full = [t.cpu().numpy() for t in ex.transcript_windows(x)]
    again = [t.cpu().numpy() for t in ex.transcript_windows(x)]
    for a, b in zip(full, again):
        assert np.array_equal(a, b)
    for w in range(3):
        one = [t.cpu().numpy() for t in ex.transcript_windows(x[w:w + 1].contiguous())]
        for a, b in zip(full, one):
            assert np.array_equal(a[w * nf:(w + 1) * nf], b)


def test_out_of_range_features_and_checkpoints_are_refused(dev):
    """The 16-bit planes (IEEE half; the exact-parity mode's two-plane splits) are sized for log-mel features in [-F, F], F = max(|min_value|, 32) (include/etude_hip.h,
    etd_transcript): the host mirror refuses features outside it instead of returning Inf / NaN probabilities, and the 16-bit mode refuses a checkpoint whose first encoder layer
    input could leave the half range for such features."""
    from etude_amd import _lib
    from etude_amd.extractor import AMTAPC_Extractor
    nf = 64

    def _cfg(n):
        c = ExtractorConfig()
        c.input.num_frame = n
        return c
    for prec in ("f16", "fp32"):
        ex = AMTAPC_Extractor(_cfg(nf), synth.extractor_state_dict(7, dict(n_frame=nf)), "cuda", precision=prec)
        feat = np.clip(np.random.default_rng(1).normal(-8, 2, (70, 256)), -18, 5).astype(np.float32)
        assert np.isfinite(ex._transcript(feat)[4]).all()
        with pytest.raises(ValueError, match="outside the"):
            ex._transcript(feat * 1000.0)
        ex.close()
    sd = {k: v.copy() for k, v in synth.extractor_state_dict(7, dict(n_frame=nf)).items()}
    sd["encoder.tok_embedding_freq.weight"] *= 100.0
    with pytest.raises(_lib.EtudeHipError, match="IEEE-half range"):
        AMTAPC_Extractor(_cfg(nf), sd, "cuda", precision="f16")


def test_unsupported_architecture_fails_loudly(dev):
    from etude_amd import _lib
    from etude_amd.extractor import AMTAPC_Extractor
    cfg = ExtractorConfig()
    cfg.model.transformer_hid_dim = 128          # 4 heads of 32: neither engine has that (other architectures with head_dim 64: tests/test_gpu_extractor_archs.py)
    with pytest.raises(_lib.EtudeHipError, match="unsupported architecture"):
        AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), "cuda", precision="f16")
    with pytest.raises(_lib.EtudeHipError, match="general engine needs"):
        AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), "cuda")
    cfg = ExtractorConfig()
    sd = synth.extractor_state_dict(0)
    del sd["decoder.fc_mpe_time.bias"]
    with pytest.raises(_lib.EtudeHipError, match="missing weight"):
        AMTAPC_Extractor(cfg, sd, "cuda")


def test_extract_end_to_end_writes_reference_json(dev, tmp_path):
    """wav file -> extract() -> JSON; compared with oracle mel + oracle model + oracle mpe2note on the same audio
    (n_frame=64 keeps the CPU oracle to a few seconds).  Notes are compared as sets with a time tolerance
    because a probability within P_TOL of a threshold may legitimately flip a borderline note."""
    from etude_amd.extractor import write_wav_f32
    from oracle import hft, mel, mpe2note
    nf = 64
    ex = _extractor(nf, seed=9, max_windows=4)
    wav = synth.clip_audio(seed=5, seconds=3.0)
    write_wav_f32(tmp_path / "origin.wav", wav, 44100)
    ex.extract(str(tmp_path / "origin.wav"), str(tmp_path / "extract.json"))
    notes = json.loads((tmp_path / "extract.json").read_text())
    assert isinstance(notes, list) and all(set(n) == {"onset", "offset", "pitch", "velocity"} for n in notes)
    assert all(n["offset"] - n["onset"] >= 0.08 for n in notes) and notes == sorted(notes, key=lambda n: n["onset"])
    sd = {k: torch.from_numpy(v) for k, v in synth.extractor_state_dict(9, dict(n_frame=nf)).items()}
    feat = mel.wav2feature(torch.from_numpy(wav), 44100).numpy()
    o = hft.transcript(sd, feat, hft.HftDims(n_frame=nf))
    # the device path on the same features agrees with the oracle within tolerance ...
    got = ex._transcript(feat)
    for i in (4, 5, 6):
        close_to(got[i], o[i], P_TOL, P_MEAN, f"extract end to end output {i}")
    # ... and its own notes are exactly the reference algorithm applied to its own frame outputs
    ref_notes = mpe2note.notes_for_json(mpe2note.mpe2note(got[4], got[5], got[6], got[7], 0.5, 1.0, 0.5), 0.08)
    dev_feat = ex.wav2feature_tensor(wav, 44100)
    on, off, mp, ve = [t.cpu().numpy() for t in ex.transcript(dev_feat)]
    assert notes == mpe2note.notes_for_json(mpe2note.mpe2note(on, off, mp, ve, 0.5, 1.0, 0.5), 0.08)
    assert abs(len(notes) - len(ref_notes)) <= max(3, len(ref_notes) // 10)


def test_short_mono_16k_clip_single_partial_window(dev, tmp_path):
    """edge of the input space: 16 kHz mono (no resampling), 0.35 s -> 22 frames, less than one window, ragged tail"""
    from etude_amd.extractor import write_wav_f32
    from oracle import hft, mel, mpe2note
    nf = 32
    ex = _extractor(nf, seed=4)
    wav = synth.clip_audio(seed=8, seconds=0.35, sr=16000)[:1]
    write_wav_f32(tmp_path / "m.wav", wav, 16000)
    ex.extract(str(tmp_path / "m.wav"), str(tmp_path / "m.json"))
    notes = json.loads((tmp_path / "m.json").read_text())
    feat = mel.wav2feature(torch.from_numpy(wav), 16000).numpy()
    assert feat.shape == (1 + wav.shape[1] // 256, 256)
    got = ex._transcript(feat)
    assert got[4].shape == (nf, 88)                               # padded to one whole window
    sd, d = _oracle(nf, seed=4)
    o = hft.transcript(sd, feat, d)
    for i in (4, 5, 6):
        close_to(got[i], o[i], P_TOL, None, f"short clip output {i}")
    dev_feat = ex.wav2feature_tensor(wav, 16000)
    assert np.abs(dev_feat.cpu().numpy() - feat).max() < 2e-3
    on, off, mp, ve = [t.cpu().numpy() for t in ex.transcript(dev_feat)]
    assert notes == mpe2note.notes_for_json(mpe2note.mpe2note(on, off, mp, ve, 0.5, 1.0, 0.5), 0.08)


def test_checkpoint_file_contract(dev, tmp_path):
    """_load_model (extractor.py:78-113): `checkpoints/extractor/latest.pth` is a FLAT state dict read with
    torch.load(weights_only=True).  The reference loads it with strict=False (extra keys ignored, missing keys silently left at
    their random init); this build ignores extra keys too but treats a MISSING key as an error (stricter on purpose: a silently
    random layer is never what a caller wants -- INTEGRATION.md section 3)."""
    from etude_amd import _lib
    from etude_amd.extractor import AMTAPC_Extractor
    nf = 32
    cfg = ExtractorConfig()
    cfg.input.num_frame = nf
    sd = synth.extractor_state_dict(7, dict(n_frame=nf))
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    tsd["encoder.some_buffer_the_reference_would_ignore"] = torch.zeros(3)
    tsd["fc_style.weight"] = torch.zeros(4, 4)
    torch.save(tsd, tmp_path / "latest.pth")
    x = torch.from_numpy(synth.window_features(5, 1, 256, nf + 64)).to(dev)
    a = AMTAPC_Extractor(cfg, str(tmp_path / "latest.pth"), "cuda")
    b = AMTAPC_Extractor(cfg, sd, "cuda")
    for p, q in zip(a.transcript_windows(x), b.transcript_windows(x)):
        assert torch.equal(p, q)
    a.close(); b.close()
    del tsd["decoder.fc_onset_time.weight"]
    torch.save(tsd, tmp_path / "bad.pth")
    with pytest.raises(_lib.EtudeHipError, match="missing weight"):
        AMTAPC_Extractor(cfg, str(tmp_path / "bad.pth"), "cuda")
    with pytest.raises(FileNotFoundError):
        AMTAPC_Extractor(cfg, str(tmp_path / "nope.pth"), "cuda")


# ------------------------------------------------------------------------------------------------ fp32 parity mode
F32_TOL = 2e-4          # fp32 end to end, exact-fp32 products: differences from the fp32 reference are summation order only


def test_fp32_parity_mode_full_window_against_reference_golden(dev, golden_dir):
    """precision="fp32" (csrc/ext_fp32.hip): the reference's own fp32 arithmetic on the device.  ONE full-size window against the
    golden the reference produced: probabilities within 2e-4 (the bf16 mode: 8e-2), velocity logits within 1e-3, every argmax
    with a top-2 gap above 2e-3 identical."""
    g = np.load(golden_dir / "hft_full.npz")
    ex = _extractor(512, precision="fp32")
    x = torch.from_numpy(synth.window_features(5, 1)).to(dev)
    vl = torch.zeros((512, 88, 128), dtype=torch.float32, device=dev)
    ex.debug_velocity_logits(vl)
    oA, fA, mA, vA, on, off, mpe, vel = [t.cpu().numpy() for t in ex.transcript_windows(x, want_A=True)]
    ex.debug_velocity_logits(None)
    worst = 0.0
    for name, got in (("onset_B", on), ("offset_B", off), ("mpe_B", mpe)):
        worst = max(worst, float(np.abs(got - g[name]).max()))
    print(f"fp32 parity mode, one 512-frame window: max |p - reference| = {worst:.2e}")
    assert worst < F32_TOL
    assert np.abs(oA - g["onset_A"].astype(np.float32)).max() < 1e-3 and np.abs(mA - g["mpe_A"].astype(np.float32)).max() < 1e-3   # (stored as fp16)
    assert np.abs(vl.cpu().numpy()[::64] - g["velocity_B_rows"]).max() < 1e-3
    clear = g["velocity_B_top2gap"].astype(np.float32) > 2e-3
    assert (vel == g["velocity_B_argmax"])[clear].all() and (vel == g["velocity_B_argmax"]).mean() > 0.9995
    ex.close()


def test_fp32_parity_mode_taps_ragged_transcript_and_notes(dev):
    """every stage of the fp32 mode against the oracle at n_frame = 64 (taps are fp32 here), then a ragged 3-window _transcript:
    frame outputs within 2e-4 and the note list IDENTICAL to the oracle's on all but threshold-grazing notes"""
    from oracle import hft, mpe2note
    nf = 64
    ex = _extractor(nf, precision="fp32")
    sd, d = _oracle(nf)
    x = synth.window_features(11, 1, 256, nf + 64)
    taps = {}
    hft.model_forward(sd, torch.from_numpy(x), d, taps)
    rows = {0: nf * 256, 1: nf * 256, 3: nf * 256, 4: nf * 88, 6: nf * 88, 7: 88 * nf, 8: 88 * nf, 10: 88 * nf}
    names = {0: "embed", 1: "enc0", 3: "enc2", 4: "dec0", 6: "dec2", 7: "time_in", 8: "time0", 10: "time2"}
    bufs = {s: torch.zeros((r, 256), dtype=torch.float32, device=dev) for s, r in rows.items()}
    for s, b in bufs.items():
        ex.debug_tap(s, b)
    ex.transcript_windows(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    for s, b in bufs.items():
        ex.debug_tap(s, None)
        ref = taps[names[s]].numpy().reshape(-1, 256)
        rel = np.abs(b.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert rel < 5e-4, (names[s], rel)            # max-abs error over max-abs value: fp32 summation order through softmax and LayerNorm
    rng = np.random.default_rng(3)
    feat = np.clip(rng.normal(-8, 2, (150, 256)), -18, 5).astype(np.float32)
    ref, ref_vl = hft.transcript(sd, feat, d, return_vel_logits=True)
    got = ex._transcript(feat)
    assert len(got) == 8 and got[0].shape == (192, 88)
    for i in (0, 1, 2, 4, 5, 6):
        assert np.abs(got[i] - ref[i]).max() < F32_TOL, i
    chosen = np.take_along_axis(ref_vl, got[7].astype(np.int64)[..., None], -1)[..., 0]
    assert (ref_vl.max(-1) - chosen).max() < 1e-3
    n_got = mpe2note.mpe2note(got[4], got[5], got[6], got[7], 0.5, 1.0, 0.5)
    n_ref = mpe2note.mpe2note(ref[4], ref[5], ref[6], ref[7], 0.5, 1.0, 0.5)
    key = lambda n: (n["pitch"], round(n["onset"] / 0.016))        # noqa: E731
    common = len({key(n) for n in n_got} & {key(n) for n in n_ref})
    assert common >= 0.995 * max(len(n_ref), 1) and abs(len(n_got) - len(n_ref)) <= max(1, len(n_ref) // 200)
    ex.close()

"""Architectures other than the default one.  The reference builds its transcriber from config dims (etude/config/schema.py:100-112 -> extractor.py:78-113 ->
amt_apc.py Encoder_SPEC2MIDI / Decoder_SPEC2MIDI); the device's general engine (csrc/ext_fp32.hip, precision "fp32") takes every such architecture with head_dim 64.
Each case below runs windows and a ragged feature stream through the C ABI and compares with the oracle (oracle/hft.py, the reference's op sequence in torch fp32)
on the same seeded weights: probabilities within 2e-4, every velocity argmax a maximum of the oracle's logits within 1e-3, the stage taps within 5e-4 relative."""
import numpy as np
import pytest
import torch

from etude_amd import synth
from etude_amd.config import ExtractorConfig

pytestmark = pytest.mark.gpu

F32_TOL = 2e-4

ARCHS = {
    # hid 128 = 2 heads, a 3 x 3 conv, two layers of each kind, 64 bins, 40 notes, 16 velocities
    "hid128": dict(n_margin=16, n_frame=48, n_bin=64, cnn_channel=3, cnn_kernel=3, hid_dim=128, pf_dim=256, n_heads=2, n_layers_enc=2, n_layers_dec=2, n_note=40, n_velocity=16),
    # hid 512 = 8 heads, one encoder layer and four decoder layers (three layers_freq), 128 bins
    "hid512": dict(n_margin=32, n_frame=32, n_bin=128, cnn_channel=4, cnn_kernel=5, hid_dim=512, pf_dim=1024, n_heads=8, n_layers_enc=1, n_layers_dec=4, n_note=88, n_velocity=128),
    # one head, no layers_freq at all (n_layers_dec = 1), odd note / frame / velocity counts, a 7-tap conv on a 17-tap window
    "hid64": dict(n_margin=8, n_frame=20, n_bin=32, cnn_channel=2, cnn_kernel=7, hid_dim=64, pf_dim=96, n_heads=1, n_layers_enc=3, n_layers_dec=1, n_note=13, n_velocity=5),
    # the default widths with other bin / margin / conv / layer numbers (the default architecture's small kernels must not be picked by mistake)
    "hid256": dict(n_margin=24, n_frame=32, n_bin=96, cnn_channel=5, cnn_kernel=4, hid_dim=256, pf_dim=512, n_heads=4, n_layers_enc=2, n_layers_dec=3, n_note=88, n_velocity=128),
    # the default architecture with another conv front end only: the specialised embedding kernel with a differently folded map
    "conv": dict(n_margin=32, n_frame=32, n_bin=256, cnn_channel=2, cnn_kernel=9, hid_dim=256, pf_dim=512, n_heads=4, n_layers_enc=3, n_layers_dec=3, n_note=88, n_velocity=128),
}


def _config(d):
    cfg = ExtractorConfig()
    cfg.input.margin_b = cfg.input.margin_f = d["n_margin"]
    cfg.input.num_frame = d["n_frame"]
    cfg.feature.n_bins = d["n_bin"]
    cfg.feature.mel_bins = d["n_bin"]
    cfg.model.cnn_channel, cfg.model.cnn_kernel = d["cnn_channel"], d["cnn_kernel"]
    cfg.model.transformer_hid_dim, cfg.model.transformer_pf_dim = d["hid_dim"], d["pf_dim"]
    cfg.model.encoder_n_head = cfg.model.decoder_n_head = d["n_heads"]
    cfg.model.encoder_n_layer, cfg.model.decoder_n_layer = d["n_layers_enc"], d["n_layers_dec"]
    cfg.midi.num_note, cfg.midi.num_velocity = d["n_note"], d["n_velocity"]
    cfg.midi.note_max = cfg.midi.note_min + d["n_note"] - 1
    return cfg


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", list(ARCHS))
def test_non_default_architecture_matches_oracle(dev, name):
    from etude_amd.extractor import AMTAPC_Extractor
    from oracle import hft
    d = synth.extractor_dims(**ARCHS[name])
    sd_np = synth.extractor_state_dict(21, d)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    od = hft.HftDims(**d)
    ex = AMTAPC_Extractor(_config(d), sd_np, "cuda", max_windows=2)      # default precision of a non-default architecture: the general engine
    assert ex.precision == "fp32" and ex.operand_dtype == torch.float32 and ex.window_flops > 0
    nf, nb, nn, nv, H = d["n_frame"], d["n_bin"], d["n_note"], d["n_velocity"], d["hid_dim"]
    # ---- two windows in the model's own input layout, A and B heads, with the stage taps of the first window
    x = synth.window_features(5, 2, nb, nf + 2 * d["n_margin"])
    taps = {}
    ref = hft.model_forward(sd, torch.from_numpy(x), od, taps)
    stage = {0: "embed", 1: "enc0", 4: "dec0", 7: "time_in", 8: "time0"}
    if d["n_layers_enc"] >= 3:
        stage[3] = "enc2"
    if d["n_layers_dec"] >= 3:
        stage[6] = "dec2"; stage[10] = "time2"
    rows = {s: (nf * nb if s < 4 else nf * nn) for s in stage}
    bufs = {s: torch.zeros((r, H), dtype=torch.float32, device=dev) for s, r in rows.items()}
    for s, b in bufs.items():
        ex.debug_tap(s, b)
    vl = torch.zeros((2 * nf, nn, nv), dtype=torch.float32, device=dev)
    ex.debug_velocity_logits(vl)
    oA, fA, mA, vA, on, off, mpe, vel = [t.cpu().numpy() for t in ex.transcript_windows(torch.from_numpy(x).to(dev), want_A=True)]
    ex.debug_velocity_logits(None)
    for s, b in bufs.items():
        ex.debug_tap(s, None)
        want = taps[stage[s]][: (nf if s < 7 else nn)].numpy().reshape(-1, H)          # first window: nf sequences (encoder / freq decoder), nn sequences (time decoder)
        rel = np.abs(b.cpu().numpy() - want).max() / np.abs(want).max()
        assert rel < 5e-4, (name, stage[s], rel)
    want = dict(oA=ref[0], fA=ref[1], mA=ref[2], on=ref[5], off=ref[6], mpe=ref[7])
    got = dict(oA=oA, fA=fA, mA=mA, on=on, off=off, mpe=mpe)
    worst = 0.0
    for k, w in want.items():
        e = float(np.abs(got[k].reshape(2, nf, nn) - w.numpy()).max())
        worst = max(worst, e)
        assert e < F32_TOL, (name, k, e)
    print(f"[measured] {name}: max |p - oracle| over six heads = {worst:.2e}")
    for v, logits in ((vel, ref[8]), (vA, ref[3])):
        lg = logits.numpy().reshape(2 * nf, nn, nv)
        chosen = np.take_along_axis(lg, v.astype(np.int64).reshape(2 * nf, nn)[..., None], -1)[..., 0]
        assert (lg.max(-1) - chosen).max() < 1e-3, name
    assert np.abs(vl.cpu().numpy() - ref[8].numpy().reshape(2 * nf, nn, nv)).max() < 1e-3
    # ---- a ragged feature stream (2.6 windows) through _transcript: padding value, window stitching
    rng = np.random.default_rng(3)
    T = int(2.6 * nf)
    feat = np.clip(rng.normal(-8, 2, (T, nb)), -18, 5).astype(np.float32)
    refs, ref_vl = hft.transcript(sd, feat, od, return_vel_logits=True)
    gots = ex._transcript(feat)
    assert len(gots) == 8 and gots[0].shape == (3 * nf, nn)
    for i in (0, 1, 2, 4, 5, 6):
        assert np.abs(gots[i] - refs[i]).max() < F32_TOL, (name, i)
    chosen = np.take_along_axis(ref_vl, gots[7].astype(np.int64)[..., None], -1)[..., 0]
    assert (ref_vl.max(-1) - chosen).max() < 1e-3
    ex.close()


def test_sixteen_bit_mode_refuses_other_architectures_and_head_dim_is_checked(dev):
    from etude_amd import _lib
    from etude_amd.extractor import AMTAPC_Extractor
    d = synth.extractor_dims(**ARCHS["hid128"])
    with pytest.raises(_lib.EtudeHipError, match="unsupported architecture for the 16-bit serving mode"):
        AMTAPC_Extractor(_config(d), synth.extractor_state_dict(0, d), "cuda", precision="f16")
    d = synth.extractor_dims(hid_dim=128, n_heads=4)              # head_dim 32
    with pytest.raises(_lib.EtudeHipError, match="general engine needs hid_dim = 64"):
        AMTAPC_Extractor(_config(d), synth.extractor_state_dict(0, d), "cuda")

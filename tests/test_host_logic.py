"""Host-side logic of the drop-in layer (no GPU): prompt assembly, vocab, wav I/O, config, front-end tables."""
import json

import numpy as np
import pytest
import torch

from etude_amd import synth
from etude_amd.decoder import ABI_ATTR_KEYS, EtudeDecoderConfig, assemble_bar_prompt, assemble_bar_prompt_np, expected_state_keys
from etude_amd.extractor import read_wav, write_wav_f32
from etude_amd.vocab import Event, Vocab
from oracle import neox


def test_prompt_assembly_matches_oracle_incl_truncation():
    rng = np.random.default_rng(0)
    d = neox.NeoxDims()
    keys = sorted(ABI_ATTR_KEYS)
    for trial in range(30):
        hist = []
        for _ in range(int(rng.integers(0, 7))):
            xs = rng.integers(4, 150, int(rng.integers(2, 200))).tolist()
            ys = rng.integers(4, 150, int(rng.integers(2, 300))).tolist()
            hist.append((xs, ys, {k: int(rng.integers(0, 3)) for k in keys}))
        x = rng.integers(4, 150, int(rng.integers(2, 120))).tolist()
        y = {k: int(rng.integers(0, 3)) for k in keys}
        a = assemble_bar_prompt(hist, x, y, keys, 4, 5, 4, 1024, 512, 0.5)
        b = neox.build_bar_prompt(hist, x, y, keys, 4, 5, d, 512, 0.5)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        assert len(a[0]) <= 513
        # the numpy twin used by the batched engine builds the same prompt
        h_np = [(np.asarray(xs, np.int32), np.asarray(ys, np.int32), np.asarray([at[k] for k in ABI_ATTR_KEYS], np.int32)) for xs, ys, at in hist]
        t, c, a4 = assemble_bar_prompt_np(h_np, np.asarray(x, np.int32), np.asarray([y[k] for k in ABI_ATTR_KEYS], np.int32), 4, 5, 4, 1024, 512, 0.5)
        assert t.tolist() == a[0] and c.tolist() == a[1]
        for j, k in enumerate(ABI_ATTR_KEYS):
            assert a4[j].tolist() == a[2][k]


def test_vocab_roundtrip_and_event_decoding(tmp_path):
    p = tmp_path / "vocab.json"
    synth.write_vocab(str(p))
    v = Vocab.load(p)
    assert len(v) == 154 and v.get_bar_bos_id() == 4 and v.get_bar_eos_id() == 5 and v.get_pad_id() == 0
    ev = v.decode_to_event(v.encode("Note_60"))
    assert ev == Event("Note", 60) and isinstance(ev.value, int)
    assert v.decode_to_event(4) == Event("Bar", "BOS")
    assert v.decode_to_event(2) == Event("<BOS>", "")
    assert v.encode("Nope_1") == v.token_to_id["<UNK>"]
    assert v.decode_sequence_to_events([4, 0, 5]) == [Event("Bar", "BOS"), Event("Bar", "EOS")]   # PAD dropped
    v.save(tmp_path / "v2.json")
    assert json.loads((tmp_path / "v2.json").read_text()) == json.loads(p.read_text())
    with pytest.raises(FileNotFoundError):
        Vocab.load(tmp_path / "missing.json")


def test_wav_roundtrip(tmp_path):
    x = synth.clip_audio(seed=1, seconds=0.25)
    write_wav_f32(tmp_path / "a.wav", x, 44100)
    y, sr = read_wav(tmp_path / "a.wav")
    assert sr == 44100 and y.shape == x.shape and np.array_equal(x, y)
    with pytest.raises(ValueError):
        (tmp_path / "b.wav").write_bytes(b"not a wav file at all")
        read_wav(tmp_path / "b.wav")


def test_decoder_config_json_contract(tmp_path):
    p = tmp_path / "etude_decoder_config.json"
    p.write_text(json.dumps(dict(synth.decoder_config_json(), architectures=["EtudeDecoder"], transformers_version="4.51.3",
                                 rotary_pct=0.25, rotary_emb_base=10000, use_parallel_residual=True, hidden_act="gelu")))
    c = EtudeDecoderConfig.from_json_file(p)
    assert (c.hidden_size, c.num_hidden_layers, c.vocab_size, c.context_num_past_xy_pairs) == (512, 8, 154, 4)
    assert c.rotary_pct == 0.25 and c.rope_theta == 10000.0 and c.layer_norm_eps == 1e-5
    keys = expected_state_keys(c)
    assert set(keys) == set(synth.decoder_state_dict(0).keys())       # == reference state_dict keys (strict-loaded in make_golden.py)


def test_frontend_tables_match_oracle_definitions():
    from etude_amd.frontend import _mel_csr, _resample_table
    from oracle import mel
    kt, width, orig, new = _resample_table(44100, 16000)
    ko, w2, o2, n2 = mel.sinc_resample_kernel(44100, 16000)
    assert (width, orig, new) == (w2, o2, n2) == (17, 441, 160) and kt.shape == (475, 160)
    assert np.array_equal(kt.T, ko.numpy())
    start, length, w = _mel_csr(1025, 8000.0, 256, 16000)
    fb = mel.melscale_fbanks(1025, 0.0, 8000.0, 256, 16000).numpy()
    dense = np.zeros_like(fb)
    p = 0
    for m in range(256):
        dense[start[m]:start[m] + length[m], m] = w[p:p + length[m]]
        p += length[m]
    assert np.array_equal(dense, fb)

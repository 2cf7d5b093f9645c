#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference; never on the GPU box).  The
reference's Python is imported read-only; ``torchaudio`` and ``pretty_midi`` (absent from this
image, imported at module top by etude/data/extractor.py:22-23) are replaced by empty stub
modules -- none of the functions exercised here touch them.  Weights are the seeded synthetic
checkpoints of ``etude_amd.synth`` (numpy PCG64), loaded into the reference's own nn.Modules with
``strict=True`` so the key names/shapes are checked against the reference as a side effect.

Outputs are DATA only (inputs + expected outputs), never reference source:
  hft_tiny.npz        tiny-config model: input, per-layer taps hashes, 8 outputs
  hft_full.npz        default-config model, ONE window: onset/offset/mpe B + velocity argmax + a few logit rows
  hft_full_cal.npz    the same window with the well-conditioned checkpoint (synth.extractor_state_dict_cal)
  transcript_tiny.npz reference ``_transcript`` on 40 frames at the tiny config (ragged: 3 windows)
  mpe2note.json       crafted + random frame arrays -> reference ``_mpe2note`` / ``_note2json`` lists
  decoder_tiny.npz    tiny GPT-NeoX config: logits for a prompt, greedy ids for a few bars
  decoder_full.npz    default config (512/8/8/2048, V=154): logits for a 64-token prompt,
                      greedy ids from the reference ``generate`` for several bars x attribute tuples
Usage:  python tests/golden/make_golden.py [--only NAME]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")
os.environ.setdefault("LOG_LEVEL", "ERROR")
# transformers probes for torchaudio via importlib; import the HF-backed decoder BEFORE stubbing.
import etude.models.etude_decoder  # noqa: E402,F401
for _m in ("torchaudio", "pretty_midi"):
    sys.modules.setdefault(_m, types.ModuleType(_m))

from etude_amd import synth  # noqa: E402

TINY_EXT = dict(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=32, pf_dim=64,
                n_heads=4, n_layers_enc=3, n_layers_dec=3, n_note=12, n_velocity=8)
TINY_DEC = dict(vocab_size=154, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                max_position_embeddings=128, attribute_emb_dim=16)
TINY_DEC_KW = dict(gain=2.0, p_eos=0.15)


def _t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def ref_extractor(dims, seed, cal=False):
    from etude.config.schema import ExtractorConfig
    from etude.data.extractor import AMTAPC_Extractor, _Spec2MIDI
    from etude.models.amt_apc import Decoder_SPEC2MIDI, Encoder_SPEC2MIDI
    d = synth.extractor_dims(**dims)
    enc = Encoder_SPEC2MIDI(d["n_margin"], d["n_frame"], d["n_bin"], d["cnn_channel"], d["cnn_kernel"], d["hid_dim"],
                            d["n_layers_enc"], d["n_heads"], d["pf_dim"], 0.1, "cpu")
    dec = Decoder_SPEC2MIDI(d["n_frame"], d["n_bin"], d["n_note"], d["n_velocity"], d["hid_dim"], d["n_layers_dec"],
                            d["n_heads"], d["pf_dim"], 0.1, "cpu")
    model = _Spec2MIDI(enc, dec, sv_dim=0)
    sd = synth.extractor_state_dict_cal(seed, dims) if cal else synth.extractor_state_dict(seed, dims)
    model.load_state_dict(_t(sd), strict=True)
    model.eval()
    cfg = ExtractorConfig()
    cfg.input.margin_b = cfg.input.margin_f = d["n_margin"]
    cfg.input.num_frame = d["n_frame"]
    cfg.feature.n_bins = cfg.feature.mel_bins = d["n_bin"]
    cfg.midi.num_note = d["n_note"]
    cfg.midi.num_velocity = d["n_velocity"]
    ex = AMTAPC_Extractor.__new__(AMTAPC_Extractor)
    ex.device = torch.device("cpu")
    ex.config = cfg
    ex.model = model
    return ex, d


def gen_hft_tiny():
    ex, d = ref_extractor(TINY_EXT, seed=11)
    x = synth.window_features(3, 2, d["n_bin"], d["n_frame"] + 2 * d["n_margin"])
    with torch.no_grad():
        r = ex.model(torch.from_numpy(x))
    np.savez_compressed(HERE / "hft_tiny.npz", x=x,
                        onset_A=r[0].numpy(), offset_A=r[1].numpy(), mpe_A=r[2].numpy(), velocity_A=r[3].numpy(),
                        attention=r[4].numpy(),
                        onset_B=r[5].numpy(), offset_B=r[6].numpy(), mpe_B=r[7].numpy(), velocity_B=r[8].numpy())


def gen_hft_full():
    ex, d = ref_extractor({}, seed=7)
    x = synth.window_features(5, 1)
    with torch.no_grad():
        r = ex.model(torch.from_numpy(x))
    np.savez_compressed(HERE / "hft_full.npz",
                        onset_A=r[0][0].numpy().astype(np.float16), mpe_A=r[2][0].numpy().astype(np.float16),
                        onset_B=r[5][0].numpy(), offset_B=r[6][0].numpy(), mpe_B=r[7][0].numpy(),
                        velocity_B_argmax=r[8][0].argmax(2).numpy().astype(np.int8),
                        velocity_B_rows=r[8][0, ::64].numpy(),       # logits of every 64th frame [8,88,128]
                        velocity_B_top2gap=(lambda t: (t[..., 0] - t[..., 1]))(torch.topk(r[8][0], 2, dim=-1).values).numpy().astype(np.float16))


def gen_hft_full_cal():
    """hft_full with the well-conditioned checkpoint (synth.extractor_state_dict_cal: first-layer attention scores with sigma ~ 3 instead of ~3 700)"""
    ex, d = ref_extractor({}, seed=7, cal=True)
    x = synth.window_features(5, 1)
    with torch.no_grad():
        r = ex.model(torch.from_numpy(x))
    np.savez_compressed(HERE / "hft_full_cal.npz",
                        onset_A=r[0][0].numpy().astype(np.float16), mpe_A=r[2][0].numpy().astype(np.float16),
                        onset_B=r[5][0].numpy(), offset_B=r[6][0].numpy(), mpe_B=r[7][0].numpy(),
                        velocity_B_argmax=r[8][0].argmax(2).numpy().astype(np.int8),
                        velocity_B_top2gap=(lambda t: (t[..., 0] - t[..., 1]))(torch.topk(r[8][0], 2, dim=-1).values).numpy().astype(np.float16))


def gen_transcript_tiny():
    ex, d = ref_extractor(TINY_EXT, seed=11)
    rng = np.random.default_rng(21)
    feat = np.clip(rng.normal(-8, 2, (40, d["n_bin"])), -18, 5).astype(np.float32)   # 40 frames -> 3 windows of 16, ragged
    out = ex._transcript(feat)
    np.savez_compressed(HERE / "transcript_tiny.npz", feature=feat, **{f"out{i}": o for i, o in enumerate(out)})


def gen_mpe2note():
    ex, _ = ref_extractor(TINY_EXT, seed=11)
    ex.config.midi.num_note = 4
    cases = []
    rng = np.random.default_rng(5)

    def run(name, on, off, mpe, vel, thr=(0.5, 1.0, 0.5), min_dur=0.08):
        on, off, mpe = (np.asarray(a, np.float32) for a in (on, off, mpe))
        vel = np.asarray(vel, np.int8)
        notes = ex._mpe2note(on, off, mpe, vel, thred_onset=thr[0], thred_offset=thr[1], thred_mpe=thr[2])
        tmp = HERE / "_tmp.json"
        ex._note2json(notes, str(tmp), min_dur)
        filt = json.loads(tmp.read_text())
        tmp.unlink()
        cases.append(dict(name=name, thr=list(thr), min_dur=min_dur, onset=on.tolist(), offset=off.tolist(),
                          mpe=mpe.tolist(), velocity=vel.tolist(), notes=notes, json=filt))

    T = 24
    z = np.zeros((T, 4), np.float32)
    # 1. plateau onset, saturated offsets (==1.0) with threshold 1.0, mpe dip
    on = z.copy(); off = z.copy(); mpe = z.copy() + 0.9; vel = np.full((T, 4), 64, np.int8)
    on[3:6, 0] = 0.8; on[10, 0] = 0.7; on[9, 0] = 0.3; on[11, 0] = 0.6
    off[7:9, 0] = 1.0; off[15, 0] = 1.0
    mpe[13:, 0] = 0.2
    on[0, 1] = 0.9; on[T - 1, 1] = 0.95; on[12, 1] = 0.55; on[11, 1] = 0.5; on[13, 1] = 0.52
    on[5, 2] = 0.6; on[6, 2] = 0.6; on[8, 2] = 0.9; vel[8, 2] = 0
    on[4, 3] = 0.51; on[5, 3] = 0.2; on[6, 3] = 0.99; mpe[5, 3] = 0.1
    run("crafted_plateau_edges", on, off, mpe, vel)
    # 2. overlapping same-pitch notes (offset clipped to next onset), threshold 0.5 offsets
    on = z.copy(); off = z.copy(); mpe = z.copy() + 0.9; vel = np.full((T, 4), 100, np.int8)
    on[2, 0] = 0.9; on[4, 0] = 0.8; on[3, 0] = 0.1
    off[20, 0] = 0.7; off[19, 0] = 0.6; off[21, 0] = 0.65
    run("crafted_overlap", on, off, mpe, vel, thr=(0.5, 0.5, 0.5), min_dur=0.0)
    # 3-5. random, with quantised values so ties/plateaus occur
    for k in range(3):
        Tn = 200
        on = np.round(rng.random((Tn, 4)) ** 4, 2).astype(np.float32)
        off = np.where(rng.random((Tn, 4)) > 0.97, 1.0, rng.random((Tn, 4)) * 0.99).astype(np.float32)
        mpe = rng.random((Tn, 4)).astype(np.float32)
        vel = rng.integers(0, 128, (Tn, 4)).astype(np.int8)
        run(f"random_{k}", on, off, mpe, vel, thr=(0.5, 1.0, 0.5) if k < 2 else (0.6, 0.9, 0.3))
    # 6. empty
    run("all_zero", z, z, z, np.zeros((T, 4), np.int8))
    (HERE / "mpe2note.json").write_text(json.dumps(cases))


def gen_mpe2note_modes():
    """the reference's _mpe2note with its non-default mode switches (extractor.py:256-258, 386-409) on random frame arrays"""
    ex, _ = ref_extractor(TINY_EXT, seed=11)
    ex.config.midi.num_note = 6
    rng = np.random.default_rng(41)
    cases, inputs = [], []
    for k in range(3):
        Tn = 160
        on = np.round(rng.random((Tn, 6)) ** 3, 2).astype(np.float32)
        off = np.round(rng.random((Tn, 6)) ** 2, 2).astype(np.float32)
        mpe = rng.random((Tn, 6)).astype(np.float32)
        vel = (rng.integers(0, 128, (Tn, 6)) * (rng.random((Tn, 6)) > 0.15)).astype(np.int8)
        inputs.append(dict(thr=[0.6, 0.5, 0.45], onset=on.tolist(), offset=off.tolist(), mpe=np.round(mpe, 4).tolist(), velocity=vel.tolist()))
        mpe = np.asarray(inputs[-1]["mpe"], np.float32)
        for mv in ("ignore_zero", "org"):
            for mo in ("shorter", "longer", "offset"):
                notes = ex._mpe2note(on, off, mpe, vel, thred_onset=0.6, thred_offset=0.5, thred_mpe=0.45, mode_velocity=mv, mode_offset=mo)
                cases.append(dict(input=k, mode_velocity=mv, mode_offset=mo, notes=notes))
    (HERE / "mpe2note_modes.json").write_text(json.dumps(dict(inputs=inputs, cases=cases)))
    print("  mpe2note_modes:", [len(c["notes"]) for c in cases])


def ref_decoder(dims, seed, **kw):
    from etude.models.etude_decoder import EtudeDecoder, EtudeDecoderConfig
    d = synth.decoder_dims(**dims)
    cfg = EtudeDecoderConfig(**{k: v for k, v in d.items()})
    model = EtudeDecoder(cfg)
    sd = synth.decoder_state_dict(seed, dims, **kw)
    model.load_state_dict(_t(sd), strict=True)
    model.eval()
    return model, d


class _Vocab:
    def __init__(self):
        from etude.data.vocab import Vocab
        p = HERE / "_vocab_tmp.json"
        synth.write_vocab(str(p))
        self.v = Vocab.load(p)
        p.unlink()


def _gen_decoder(name, dims, seed, kw, n_bars, attr_sets, prompt_len, max_bar_token_limit):
    model, d = ref_decoder(dims, seed, **kw)
    vocab = _Vocab().v
    rng = np.random.default_rng(99)
    ids = rng.integers(4, d["vocab_size"], (1, prompt_len))
    cls = rng.integers(1, 3, (1, prompt_len))
    at = {k: rng.integers(0, 3, (1, prompt_len)) for k in ("polyphony", "rhythm", "sustain", "overlap")}
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), class_ids=torch.from_numpy(cls),
                    polyphony_bin_ids=torch.from_numpy(at["polyphony"]), rhythm_intensity_bin_ids=torch.from_numpy(at["rhythm"]),
                    note_sustain_bin_ids=torch.from_numpy(at["sustain"]), pitch_overlap_bin_ids=torch.from_numpy(at["overlap"]),
                    use_cache=True)
    save = dict(prompt_ids=ids, prompt_cls=cls, prompt_polyphony=at["polyphony"], prompt_rhythm=at["rhythm"],
                prompt_sustain=at["sustain"], prompt_overlap=at["overlap"], logits=out.logits[0].numpy())
    bars = synth.song_bars(seed=3, n_bars=n_bars)
    save["n_bars"] = np.int64(n_bars)
    for j, a in enumerate(attr_sets):
        attrs = synth.attrs(*a)
        ev = model.generate(vocab, bars, [attrs] * n_bars, temperature=0.0, top_p=0.9, max_bar_token_limit=max_bar_token_limit)
        gen_ids = [vocab.encode(e) if e.type_ not in vocab.special_tokens else vocab.token_to_id[e.type_] for e in ev]
        save[f"gen_ids_{j}"] = np.asarray(gen_ids, np.int32)
        save[f"gen_attrs_{j}"] = np.asarray(a, np.int64)
        n_eos = int((np.asarray(gen_ids) == vocab.get_bar_eos_id()).sum())
        print(f"  {name} attrs={a}: {len(gen_ids)} ids, {n_eos} Bar_EOS, {len(set(gen_ids))} distinct")
    np.savez_compressed(HERE / f"{name}.npz", **save)


def gen_decoder_tiny():
    _gen_decoder("decoder_tiny", TINY_DEC, seed=2, kw=TINY_DEC_KW, n_bars=6, attr_sets=[(1, 1, 1, 2), (0, 2, 1, 2)],
                 prompt_len=24, max_bar_token_limit=40)


def gen_decoder_full():
    _gen_decoder("decoder_full", {}, seed=1, kw={}, n_bars=7, attr_sets=[(1, 1, 1, 2), (0, 2, 1, 2), (2, 0, 2, 2)],
                 prompt_len=64, max_bar_token_limit=48)


def ref_hft_wrapper(dims, seed, n_offset):
    """the reference's HFT_Transformer (etude/models/hft_transformer.py) around a seeded Model_SPEC2MIDI, without the pickle"""
    from etude.config.schema import HFTConfig
    from etude.models import amt_apc
    from etude.models.hft_transformer import HFT_Transformer
    d = synth.extractor_dims(**dims)
    enc = amt_apc.Encoder_SPEC2MIDI(d["n_margin"], d["n_frame"], d["n_bin"], d["cnn_channel"], d["cnn_kernel"], d["hid_dim"],
                                    d["n_layers_enc"], d["n_heads"], d["pf_dim"], 0.1, "cpu")
    dec = amt_apc.Decoder_SPEC2MIDI(d["n_frame"], d["n_bin"], d["n_note"], d["n_velocity"], d["hid_dim"], d["n_layers_dec"],
                                    d["n_heads"], d["pf_dim"], 0.1, "cpu")
    model = amt_apc.Model_SPEC2MIDI(enc, dec)
    sd = synth.extractor_state_dict(seed, dims)
    ren = {k.replace("encoder.", "encoder_spec2midi.", 1).replace("decoder.", "decoder_spec2midi.", 1) if k.split(".")[0] in ("encoder", "decoder") else k: v
           for k, v in sd.items()}
    model.load_state_dict(_t(ren), strict=True)
    model.eval()
    cfg = HFTConfig()
    cfg.input.margin_b = cfg.input.margin_f = d["n_margin"]
    cfg.input.num_frame = d["n_frame"]
    cfg.feature.n_bins = cfg.feature.mel_bins = d["n_bin"]
    cfg.midi.num_note = d["n_note"]
    cfg.midi.num_velocity = d["n_velocity"]
    cfg.infer.n_stride = n_offset
    tr = HFT_Transformer.__new__(HFT_Transformer)
    tr.device = "cpu"
    tr.config = cfg
    tr.model = model
    return tr, d, sd


def gen_hft_wrapper():
    """HFT_Transformer._transcript_stride / _transcript / _mpe2note (hft_transformer.py:140-674) at a tiny config and at the
    default architecture with num_frame=128; plus a plain-pickled tiny model object for the checkpoint loader."""
    import pickle
    # tiny: n_frame 16 -> half 8, n_offset 4; 27 frames -> 4 stride windows, ragged
    tr, d, sd = ref_hft_wrapper(TINY_EXT, seed=13, n_offset=4)
    rng = np.random.default_rng(31)
    feat = np.clip(rng.normal(-8, 2, (27, d["n_bin"])), -18, 5).astype(np.float32)
    so = tr._transcript_stride(torch.from_numpy(feat), 4)
    to = tr._transcript(torch.from_numpy(feat))
    np.savez_compressed(HERE / "hft_wrapper_tiny.npz", feature=feat, **{f"stride{i}": o for i, o in enumerate(so)},
                        **{f"plain{i}": o for i, o in enumerate(to)})
    with open(HERE / "hft_tiny_model.pkl", "wb") as f:
        pickle.dump(tr.model, f)                                   # how the hFT-Transformer project ships checkpoints (hft_transformer.py:53-54)
    np.savez_compressed(HERE / "hft_tiny_model_state.npz", **sd)
    # the wrapper's own _mpe2note (a second copy of the algorithm, :462-674) on its default thresholds
    ex, _ = ref_extractor(TINY_EXT, seed=11)
    on = np.round(rng.random((120, d["n_note"])) ** 3, 2).astype(np.float32)
    off = np.round(rng.random((120, d["n_note"])) ** 2, 2).astype(np.float32)
    mpe = rng.random((120, d["n_note"])).astype(np.float32)
    vel = (rng.integers(0, 100, (120, d["n_note"])) * (rng.random((120, d["n_note"])) > 0.1)).astype(np.int8)
    notes = tr._mpe2note(on, off, mpe, vel, thred_onset=0.75, thred_offset=0.5, thred_mpe=0.5)
    assert notes == ex._mpe2note(on, off, mpe, vel, thred_onset=0.75, thred_offset=0.5, thred_mpe=0.5), "the two reference copies differ"
    (HERE / "hft_wrapper_mpe2note.json").write_text(json.dumps(dict(thr=[0.75, 0.5, 0.5], onset=on.tolist(), offset=off.tolist(), mpe=mpe.tolist(),
                                                                    velocity=vel.tolist(), notes=notes)))
    # default architecture, num_frame 128, n_stride 32: 150 frames -> 3 stride windows
    tr, d, _ = ref_hft_wrapper(dict(n_frame=128), seed=7, n_offset=32)
    feat = np.clip(rng.normal(-8, 2, (150, 256)), -18, 5).astype(np.float32)
    so = tr._transcript_stride(torch.from_numpy(feat), 32)
    vl = None
    np.savez_compressed(HERE / "hft_wrapper_full.npz", feature=feat,
                        onset_A=so[0].astype(np.float16), mpe_A=so[2].astype(np.float16),
                        onset_B=so[4], offset_B=so[5], mpe_B=so[6], velocity_B=so[7])


def gen_tokenizer():
    """TinyREMITokenizer (etude/data/tokenizer.py): encode (with / without grace notes), split_sequence_into_bars and
    decode_to_notes (glissandos, velocity rules with and without a volume map) on synthetic songs."""
    import tempfile
    from etude.data.tokenizer import TinyREMITokenizer
    from etude.data.vocab import Event
    rng = np.random.default_rng(77)
    tmp = Path(tempfile.mkdtemp())

    def song(seed, tempo):
        r = np.random.default_rng(seed)
        t_end = max(rg["downbeats"][-1] for rg in tempo if rg["downbeats"]) + 3.0
        notes = []
        t = tempo[0]["downbeats"][0] - 1.0
        while t < t_end:
            k = int(r.integers(1, 5))
            base = int(r.integers(36, 100))
            for c in range(k):
                pitch = int(np.clip(base + int(r.integers(-7, 8)), 21, 108))
                dur = float(r.choice([0.05, 0.12, 0.25, 0.5, 0.9, 1.7, 3.1]))
                notes.append({"onset": round(float(t + (0.0 if r.random() < 0.6 else r.uniform(0, 0.03))), 6), "offset": round(float(t + dur), 6),
                              "pitch": pitch, "velocity": int(r.integers(1, 127))})
            if r.random() < 0.25:                       # a grace-note-like pair: neighbour pitch 20-90 ms earlier
                notes.append({"onset": round(float(t - r.uniform(0.02, 0.09)), 6), "offset": round(float(t), 6), "pitch": notes[-1]["pitch"] + int(r.choice([-1, 1])),
                              "velocity": 50})
            if r.random() < 0.1:
                notes.append(dict(notes[-1]))           # exact duplicate (same pitch, same onset)
            t += float(r.choice([0.125, 0.25, 0.25, 0.5, 1.0]))
        r.shuffle(notes)
        return notes

    def ev_list(events):
        return [[e.type_, e.value] for e in events]

    tempos = {
        "one_region_4_4": [{"start": 0.5, "bpm": 120, "time_sig": 4, "downbeats": [round(0.5 + 2.0 * i, 6) for i in range(12)]}],
        "two_regions_3_4": [{"start": 1.0, "bpm": 90.0, "time_sig": 4, "downbeats": [round(1.0 + (240 / 90) * i, 6) for i in range(5)]},
                            {"start": 14.5, "bpm": 140.0, "time_sig": 3, "downbeats": [round(14.5 + (180 / 140) * i, 6) for i in range(9)]}],
        "jittered": [{"start": 0.2, "bpm": 100.0, "time_sig": 4, "downbeats": np.round(np.cumsum(np.r_[0.2, rng.uniform(2.2, 2.6, 14)]), 5).tolist()},
                     {"start": 36.5, "bpm": 80.0, "time_sig": 4, "downbeats": []},          # skipped, but it ends the region before it
                     {"start": 37.0, "bpm": 80.0, "time_sig": 4, "downbeats": [37.0, 40.0, 43.0]}],
    }
    cases = []
    for name, tempo in tempos.items():
        tp = tmp / f"{name}_tempo.json"
        tp.write_text(json.dumps(tempo))
        for seed, grace in ((1, False), (2, True)):
            notes = song(seed + len(name), tempo)
            mp = tmp / "extract.json"
            mp.write_text(json.dumps(notes))
            tk = TinyREMITokenizer(str(tp))
            measures = [[m["start"], m["end"], m["bpm"], m["time_sig"]] for m in tk.global_measures]
            events = list(tk.encode(str(mp), with_grace_note=grace))
            # decode what was encoded (+ a few extra Grace events in a row to trigger the glissando path)
            dec_in = list(events)
            if grace:
                ins = []
                for e in dec_in:
                    if e.type_ == "Note" and rng.random() < 0.35:
                        ins.append(Event(type_="Grace", value=int(rng.choice([-1, 1]))))
                    ins.append(e)
                dec_in = ins
            vol = np.round(rng.random(int(20 * (measures[-1][1] + 1))) ** 2, 4).tolist()
            vp = tmp / "volume.json"
            vp.write_text(json.dumps(vol))
            keep = lambda ns: [{k: n[k] for k in ("pitch", "onset", "offset", "velocity")} for n in ns]      # noqa: E731
            d0 = keep(TinyREMITokenizer(str(tp)).decode_to_notes(list(dec_in)))
            d1 = keep(TinyREMITokenizer(str(tp)).decode_to_notes(list(dec_in), volume_map_path=str(vp)))
            cases.append(dict(name=f"{name}_{'grace' if grace else 'plain'}", tempo=tempo, notes=notes, with_grace=grace, measures=measures,
                              events=ev_list(events), decode_in=ev_list(dec_in), decoded=d0, volume=vol, decoded_vol=d1))
    # split_sequence_into_bars on raw id streams (Bar_BOS = 4, Bar_EOS = 5 as in the synthetic vocabulary)
    tk = TinyREMITokenizer(None)
    splits = []
    for k in range(6):
        ids = rng.integers(0, 12, int(rng.integers(0, 60))).tolist()
        splits.append(dict(ids=ids, bos=4, eos=5, bars=tk.split_sequence_into_bars(list(ids), 4, 5)))
    splits.append(dict(ids=[4, 7, 8, 4, 9, 5, 5, 6, 4, 10], bos=4, eos=5, bars=tk.split_sequence_into_bars([4, 7, 8, 4, 9, 5, 5, 6, 4, 10], 4, 5)))
    splits.append(dict(ids=[1, 2, 3], bos=-1, eos=5, bars=tk.split_sequence_into_bars([1, 2, 3], -1, 5)))
    (HERE / "tokenizer.json").write_text(json.dumps(dict(cases=cases, splits=splits)))


def gen_clip_full():
    """BASELINE configs[1]: ONE 3-min 44.1 kHz stereo clip (synth.clip_audio seed 1234) through the reference chain at the
    default configuration (n_frame 512 -> 22 windows, ragged tail):
        features (oracle/mel.py: the torchaudio front end is absent from this image -- parity-unpinned)
        -> reference AMTAPC_Extractor._transcript -> reference _mpe2note (0.5 / 1.0 / 0.5) -> reference _note2json (0.08 s)
        -> reference TinyREMITokenizer(synthetic tempo.json: 120 bpm, 4/4, 90 downbeats from 0.5 s).encode -> Vocab.encode_sequence
        -> split_sequence_into_bars -> reference EtudeDecoder.generate (greedy, attrs 1/1/1 + overlap 2, default limits).
    Extractor weights = synth.extractor_state_dict(0), decoder weights = synth.decoder_state_dict(1): what bench.py runs.
    Takes ~6 min of CPU (22 reference forward passes).  Stored: every 8th frame of the B outputs (fp16 is 5e-4 abs, far below the
    test tolerance), the full velocity argmax + top-2 gap of every 8th frame, the full note list, the bars and the generated ids."""
    import tempfile
    from oracle import mel
    from etude.data.tokenizer import TinyREMITokenizer
    wav = synth.clip_audio(seed=1234, seconds=180.0)
    feat = mel.wav2feature(torch.from_numpy(wav), 44100).numpy()
    ex, d = ref_extractor({}, seed=0)
    torch.set_num_threads(8)
    out = ex._transcript(feat)
    on, off, mpe, vel = out[4], out[5], out[6], out[7]
    notes = ex._mpe2note(on, off, mpe, vel, thred_onset=0.5, thred_offset=1.0, thred_mpe=0.5)
    tmp = Path(tempfile.mkdtemp())
    ex._note2json(notes, str(tmp / "extract.json"), 0.08)
    kept = json.loads((tmp / "extract.json").read_text())
    tempo = [{"start": 0.5, "bpm": 120, "time_sig": 4, "downbeats": [round(0.5 + 2.0 * i, 6) for i in range(90)]}]
    (tmp / "tempo.json").write_text(json.dumps(tempo))
    vocab = _Vocab().v
    tk = TinyREMITokenizer(tempo_path=str(tmp / "tempo.json"))
    ev = tk.encode(str(tmp / "extract.json"))
    ids = vocab.encode_sequence(ev)
    bars = tk.split_sequence_into_bars(ids, vocab.get_bar_bos_id(), vocab.get_bar_eos_id())
    model, _ = ref_decoder({}, 1)
    attrs = synth.attrs(1, 1, 1, 2)
    gev = model.generate(vocab, bars, [attrs] * len(bars), temperature=0.0, top_p=0.9)
    gen_ids = [vocab.encode(e) if e.type_ not in vocab.special_tokens else vocab.token_to_id[e.type_] for e in gev]
    print(f"  clip_full: {feat.shape[0]} frames, {len(notes)} notes ({len(kept)} after the 0.08 s filter), {len(bars)} bars, "
          f"{sum(len(b) for b in bars)} condition ids (UNK {sum(1 for i in ids if i == 1)}), {len(gen_ids)} generated ids, "
          f"{int((np.asarray(gen_ids) == vocab.get_bar_eos_id()).sum())} Bar_EOS")
    sub = slice(None, None, 8)
    arr = lambda key: np.asarray([n[key] for n in notes])       # noqa: E731
    np.savez_compressed(HERE / "clip_full.npz", n_frames=np.int64(feat.shape[0]), feat_rows=feat[::512],
                        onset_B=on[sub].astype(np.float16), offset_B=off[sub].astype(np.float16), mpe_B=mpe[sub].astype(np.float16),
                        offset_B_sat=np.packbits(off >= 1.0), velocity_B=vel,
                        note_onset=arr("onset").astype(np.float64), note_offset=arr("offset").astype(np.float64),
                        note_pitch=arr("pitch").astype(np.int32), note_velocity=arr("velocity").astype(np.int32),
                        n_kept=np.int64(len(kept)),
                        bar_ids=np.asarray([t for b in bars for t in b], np.int32), bar_lens=np.asarray([len(b) for b in bars], np.int32),
                        gen_ids=np.asarray(gen_ids, np.int32))


def _clip_full_bars():
    """the condition bars of configs[1] as the reference chain produced them (data of clip_full.npz)"""
    g = np.load(HERE / "clip_full.npz")
    flat, lens = g["bar_ids"].tolist(), g["bar_lens"].tolist()
    bars, p = [], 0
    for l in lens:
        bars.append(flat[p:p + l]); p += l
    return bars


def _ref_logits(model, ids, cls, at):
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), class_ids=torch.from_numpy(cls),
                    polyphony_bin_ids=torch.from_numpy(at["polyphony"]), rhythm_intensity_bin_ids=torch.from_numpy(at["rhythm"]),
                    note_sustain_bin_ids=torch.from_numpy(at["sustain"]), pitch_overlap_bin_ids=torch.from_numpy(at["overlap"]), use_cache=False)
    return out.logits[0].numpy()


def gen_decoder_ctx():
    """Decoder goldens whose greedy path DEPENDS ON THE CONTEXT (synth.decoder_state_dict_ctx) and logits at LONG contexts:
      * reference logits of a 1 024-token and a 3 500-token prompt at sampled positions, for the context weights and for the
        benchmark's weights (decoder_state_dict(1)) -- pins the oracle's RoPE / attention far beyond the 64 tokens of decoder_full;
      * greedy ids of the reference's generate() on the first 20 condition bars of configs[1] (clip_full.npz's bars, ~84 ids each:
        prompts reach the 512-token truncation from bar 4 on) for two attribute tuples, max_bar_token_limit 128."""
    vocab = _Vocab().v
    bars = _clip_full_bars()[:20]
    save = {}
    rng = np.random.default_rng(2024)
    for T in (1024, 3500):
        ids = rng.integers(4, 154, (1, T)); cls = rng.integers(1, 3, (1, T))
        at = {k: rng.integers(0, 3, (1, T)) for k in ("polyphony", "rhythm", "sustain", "overlap")}
        pos = np.asarray([0, 1, T // 2 - 1, T // 2, T - 2, T - 1] + sorted(rng.integers(2, T - 2, 10).tolist()), np.int64)
        save.update({f"p{T}_ids": ids, f"p{T}_cls": cls, f"p{T}_pos": pos, **{f"p{T}_{k}": v for k, v in at.items()}})
        for tag, (seed, kw, ctxw) in (("ctx", (1, {}, True)), ("bench", (1, {}, False))):
            if ctxw:
                from etude.models.etude_decoder import EtudeDecoder, EtudeDecoderConfig
                model = EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()))
                model.load_state_dict(_t(synth.decoder_state_dict_ctx(seed)), strict=True)
                model.eval()
            else:
                model, _ = ref_decoder({}, seed, **kw)
            lg = _ref_logits(model, ids, cls, at)
            save[f"p{T}_logits_{tag}"] = lg[pos]
            print(f"  decoder_ctx: T={T} {tag}: logits range {lg.min():.2f} .. {lg.max():.2f}")
    from etude.models.etude_decoder import EtudeDecoder, EtudeDecoderConfig
    model = EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()))
    model.load_state_dict(_t(synth.decoder_state_dict_ctx(1)), strict=True)
    model.eval()
    save["n_bars"] = np.int64(len(bars))
    for j, a in enumerate([(1, 1, 1, 2), (0, 2, 1, 2)]):
        ev = model.generate(vocab, bars, [synth.attrs(*a)] * len(bars), temperature=0.0, top_p=0.9, max_bar_token_limit=128)
        gen_ids = [vocab.encode(e) if e.type_ not in vocab.special_tokens else vocab.token_to_id[e.type_] for e in ev]
        save[f"gen_ids_{j}"] = np.asarray(gen_ids, np.int32); save[f"gen_attrs_{j}"] = np.asarray(a, np.int64)
        arr = np.asarray(gen_ids)
        from collections import Counter, defaultdict
        m = defaultdict(Counter)
        for x, y in zip(arr[:-1], arr[1:]):
            m[int(x)][int(y)] += 1
        pred = sum(c.most_common(1)[0][1] for c in m.values()) / max(1, len(arr) - 1)
        print(f"  decoder_ctx attrs={a}: {len(gen_ids)} ids, {int((arr == vocab.get_bar_eos_id()).sum())} Bar_EOS in {len(bars)} bars, {len(set(gen_ids))} distinct, "
              f"{100 * pred:.1f} % predictable from the previous id")
    np.savez_compressed(HERE / "decoder_ctx.npz", **save)


def gen_clip_ctx():
    """configs[1]'s decode stage with the context weights: the reference's generate() over ALL 92 condition bars of clip_full.npz,
    default limits (max_bar_token_limit 512, max_output_tokens 25 600), attributes 1/1/1 + overlap 2."""
    from etude.models.etude_decoder import EtudeDecoder, EtudeDecoderConfig
    vocab = _Vocab().v
    bars = _clip_full_bars()
    model = EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()))
    model.load_state_dict(_t(synth.decoder_state_dict_ctx(1)), strict=True)
    model.eval()
    torch.set_num_threads(8)
    ev = model.generate(vocab, bars, [synth.attrs(1, 1, 1, 2)] * len(bars), temperature=0.0, top_p=0.9)
    gen_ids = [vocab.encode(e) if e.type_ not in vocab.special_tokens else vocab.token_to_id[e.type_] for e in ev]
    arr = np.asarray(gen_ids)
    print(f"  clip_ctx: {len(gen_ids)} ids, {int((arr == vocab.get_bar_bos_id()).sum())} bars, {int((arr == vocab.get_bar_eos_id()).sum())} Bar_EOS, {len(set(gen_ids))} distinct")
    np.savez_compressed(HERE / "clip_ctx.npz", gen_ids=np.asarray(gen_ids, np.int32))


ALL = dict(hft_full_cal=gen_hft_full_cal, decoder_ctx=gen_decoder_ctx, clip_ctx=gen_clip_ctx, clip_full=gen_clip_full, mpe2note_modes=gen_mpe2note_modes, tokenizer=gen_tokenizer, hft_wrapper=gen_hft_wrapper, hft_tiny=gen_hft_tiny, hft_full=gen_hft_full, transcript_tiny=gen_transcript_tiny, mpe2note=gen_mpe2note,
           decoder_tiny=gen_decoder_tiny, decoder_full=gen_decoder_full)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    for n, f in ALL.items():
        if a.only and a.only != n:
            continue
        print("generating", n)
        f()

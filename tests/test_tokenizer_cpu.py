"""TinyREMITokenizer glue through the C ABI against vectors captured from the reference class
(etude/data/tokenizer.py): measures, encode with / without grace-note linking, split_sequence_into_bars, decode_to_notes with
glissandos and both velocity rules.  Everything is compared exactly (doubles bit for bit)."""
import json

import pytest

from etude_amd.vocab import Event


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.loads((golden_dir / "tokenizer.json").read_text())


def _tok(tmp_path, tempo):
    from etude_amd.tokenizer import TinyREMITokenizer
    p = tmp_path / "tempo.json"
    p.write_text(json.dumps(tempo))
    return TinyREMITokenizer(str(p))


def _ev(pairs):
    return [Event(type_=t, value=v) for t, v in pairs]


def test_measures_and_encode(gold, tmp_path):
    for c in gold["cases"]:
        tk = _tok(tmp_path, c["tempo"])
        got_m = [[m["start"], m["end"], m["bpm"], m["time_sig"]] for m in tk.global_measures]
        assert got_m == c["measures"], c["name"]
        mp = tmp_path / "extract.json"
        mp.write_text(json.dumps(c["notes"]))
        events = tk.encode(str(mp), with_grace_note=c["with_grace"])
        assert [[e.type_, e.value] for e in events] == c["events"], c["name"]
        assert any(e.type_ == "Grace" for e in events) == any(t == "Grace" for t, _ in c["events"])


def test_decode_to_notes_with_and_without_volume_map(gold, tmp_path):
    saw_gliss = False
    for c in gold["cases"]:
        tk = _tok(tmp_path, c["tempo"])
        got = tk.decode_to_notes(_ev(c["decode_in"]))
        assert got == c["decoded"], c["name"]
        vp = tmp_path / "volume.json"
        vp.write_text(json.dumps(c["volume"]))
        assert tk.decode_to_notes(_ev(c["decode_in"]), volume_map_path=str(vp)) == c["decoded_vol"], c["name"]
        assert tk.decode_to_notes(_ev(c["decode_in"]), volume_map_path=str(tmp_path / "missing.json")) == c["decoded"]    # unreadable map -> count rule
        saw_gliss = saw_gliss or any(abs((n["offset"] - n["onset"]) - 0.1) < 1e-12 and n["velocity"] != 65 for n in c["decoded"])
    assert saw_gliss, "no case exercised the glissando rewrite"


def test_split_sequence_into_bars(gold):
    from etude_amd.tokenizer import TinyREMITokenizer
    tk = TinyREMITokenizer(None)
    assert tk.global_measures == [] and tk.tempo_data == []
    for s in gold["splits"]:
        assert tk.split_sequence_into_bars(s["ids"], s["bos"], s["eos"]) == s["bars"]


def test_round_trip_through_the_decoder_boundary(gold, tmp_path):
    """encode -> vocabulary ids -> bars -> events -> notes: the path infer.py:180-206 walks around the decoder"""
    from etude_amd.vocab import Vocab
    c = gold["cases"][0]
    tk = _tok(tmp_path, c["tempo"])
    events = tk.encode_notes(c["notes"])
    v = Vocab()
    for e in events:
        v._add_token(str(e))
    ids = v.encode_sequence(events)
    bars = tk.split_sequence_into_bars(ids, v.get_bar_bos_id(), v.get_bar_eos_id())
    assert len(bars) == len(tk.global_measures) and sum(len(b) for b in bars) == len(ids)
    back = [e for b in bars for e in v.decode_sequence_to_events(b)]
    assert tk.decode_to_notes(back) == c["decoded"]


def test_array_fast_path_equals_object_path(gold, tmp_path):
    """ids -> notes without per-token Python objects (what a batched run over (clip, attribute tuple) jobs uses)"""
    import numpy as np

    from etude_amd.vocab import Vocab
    c = gold["cases"][3]
    tk = _tok(tmp_path, c["tempo"])
    events = _ev(c["decode_in"])
    v = Vocab()
    for e in events:
        v._add_token(str(e))
    ids = v.encode_sequence(events)
    tab = tk.event_table(v)
    arr = tk.decode_ids_to_note_array(ids + [v.get_pad_id()] * 3, tab, volume=np.asarray(c["volume"]), pad_id=v.get_pad_id())
    got = [{"pitch": int(p), "onset": float(a), "offset": float(b), "velocity": int(w)} for p, a, b, w in zip(arr["pitch"], arr["onset"], arr["offset"], arr["velocity"])]
    assert got == c["decoded_vol"]
    ev = tk.encode_note_array_to_events(np.asarray([(n["onset"], n["offset"], n["pitch"], n["velocity"]) for n in c["notes"]], dtype=arr.dtype), with_grace_note=True)
    names = ("Bar", "Pos", "Note", "Duration", "Grace")
    assert [[names[t], ("BOS" if x == 1 else "EOS") if t == 0 else x] for t, x in zip(ev["type"].tolist(), ev["value"].tolist())] == c["events"]


def test_empty_first_or_last_tempo_region_raises_like_the_reference(tmp_path):
    """tokenizer.py:166-229: `_create_measures` indexes the first / last region's downbeats -> IndexError when one is empty
    (an empty region in the middle is skipped).  Found by a differential run against the reference class over random tempo maps."""
    import json
    from etude_amd.tokenizer import TinyREMITokenizer
    good = {"start": 0.0, "bpm": 120, "time_sig": 4, "downbeats": [0.0, 2.0, 4.0]}
    empty = {"start": 5.0, "bpm": 100, "time_sig": 4, "downbeats": []}
    for regions in ([good, empty], [empty, dict(good, start=6.0, downbeats=[6.0, 8.0])]):
        p = tmp_path / "t.json"
        p.write_text(json.dumps(regions))
        with pytest.raises(IndexError):
            TinyREMITokenizer(str(p))
    p.write_text(json.dumps([good, empty, dict(good, start=9.0, downbeats=[9.0, 11.0])]))
    assert len(TinyREMITokenizer(str(p)).global_measures) > 0


def test_full_clip_notes_to_bars_match_reference(golden_dir, tmp_path):
    """configs[1] chain, host part: the reference's note list of the 3-min clip (tests/golden/clip_full.npz) -> native tokenizer
    -> exactly the condition bars the reference's TinyREMITokenizer + Vocab produced."""
    import json
    import numpy as np
    from etude_amd import synth
    from etude_amd.tokenizer import TinyREMITokenizer
    from etude_amd.vocab import Vocab
    p = golden_dir / "clip_full.npz"
    if not p.exists():
        import pytest
        pytest.skip("clip_full.npz not generated")
    g = np.load(p)
    notes = [dict(onset=float(a), offset=float(b), pitch=int(q), velocity=int(v))
             for a, b, q, v in zip(g["note_onset"], g["note_offset"], g["note_pitch"], g["note_velocity"])]
    kept = [n for n in notes if not (n["offset"] - n["onset"] < 0.08)]
    assert len(kept) == int(g["n_kept"])
    tempo = [{"start": 0.5, "bpm": 120, "time_sig": 4, "downbeats": [round(0.5 + 2.0 * i, 6) for i in range(90)]}]
    (tmp_path / "tempo.json").write_text(json.dumps(tempo))
    (tmp_path / "extract.json").write_text(json.dumps(kept))
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    tk = TinyREMITokenizer(str(tmp_path / "tempo.json"))
    ids = v.encode_sequence(tk.encode(str(tmp_path / "extract.json")))
    bars = tk.split_sequence_into_bars(ids, v.get_bar_bos_id(), v.get_bar_eos_id())
    assert [len(b) for b in bars] == g["bar_lens"].tolist()
    assert [t for b in bars for t in b] == g["bar_ids"].tolist()


def test_array_paths_of_the_batch_pipeline_match_the_object_paths():
    """etude_amd/pipeline.py hands notes -> events -> ids -> bars from stage to stage as arrays (no per-note Python objects): every array path must give exactly what the
    reference-shaped object path gives -- `id_lookup` / `events_to_ids` vs `Vocab.encode_sequence`, `split_ids_into_packed_bars` vs `split_sequence_into_bars`, `PackedBars`."""
    import numpy as np
    from etude_amd import synth
    from etude_amd.decoder import PackedBars
    from etude_amd.extractor import NOTE_DTYPE
    from etude_amd.pipeline import attr_grid, synthetic_tempo
    from etude_amd.tokenizer import TinyREMITokenizer
    from etude_amd.vocab import Vocab
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    lut = TinyREMITokenizer.id_lookup(v)
    table = TinyREMITokenizer.event_table(v)
    # every vocabulary token that is an event maps back to its own id
    for i in range(len(v)):
        t, val = int(table[i]["type"]), int(table[i]["value"])
        if t < 5:
            assert int(lut[t, val + 2048]) == i, v.id_to_token[i]
    rng = np.random.default_rng(3)
    for trial in range(4):
        n = 300 + 50 * trial
        notes = np.zeros(n, NOTE_DTYPE)
        notes["onset"] = np.sort(rng.uniform(0, 40, n)); notes["offset"] = notes["onset"] + rng.uniform(0.05, 2.0, n)
        notes["pitch"] = rng.integers(10, 120, n); notes["velocity"] = rng.integers(1, 127, n)      # pitches outside 21..108 become <UNK>
        tempo = synthetic_tempo(n_downbeats=20)
        tk_a, tk_o = TinyREMITokenizer.from_tempo_data(tempo), TinyREMITokenizer.from_tempo_data(tempo)
        ids_a = tk_a.events_to_ids(tk_a.encode_note_array_to_events(notes), lut)
        ids_o = v.encode_sequence(tk_o.encode_notes(notes))
        assert ids_a.tolist() == ids_o
        bi, bo = tk_a.split_ids_into_packed_bars(ids_a, v.get_bar_bos_id(), v.get_bar_eos_id())
        bars = tk_o.split_sequence_into_bars(ids_o, v.get_bar_bos_id(), v.get_bar_eos_id())
        pb = PackedBars(bi, bo)
        assert len(pb) == len(bars) and [pb.bar(i) for i in range(len(pb))] == bars
        pb2 = PackedBars.from_lists(bars)
        assert np.array_equal(pb2.ids, pb.ids) and np.array_equal(pb2.offsets, pb.offsets)
    g = attr_grid(27)
    assert len({(a["polyphony_bin"], a["rhythm_intensity_bin"], a["sustain_bin"]) for a in g}) == 27 and all(a["pitch_overlap_bin"] == 2 for a in g)
    assert attr_grid(1) == [dict(polyphony_bin=1, rhythm_intensity_bin=1, sustain_bin=1, pitch_overlap_bin=2)]
    import pytest
    with pytest.raises(ValueError):
        PackedBars(np.zeros(4, np.int32), np.asarray([0, 5], np.int32))

"""note_to_midi (etude/data/tokenizer.py:499-524) through the native writer, etd_midi_write.

pretty_midi is not in the image, so there is no golden file from the reference's dependency ("parity unpinned" for the
byte stream).  What is checked: the file parses as a Standard MIDI File with the layout pretty_midi's write() produces for
`PrettyMIDI()` + `Instrument(program=0)` (format 1, 220 ticks per beat, 4/4, 120 bpm), every note comes back at
round(time * 440) ticks (Python round, half to even), simultaneous events follow pretty_midi's secondary ordering, and one
small case matches a byte string worked out by hand from that algorithm.
"""
import numpy as np
import pytest

from etude_amd.extractor import NOTE_DTYPE
from etude_amd.tokenizer import TinyREMITokenizer


def _varint(b, i):
    v = 0
    while True:
        c = b[i]; i += 1
        v = (v << 7) | (c & 0x7F)
        if not c & 0x80:
            return v, i


def parse_smf(data: bytes):
    assert data[:4] == b"MThd" and int.from_bytes(data[4:8], "big") == 6
    fmt, ntrk, div = (int.from_bytes(data[8 + 2 * k: 10 + 2 * k], "big") for k in range(3))
    i, tracks = 14, []
    for _ in range(ntrk):
        assert data[i:i + 4] == b"MTrk"
        n = int.from_bytes(data[i + 4:i + 8], "big"); i += 8
        end, tick, status, evs = i + n, 0, None, []
        while i < end:
            d, i = _varint(data, i); tick += d
            if data[i] == 0xFF:
                kind = data[i + 1]; ln, j = _varint(data, i + 2)
                evs.append((tick, "meta", kind, bytes(data[j:j + ln]))); i = j + ln; status = None
            else:
                if data[i] & 0x80:
                    status = data[i]; i += 1
                nargs = 1 if (status & 0xF0) in (0xC0, 0xD0) else 2
                evs.append((tick, "msg", status, tuple(data[i:i + nargs]))); i += nargs
        assert i == end
        tracks.append(evs)
    assert i == len(data)
    return fmt, div, tracks


def test_midi_layout_and_ticks(tmp_path):
    rng = np.random.default_rng(5)
    n = 300
    onset = np.sort(rng.uniform(0, 120, n)); onset[:4] = [0.0, 0.0, 1.0 / 880, 3.0 / 880]          # zero time and exact .5-tick ties (round half to even)
    notes = [{"pitch": int(p), "onset": float(a), "offset": float(a + d), "velocity": int(v)}
             for p, a, d, v in zip(rng.integers(21, 109, n), onset, rng.uniform(0.01, 3, n), rng.integers(1, 128, n))]
    out = tmp_path / "sub" / "x.mid"
    TinyREMITokenizer.note_to_midi(notes, out)
    fmt, div, (t0, t1) = parse_smf(out.read_bytes())
    assert (fmt, div) == (1, 220)
    # pretty_midi sorts the timing track with event_compare: set_tempo (rank 1) before time_signature (rank 2) at tick 0
    assert t0 == [(0, "meta", 0x51, (500000).to_bytes(3, "big")), (0, "meta", 0x58, bytes([4, 2, 24, 8])), (1, "meta", 0x2F, b"")]
    assert t1[0] == (0, "msg", 0xC0, (0,))
    assert t1[-1][1:] == ("meta", 0x2F, b"") and t1[-1][0] == t1[-2][0] + 1
    body = t1[1:-1]
    tick = lambda t: int(round(t / (60.0 / (120.0 * 220)))) if t > 0 else 0
    want = []
    for nd in notes:
        want.append((tick(nd["onset"]), nd["pitch"] * 256 + nd["velocity"], nd["pitch"], nd["velocity"]))
        want.append((tick(nd["offset"]), nd["pitch"] * 256, nd["pitch"], 0))
    want.sort(key=lambda e: (e[0], e[1]))                           # Python's sort is stable, like sorted(..., key=cmp_to_key(event_compare))
    assert [(t, a[0], a[1]) for t, _, st, a in body] == [(t, p, v) for t, _, p, v in want]
    assert all(st == 0x90 for _, _, st, _ in body)
    assert tick(1.0 / 880) == 0 and tick(3.0 / 880) == 2            # the half-tick cases really are ties


def test_midi_small_case_bytes_and_array_input(tmp_path):
    arr = np.array([(0.5, 1.0, 60, 100), (0.5, 0.75, 64, 80)], dtype=NOTE_DTYPE)
    out = tmp_path / "y.mid"
    TinyREMITokenizer.note_to_midi(arr, out)
    trk1 = bytes([0x00, 0xC0, 0x00,                     # program change
                  0x81, 0x5C, 0x90, 60, 100,            # tick 220: note_on 60 (varint 220 = 81 5C)
                  0x00, 64, 80,                         # same tick, running status: note_on 64
                  0x6E, 64, 0,                          # +110 -> tick 330: 64 off
                  0x6E, 60, 0,                          # +110 -> tick 440: 60 off
                  0x01, 0xFF, 0x2F, 0x00])
    trk0 = bytes([0, 0xFF, 0x51, 3, 0x07, 0xA1, 0x20, 0, 0xFF, 0x58, 4, 4, 2, 24, 8, 1, 0xFF, 0x2F, 0])
    want = b"MThd" + (6).to_bytes(4, "big") + bytes([0, 1, 0, 2, 0, 220]) + b"MTrk" + len(trk0).to_bytes(4, "big") + trk0 + b"MTrk" + len(trk1).to_bytes(4, "big") + trk1
    assert out.read_bytes() == want
    TinyREMITokenizer.note_to_midi([], tmp_path / "empty.mid")
    _, _, (_, t1) = parse_smf((tmp_path / "empty.mid").read_bytes())
    assert t1 == [(0, "msg", 0xC0, (0,)), (1, "meta", 0x2F, b"")]


def test_midi_rejects_out_of_range(tmp_path):
    with pytest.raises(RuntimeError):
        TinyREMITokenizer.note_to_midi([{"pitch": 128, "onset": 0.0, "offset": 1.0, "velocity": 64}], tmp_path / "bad.mid")
    with pytest.raises(RuntimeError):
        TinyREMITokenizer.note_to_midi([{"pitch": 60, "onset": 0.0, "offset": 1.0, "velocity": 200}], tmp_path / "bad.mid")


def test_extractor_note2midi_filters_short_notes(tmp_path):
    """AMTAPC_Extractor._note2midi (extractor.py:421-429): notes shorter than min_length are dropped, the rest written."""
    from etude_amd.extractor import AMTAPC_Extractor
    notes = [{"onset": 0.1, "offset": 0.5, "pitch": 60, "velocity": 90}, {"onset": 0.2, "offset": 0.21, "pitch": 61, "velocity": 90}]
    AMTAPC_Extractor._note2midi(None, notes, str(tmp_path / "a.mid"), 0.05)
    _, _, (_, t1) = parse_smf((tmp_path / "a.mid").read_bytes())
    assert [e for e in t1 if e[1] == "msg" and e[2] == 0x90] == [(44, "msg", 0x90, (60, 90)), (220, "msg", 0x90, (60, 0))]

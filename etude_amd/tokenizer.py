"""TinyREMITokenizer -- drop-in for etude/data/tokenizer.py:23-524.

The glue on either side of the decoder in infer.py:180-206: ``encode`` (extract.json + tempo.json -> REMI events),
``split_sequence_into_bars`` (id sequence -> condition bars) and ``decode_to_notes`` (generated events -> notes with
velocities).  All of it runs in the library's native code (csrc/tokenizer.cpp, ``etd_tok_*``), bit-identical to the
reference; this class only converts between Python objects and the C structs.  Host code: no GPU involved.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .extractor import NOTE_DTYPE
from .vocab import Event

PAD_CLASS_ID, SRC_CLASS_ID, TGT_CLASS_ID = 0, 1, 2          # tokenizer.py:15-17
_EV_TYPES = ("Bar", "Pos", "Note", "Duration", "Grace")
_EVENT_DTYPE = np.dtype([("type", "<i4"), ("value", "<i4")])


def _events_to_array(events: Sequence[Event]) -> np.ndarray:
    arr = np.empty(len(events), dtype=_EVENT_DTYPE)
    for i, e in enumerate(events):
        t = e.type_
        if t == "Bar":
            arr[i] = (0, 1) if e.value == "BOS" else ((0, 0) if e.value == "EOS" else (5, 0))
        elif t in ("Pos", "Note", "Duration", "Grace") and isinstance(e.value, (int, np.integer)):
            arr[i] = (_EV_TYPES.index(t), int(e.value))
        else:
            arr[i] = (5, 0)                                   # specials / unknown types: skipped by decode_to_notes, like the reference
    return arr


class TinyREMITokenizer:
    """Signature of etude/data/tokenizer.py:24."""

    def __init__(self, tempo_path: Optional[Union[str, Path]]):
        self.all_events: List[Event] = []
        if tempo_path and Path(tempo_path).exists():
            with open(tempo_path, "r") as f:
                tempo_data = json.load(f)
        else:
            tempo_data = []
        self.time_resolution_for_map = 20
        self._h = None
        self.set_tempo(tempo_data)

    @classmethod
    def from_tempo_data(cls, tempo_data: Sequence[dict]) -> "TinyREMITokenizer":
        """The tokenizer of an in-memory tempo.json (list of regions) -- batch pipelines hand it over without a file."""
        tk = cls(None)
        tk.set_tempo(tempo_data)
        return tk

    def set_tempo(self, tempo_data: Sequence[dict]) -> None:
        """(Re)build the measure grid from tempo.json content (tokenizer.py:24-41, 166-229)."""
        lib = _lib.lib()
        if getattr(self, "_h", None):
            lib.etd_tok_destroy(self._h)
            self._h = None
        self.tempo_data = list(tempo_data) if tempo_data else []
        self._keep = []
        regs = (_lib.TempoRegion * max(1, len(self.tempo_data)))()
        for i, r in enumerate(self.tempo_data):
            db = np.ascontiguousarray(r.get("downbeats", []), np.float64)
            self._keep.append(db)
            regs[i] = _lib.TempoRegion(float(r["bpm"]) if db.size else float(r.get("bpm", 120.0)), int(r["time_sig"]) if db.size else int(r.get("time_sig", 4)),
                                       float(r.get("start", 0.0)), db.ctypes.data if db.size else None, int(db.size))
        h = C.c_void_p()
        if self.tempo_data and (not self.tempo_data[0].get("downbeats") or not self.tempo_data[-1].get("downbeats")):
            raise IndexError("list index out of range")           # what the reference's _create_measures does with an empty first / last region (tokenizer.py:166-229)
        _lib.check(lib.etd_tok_create(C.cast(regs, C.c_void_p), len(self.tempo_data), C.byref(h)), "etd_tok_create")
        self._h = h
        n = lib.etd_tok_num_measures(h)
        st, en, bp = np.zeros(n), np.zeros(n), np.zeros(n)
        ts = np.zeros(n, np.int32)
        if n:
            _lib.check(lib.etd_tok_measures(h, st.ctypes.data, en.ctypes.data, bp.ctypes.data, ts.ctypes.data), "etd_tok_measures")
        self.global_measures = [{"bpm": float(b), "start": float(s), "end": float(e), "time_sig": int(t)} for s, e, b, t in zip(st, en, bp, ts)]

    # ------------------------------------------------------------------ reference surface
    def split_sequence_into_bars(self, id_sequence: list, bar_bos_id: int, bar_eos_id: int) -> List[List[int]]:
        """tokenizer.py:43-76."""
        ids = np.ascontiguousarray(id_sequence, np.int32)
        n = int(ids.size)
        out = np.empty(n + 1, np.int32)
        offs = np.empty(n + 2, np.int64)
        nb = C.c_longlong()
        _lib.check(_lib.lib().etd_tok_split_bars(ids.ctypes.data, n, int(bar_bos_id), int(bar_eos_id), out.ctypes.data, n + 1, offs.ctypes.data, n + 2,
                                                 C.byref(nb)), "etd_tok_split_bars")
        return [out[offs[b]: offs[b + 1]].tolist() for b in range(nb.value)]

    def encode_notes(self, midi_data: Sequence[dict], with_grace_note: bool = False) -> List[Event]:
        """`encode` on an in-memory note list ({onset, offset, pitch, ...} dicts or a NOTE_DTYPE array)."""
        if isinstance(midi_data, np.ndarray) and midi_data.dtype == NOTE_DTYPE:
            notes = np.ascontiguousarray(midi_data)
        else:
            notes = np.empty(len(midi_data), dtype=NOTE_DTYPE)
            for i, nt in enumerate(midi_data):
                notes[i] = (float(nt["onset"]), float(nt["offset"]), int(nt["pitch"]), int(nt.get("velocity", 0)))
        n = int(notes.size)
        cap = 3 * n + 2 * len(self.global_measures) + 2 * n + 8
        ev = np.empty(cap, dtype=_EVENT_DTYPE)
        k = C.c_longlong()
        _lib.check(_lib.lib().etd_tok_encode(self._h, notes.ctypes.data, n, 1 if with_grace_note else 0, ev.ctypes.data, cap, C.byref(k)), "etd_tok_encode")
        out = []
        for t, v in zip(ev["type"][: k.value].tolist(), ev["value"][: k.value].tolist()):
            out.append(Event(type_="Bar", value="BOS" if v == 1 else "EOS") if t == 0 else Event(type_=_EV_TYPES[t], value=v))
        self.all_events.extend(out)                           # the reference accumulates across calls (tokenizer.py:31,297)
        return self.all_events

    def encode(self, midi_path: str, with_grace_note: bool = False) -> List[Event]:
        """tokenizer.py:265-297."""
        with open(midi_path, "r") as f:
            midi_data = json.load(f)
        return self.encode_notes(midi_data, with_grace_note)

    def decode_to_notes(self, events: List[Event], volume_map_path: Optional[str] = None) -> List[dict]:
        """tokenizer.py:446-496.  Returns dicts with pitch / onset / offset / velocity (what note_to_midi consumes)."""
        vol = None
        if volume_map_path:
            try:
                with open(volume_map_path, "r") as f:
                    vol = np.ascontiguousarray(np.array(json.load(f)), np.float64)
            except Exception:                                  # the reference logs a warning and falls back to the note-count rule
                vol = None
        ev = _events_to_array(events)
        n = int(ev.size)
        cap = 2 * n + 256
        while True:
            out = np.empty(cap, dtype=NOTE_DTYPE)
            k = C.c_longlong()
            rc = _lib.lib().etd_tok_decode(self._h, ev.ctypes.data, n, vol.ctypes.data if vol is not None else None, int(vol.size) if vol is not None else 0,
                                           out.ctypes.data, cap, C.byref(k))
            if rc == -12 and k.value > cap:
                cap = int(k.value)
                continue
            _lib.check(rc, "etd_tok_decode")
            break
        out = out[: k.value]
        return [{"pitch": p, "onset": a, "offset": b, "velocity": v}
                for p, a, b, v in zip(out["pitch"].tolist(), out["onset"].tolist(), out["offset"].tolist(), out["velocity"].tolist())]

    # ------------------------------------------------------------------ array fast paths (batched jobs: no per-token Python objects)
    @staticmethod
    def event_table(vocab) -> np.ndarray:
        """[vocab_size, 2] int32 (type, value) per token id -- what `vocab.decode_to_event` (vocab.py:120-134) yields, once."""
        n = len(vocab)
        tab = np.empty(n, dtype=_EVENT_DTYPE)
        tab[:] = _events_to_array([vocab.decode_to_event(i) for i in range(n)])
        return tab

    @staticmethod
    def id_lookup(vocab) -> np.ndarray:
        """Inverse of `event_table` for `encode`'s output: int32 [5, 4096] with ``lut[type, value + 2048]`` = what
        ``vocab.encode(Event(type, value))`` returns (the UNK id for events the vocabulary does not hold, vocab.py:51-56)."""
        unk = vocab.token_to_id.get("<UNK>")
        lut = np.full((5, 4096), -1 if unk is None else int(unk), np.int32)
        for tok, i in vocab.token_to_id.items():
            t, _, v = tok.rpartition("_")
            if t == "Bar" and v in ("BOS", "EOS"):
                lut[0, 2048 + (1 if v == "BOS" else 0)] = i
            elif t in _EV_TYPES[1:]:
                try:
                    iv = int(v)
                except ValueError:
                    continue
                if -2048 <= iv < 2048 and str(iv) == v:
                    lut[_EV_TYPES.index(t), 2048 + iv] = i
        return lut

    @staticmethod
    def events_to_ids(ev: np.ndarray, lut: np.ndarray) -> np.ndarray:
        """`vocab.encode_sequence(events)` (infer.py:182) on an event array from `encode_note_array_to_events`."""
        t, v = ev["type"].astype(np.int64), ev["value"].astype(np.int64)
        if ev.size and (t.min() < 0 or t.max() > 4 or v.min() < -2048 or v.max() >= 2048):
            raise ValueError("events_to_ids: event outside the tokenizer's types / value range")
        ids = lut[t, v + 2048]
        if ev.size and ids.min() < 0:
            raise ValueError("Token is not in the vocabulary, and no '<UNK>' is defined")
        return ids

    @staticmethod
    def split_ids_into_packed_bars(ids: np.ndarray, bar_bos_id: int, bar_eos_id: int):
        """`split_sequence_into_bars` (tokenizer.py:43-76) returning (ids int32, offsets int32 [n_bars + 1]) -- the arrays a
        `decoder.PackedBars` holds."""
        ids = np.ascontiguousarray(ids, np.int32)
        n = int(ids.size)
        out = np.empty(n + 1, np.int32)
        offs = np.empty(n + 2, np.int64)
        nb = C.c_longlong()
        _lib.check(_lib.lib().etd_tok_split_bars(ids.ctypes.data, n, int(bar_bos_id), int(bar_eos_id), out.ctypes.data, n + 1, offs.ctypes.data, n + 2,
                                                 C.byref(nb)), "etd_tok_split_bars")
        k = int(nb.value)
        return out[: int(offs[k])].copy(), offs[: k + 1].astype(np.int32)

    def decode_ids_to_note_array(self, ids: Sequence[int], table: np.ndarray, volume: Optional[np.ndarray] = None, pad_id: int = 0) -> np.ndarray:
        """`decode_to_notes(vocab.decode_sequence_to_events(ids))` as arrays: token ids -> NOTE_DTYPE array (same notes, same order)."""
        ids = np.asarray(ids, np.int64)
        ev = np.ascontiguousarray(table[ids[ids != pad_id]])
        n = int(ev.size)
        vol = None if volume is None else np.ascontiguousarray(volume, np.float64)
        cap = 2 * n + 256
        while True:
            out = np.empty(cap, dtype=NOTE_DTYPE)
            k = C.c_longlong()
            rc = _lib.lib().etd_tok_decode(self._h, ev.ctypes.data, n, vol.ctypes.data if vol is not None else None, int(vol.size) if vol is not None else 0,
                                           out.ctypes.data, cap, C.byref(k))
            if rc == -12 and k.value > cap:
                cap = int(k.value)
                continue
            _lib.check(rc, "etd_tok_decode")
            return out[: k.value]

    def encode_note_array_to_events(self, notes: np.ndarray, with_grace_note: bool = False) -> np.ndarray:
        """`encode` as arrays: NOTE_DTYPE notes -> (type, value) event array (Bar BOS/EOS = (0, 1)/(0, 0))."""
        notes = np.ascontiguousarray(notes, dtype=NOTE_DTYPE)
        n = int(notes.size)
        cap = 5 * n + 2 * len(self.global_measures) + 8
        ev = np.empty(cap, dtype=_EVENT_DTYPE)
        k = C.c_longlong()
        _lib.check(_lib.lib().etd_tok_encode(self._h, notes.ctypes.data, n, 1 if with_grace_note else 0, ev.ctypes.data, cap, C.byref(k)), "etd_tok_encode")
        return ev[: k.value]

    @staticmethod
    def note_to_midi(note_list, output_path: Union[str, Path]):
        """tokenizer.py:499-524: notes -> .mid, laid out as pretty_midi writes `PrettyMIDI()` + one `Instrument(program=0)`.

        Native writer (csrc/midi.cpp, ``etd_midi_write``): pretty_midi / mido are not needed.  ``note_list`` is the list of
        {pitch, onset, offset, velocity} dicts `decode_to_notes` returns, or a NOTE_DTYPE array."""
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        if isinstance(note_list, np.ndarray) and note_list.dtype == NOTE_DTYPE:
            notes = np.ascontiguousarray(note_list)
        else:
            notes = np.empty(len(note_list), dtype=NOTE_DTYPE)
            for i, nd in enumerate(note_list):
                notes[i] = (nd["onset"], nd["offset"], int(nd["pitch"]), int(nd["velocity"]))
        _lib.check(_lib.lib().etd_midi_write(notes.ctypes.data if notes.size else None, int(notes.size), str(output_path).encode()), "etd_midi_write")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().etd_tok_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

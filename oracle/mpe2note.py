"""Oracle: frame-wise predictions -> note list.  TEST INFRASTRUCTURE ONLY.

Restates ``AMTAPC_Extractor._mpe2note`` (etude/data/extractor.py:256-418) and
``_note2json`` (etude/data/extractor.py:432-446) with run-length peak detection instead
of the reference's per-frame neighbour scans; results are identical (pinned by
tests/golden/mpe2note_*.json).

Numeric contract kept from the reference as it executes under numpy>=2 (NEP 50; the
reference only requires ``numpy>=1.24``, pyproject.toml:45, and this image has 2.2):
``hop_sec`` and ``i*hop_sec`` are Python floats; the three-point interpolation mixes them
with np.float32 array scalars, so the interpolated peak time is computed in **float32**
(``fp32(i*hop) -/+ fp32(hop/2) * (a-b) / (c-b)``) and only then widened by ``float()``;
un-interpolated times (edges, symmetric neighbours, mpe offsets) stay float64.
We keep that by doing the arithmetic on the same scalar types in the same order.
"""
from __future__ import annotations

import json
from typing import Dict, List

import numpy as np


def _peaks(x: np.ndarray, thr: float, hop_sec: float):
    """extractor.py:267-296 / :297-326 for one pitch column x[T] (np.float32).

    A frame is a peak when x>=thr and the nearest *different* value on each side is lower
    (so every frame of a qualifying plateau is a peak).  Returns (locs, times).
    """
    T = len(x)
    if T == 0:
        return [], []
    change = np.flatnonzero(x[1:] != x[:-1]) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [T]]) - 1
    v = x[starts]
    left_ok = np.ones(len(starts), bool)
    left_ok[1:] = v[1:] > x[starts[1:] - 1]
    right_ok = np.ones(len(starts), bool)
    right_ok[:-1] = v[:-1] > x[ends[:-1] + 1]
    ok = left_ok & right_ok & (v >= thr)
    locs: List[int] = []
    times: List[float] = []
    for s, e in zip(starts[ok], ends[ok]):
        for i in range(int(s), int(e) + 1):
            if i == 0 or i == T - 1:
                t = i * hop_sec
            else:
                a, b, c = x[i - 1], x[i + 1], x[i]
                if a == b:
                    t = i * hop_sec
                elif a > b:
                    t = (i * hop_sec - (hop_sec * 0.5 * (a - b) / (c - b)))
                else:
                    t = (i * hop_sec + (hop_sec * 0.5 * (b - a) / (c - a)))
            locs.append(i)
            times.append(t)
    return locs, times


def mpe2note(a_onset, a_offset, a_mpe, a_velocity, thred_onset=0.5, thred_offset=0.5, thred_mpe=0.5,
             hop_sample: int = 256, sr: int = 16000, note_min: int = 21,
             mode_velocity: str = "ignore_zero", mode_offset: str = "shorter") -> List[Dict]:
    """extractor.py:256-418."""
    hop_sec = float(hop_sample / sr)
    T, num_note = a_onset.shape
    notes: List[Dict] = []
    for j in range(num_note):
        on_loc, on_time = _peaks(np.ascontiguousarray(a_onset[:, j]), thred_onset, hop_sec)
        off_loc, off_time = _peaks(np.ascontiguousarray(a_offset[:, j]), thred_offset, hop_sec)
        off_loc_arr = np.asarray(off_loc, dtype=np.int64)
        mpe_col = a_mpe[:, j]
        below = np.flatnonzero(mpe_col < thred_mpe)
        for k, (loc_on, t_on) in enumerate(zip(on_loc, on_time)):
            if k + 1 < len(on_loc):
                loc_next, t_next = on_loc[k + 1], on_time[k + 1]
            else:
                loc_next, t_next = len(a_mpe), (len(a_mpe) - 1) * hop_sec
            # first offset peak strictly after the onset (extractor.py:345-356)
            p = int(np.searchsorted(off_loc_arr, loc_on, side="right"))
            flag_off = p < len(off_loc)
            loc_off, t_off = (off_loc[p], off_time[p]) if flag_off else (loc_on + 1, 0.0)
            if loc_off > loc_next:
                loc_off, t_off = loc_next, t_next
            # first frame in (loc_on, loc_next) with mpe below threshold (extractor.py:360-368)
            q = int(np.searchsorted(below, loc_on, side="right"))
            flag_mpe = q < len(below) and below[q] < loc_next
            loc_mpe = int(below[q]) if flag_mpe else loc_on + 1
            t_mpe = loc_mpe * hop_sec
            vel = int(a_velocity[loc_on][j])
            if not flag_off and not flag_mpe:
                off_val = float(t_next)
            elif flag_off and not flag_mpe:
                off_val = float(t_off)
            elif flag_mpe and not flag_off:
                off_val = float(t_mpe)
            elif mode_offset == "offset":
                off_val = float(t_off)
            elif mode_offset == "longer":
                off_val = float(t_off) if loc_off >= loc_mpe else float(t_mpe)
            else:
                off_val = float(t_off) if loc_off <= loc_mpe else float(t_mpe)
            if mode_velocity != "ignore_zero" or vel > 0:
                notes.append({"pitch": int(j + note_min), "onset": float(t_on), "offset": off_val, "velocity": vel})
            # same-pitch overlap clip (extractor.py:411-414)
            if len(notes) > 1 and notes[-1]["pitch"] == notes[-2]["pitch"] and notes[-1]["onset"] < notes[-2]["offset"]:
                notes[-2]["offset"] = notes[-1]["onset"]
    return sorted(sorted(notes, key=lambda n: n["pitch"]), key=lambda n: n["onset"])


def notes_for_json(notes: List[Dict], min_length: float = 0.0) -> List[Dict]:
    """extractor.py:432-443 (the list that gets json.dump'ed)."""
    return [{"onset": n["onset"], "offset": n["offset"], "pitch": n["pitch"], "velocity": n["velocity"]}
            for n in notes if not (n["offset"] - n["onset"] < min_length)]


def note2json(notes: List[Dict], path: str, min_length: float = 0.0) -> None:
    with open(path, "w", encoding="utf-8") as f:
        json.dump(notes_for_json(notes, min_length), f, ensure_ascii=False, indent=2)

#!/usr/bin/env python3
"""Tokenizer glue, native vs the reference's Python (runs in the build container only: needs /root/reference).
One synthetic 92-bar song: encode once, then decode_to_notes for 27 attribute variants (the configs[4] shape per clip)."""
import json
import sys
import tempfile
import time
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")
for m in ("torchaudio", "pretty_midi"):
    sys.modules.setdefault(m, types.ModuleType(m))

if __name__ == "__main__":
    from etude.data.tokenizer import TinyREMITokenizer as Ref
    from etude_amd.tokenizer import TinyREMITokenizer as Nat
    rng = np.random.default_rng(0)
    tmp = Path(tempfile.mkdtemp())
    tempo = [{"start": 0.5, "bpm": 120, "time_sig": 4, "downbeats": [0.5 + 2.0 * i for i in range(90)]}]
    (tmp / "tempo.json").write_text(json.dumps(tempo))
    notes = []
    for i in range(90 * 8 * 4):
        on = 0.5 + i * 0.0625 + float(rng.uniform(0, 0.01))
        notes.append({"onset": on, "offset": on + float(rng.choice([0.1, 0.25, 0.5, 1.0])), "pitch": int(rng.integers(40, 90)), "velocity": 64})
    (tmp / "extract.json").write_text(json.dumps(notes))
    res = {}
    for name, cls in (("reference", Ref), ("native", Nat)):
        t0 = time.perf_counter()
        tk = cls(str(tmp / "tempo.json"))
        ev = list(tk.encode(str(tmp / "extract.json")))
        t1 = time.perf_counter()
        outs = [cls(str(tmp / "tempo.json")).decode_to_notes(list(ev)) for _ in range(27)]
        t2 = time.perf_counter()
        res[name] = (t1 - t0, t2 - t1, len(ev), len(outs[0]))
        print(f"{name:10s} encode {1e3 * (t1 - t0):8.1f} ms ({len(ev)} events) | 27 x decode_to_notes {1e3 * (t2 - t1):8.1f} ms ({len(outs[0])} notes each)")
    # array fast path of the batched flow: generated ids -> notes without per-token Python objects
    from etude_amd.vocab import Vocab
    v = Vocab()
    evs = list(Nat(str(tmp / "tempo.json")).encode(str(tmp / "extract.json")))
    for e in evs:
        v._add_token(str(e))
    ids = np.asarray(v.encode_sequence(evs), np.int32)
    nat = Nat(str(tmp / "tempo.json"))
    tab = nat.event_table(v)
    t0 = time.perf_counter()
    arrs = [nat.decode_ids_to_note_array(ids, tab) for _ in range(27)]
    t1 = time.perf_counter()
    ref = Ref(str(tmp / "tempo.json"))
    t2 = time.perf_counter()
    refn = [Ref(str(tmp / "tempo.json")).decode_to_notes(v.decode_sequence_to_events(ids.tolist())) for _ in range(27)]
    t3 = time.perf_counter()
    same = [{"pitch": int(p), "onset": float(a), "offset": float(b), "velocity": int(w)} for p, a, b, w in zip(arrs[0]["pitch"], arrs[0]["onset"], arrs[0]["offset"], arrs[0]["velocity"])] == \
        [{k: n[k] for k in ("pitch", "onset", "offset", "velocity")} for n in refn[0]]
    print(f"ids -> notes, 27 jobs: reference {1e3 * (t3 - t2):.1f} ms, native arrays {1e3 * (t1 - t0):.1f} ms ({(t3 - t2) / (t1 - t0):.0f}x), identical: {same}")
    keep = lambda ns: [{k: n[k] for k in ("pitch", "onset", "offset", "velocity")} for n in ns]      # noqa: E731
    a = keep(Ref(str(tmp / "tempo.json")).decode_to_notes(list(Ref(str(tmp / "tempo.json")).encode(str(tmp / "extract.json")))))
    b = Nat(str(tmp / "tempo.json")).decode_to_notes(list(Nat(str(tmp / "tempo.json")).encode(str(tmp / "extract.json"))))
    print("identical:", a == b, f"| encode speedup {res['reference'][0] / res['native'][0]:.1f}x, decode speedup {res['reference'][1] / res['native'][1]:.1f}x")

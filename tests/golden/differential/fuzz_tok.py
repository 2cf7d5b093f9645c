import sys, types, json, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/reference")   # this container only
for _m in ("torchaudio", "pretty_midi", "librosa", "madmom", "mido"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
from etude.data.tokenizer import TinyREMITokenizer as Ref
from etude.data.vocab import Event as REvent
from etude_amd.tokenizer import TinyREMITokenizer as Mine
from etude_amd.vocab import Event as MEvent
tmp = Path(tempfile.mkdtemp())
bad = 0; n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for case in range(n_cases):
    r = np.random.default_rng(50000 + case)
    # random tempo map: 1-3 regions, jittered downbeats, occasional empty region, time sig 3/4
    tempo = []; t = float(r.uniform(0, 2))
    for reg in range(int(r.integers(1, 4))):
        bpm = float(r.choice([60, 72.5, 90, 100, 120, 140, 171.3])); ts = int(r.choice([3, 4]))
        nb = int(r.integers(0, 12)) if reg else int(r.integers(2, 12))
        per = ts * 60.0 / bpm
        db = np.round(t + np.cumsum(np.r_[0.0, per * r.uniform(0.93, 1.07, max(nb - 1, 0))]), 5).tolist() if nb else []
        tempo.append({"start": round(t, 5), "bpm": bpm, "time_sig": ts, "downbeats": db})
        t = (db[-1] if db else t) + per * float(r.uniform(0.5, 1.5))
    t_end = t + 2
    notes = []; tt = max(0.0, tempo[0]["start"] - 1.0)
    while tt < t_end:
        for c in range(int(r.integers(1, 5))):
            dur = float(r.choice([0.03, 0.05, 0.12, 0.25, 0.5, 0.9, 1.7, 3.1, 6.0]))
            notes.append({"onset": round(float(tt + (0.0 if r.random() < 0.6 else r.uniform(0, 0.04))), 6), "offset": round(float(tt + dur), 6),
                          "pitch": int(r.integers(21, 109)), "velocity": int(r.integers(1, 127))})
        if r.random() < 0.3:
            notes.append({"onset": round(float(tt - r.uniform(0.01, 0.1)), 6), "offset": round(float(tt), 6), "pitch": int(np.clip(notes[-1]["pitch"] + int(r.choice([-2, -1, 1, 2])), 21, 108)), "velocity": 40})
        if r.random() < 0.1: notes.append(dict(notes[-1]))
        tt += float(r.choice([0.0625, 0.125, 0.25, 0.25, 0.5, 1.0, 2.0]))
    r.shuffle(notes)
    tp = tmp / "tempo.json"; tp.write_text(json.dumps(tempo)); mp = tmp / "ex.json"; mp.write_text(json.dumps(notes))
    for grace in (False, True):
        try:
            ref = Ref(str(tp)); ev_r = [(e.type_, e.value) for e in ref.encode(str(mp), with_grace_note=grace)]
        except Exception as e:
            ev_r = ("EXC", type(e).__name__)
        try:
            mine = Mine(str(tp)); ev_m = [(e.type_, e.value) for e in mine.encode(str(mp), with_grace_note=grace)]
        except Exception as e:
            ev_m = ("EXC", type(e).__name__)
        if ev_r != ev_m:
            bad += 1; print("ENCODE MISMATCH case", case, grace, (ev_r if isinstance(ev_r, tuple) else len(ev_r)), (ev_m if isinstance(ev_m, tuple) else len(ev_m)))
            if not isinstance(ev_r, tuple) and not isinstance(ev_m, tuple):
                for i, (a, b) in enumerate(zip(ev_r, ev_m)):
                    if a != b: print("  first diff at", i, a, b); break
            continue
        if isinstance(ev_r, tuple): continue
        # decode: with extra grace events + volume map
        dec_in = []
        for (ty, va) in ev_r:
            if ty == "Note" and r.random() < 0.3: dec_in.append(("Grace", int(r.choice([-1, 1]))))
            dec_in.append((ty, va))
        vol = np.round(r.random(int(20 * (t_end + 2))) ** 2, 4).tolist(); vp = tmp / "vol.json"; vp.write_text(json.dumps(vol))
        for vpath in (None, str(vp)):
            keep = lambda ns: [(n["pitch"], n["onset"], n["offset"], n["velocity"]) for n in ns]
            try: d_r = keep(Ref(str(tp)).decode_to_notes([REvent(type_=a, value=b) for a, b in dec_in], volume_map_path=vpath))
            except Exception as e: d_r = ("EXC", type(e).__name__)
            try: d_m = keep(Mine(str(tp)).decode_to_notes([MEvent(type_=a, value=b) for a, b in dec_in], volume_map_path=vpath))
            except Exception as e: d_m = ("EXC", type(e).__name__)
            if d_r != d_m:
                bad += 1; print("DECODE MISMATCH case", case, grace, vpath is not None, (d_r if isinstance(d_r, tuple) else len(d_r)), (d_m if isinstance(d_m, tuple) else len(d_m)))
                if not isinstance(d_r, tuple) and not isinstance(d_m, tuple):
                    for i, (a, b) in enumerate(zip(d_r, d_m)):
                        if a != b: print("  first diff at", i, a, b); break
print("cases", n_cases, "mismatches", bad)

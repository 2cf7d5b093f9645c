import sys, types, json, tempfile, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/reference")   # this container only
for _m in ("torchaudio", "pretty_midi", "librosa", "madmom", "mido"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
from etude.data.vocab import Vocab as RV, Event as RE
from etude_amd.vocab import Vocab as MV, Event as ME
import inspect
print([n for n, _ in inspect.getmembers(RV, predicate=inspect.isfunction)])
bad = 0
for case in range(200):
    r = np.random.default_rng(case)
    types_ = ["Bar", "Pos", "Note", "Duration", "Grace", "Weird"]
    evs = []
    for _ in range(int(r.integers(0, 300))):
        t = str(r.choice(types_))
        v = ("BOS" if r.random() < 0.5 else "EOS") if t == "Bar" else (int(r.integers(-3, 130)) if r.random() < 0.9 else str(r.integers(0, 5)))
        evs.append((t, v))
    rv, mv = RV(), MV()
    try: rv.build_from_events([RE(type_=a, value=b) for a, b in evs]); r_ok = True
    except Exception as e: r_ok = repr(e)
    try: mv.build_from_events([ME(type_=a, value=b) for a, b in evs]); m_ok = True
    except Exception as e: m_ok = repr(e)
    if (r_ok is True) != (m_ok is True): bad += 1; print("build mismatch", case, r_ok, m_ok); continue
    if r_ok is not True: continue
    if rv.token_to_id != mv.token_to_id or len(rv) != len(mv): bad += 1; print("vocab mismatch", case); continue
    seq = [evs[int(i)] for i in r.integers(0, max(len(evs), 1), 50)] if evs else []
    seq.append(("Nope", 7))
    a = rv.encode_sequence([RE(type_=x, value=y) for x, y in seq]); b = mv.encode_sequence([ME(type_=x, value=y) for x, y in seq])
    if a != b: bad += 1; print("encode mismatch", case)
    ids = r.integers(-2, len(rv) + 3, 40).tolist()
    try: da = [(e.type_, e.value) for e in rv.decode_sequence_to_events(ids)]
    except Exception as e: da = ("EXC", type(e).__name__, str(e))
    try: db = [(e.type_, e.value) for e in mv.decode_sequence_to_events(ids)]
    except Exception as e: db = ("EXC", type(e).__name__, str(e))
    if da != db: bad += 1; print("decode mismatch", case, da[:5] if isinstance(da, list) else da, db[:5] if isinstance(db, list) else db)
    ids2 = r.integers(0, len(rv), 40).tolist() if len(rv) else []
    da = [(e.type_, e.value) for e in rv.decode_sequence_to_events(ids2)]; db = [(e.type_, e.value) for e in mv.decode_sequence_to_events(ids2)]
    if da != db: bad += 1; print("decode2 mismatch", case, da[:5], db[:5])
    for f in ("get_bar_bos_id", "get_bar_eos_id", "get_pad_id"):
        if hasattr(rv, f) and getattr(rv, f)() != getattr(mv, f)(): bad += 1; print(f, "mismatch", case)
    d = tempfile.mkdtemp(); rv.save(os.path.join(d, "r.json")); mv.save(os.path.join(d, "m.json"))
    if json.load(open(os.path.join(d, "r.json"))) != json.load(open(os.path.join(d, "m.json"))): bad += 1; print("save mismatch", case)
    m2 = MV.load(os.path.join(d, "r.json")); r2 = RV.load(os.path.join(d, "m.json"))
    if m2.token_to_id != r2.token_to_id: bad += 1; print("load mismatch", case)
print("mismatches", bad)

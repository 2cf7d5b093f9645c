import sys, types
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/reference")   # this container only
for _m in ("torchaudio", "pretty_midi", "librosa", "madmom", "mido"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
from etude.data.tokenizer import TinyREMITokenizer as Ref
from etude_amd.tokenizer import TinyREMITokenizer as Mine
r_, m_ = Ref(None), Mine(None)
bad = 0
for case in range(2000):
    r = np.random.default_rng(case)
    n = int(r.integers(0, 80)); hi = int(r.choice([3, 6, 12]))
    ids = r.integers(0, hi, n).tolist()
    bos, eos = int(r.integers(-1, hi)), int(r.integers(-1, hi))
    a = r_.split_sequence_into_bars(list(ids), bos, eos); b = m_.split_sequence_into_bars(list(ids), bos, eos)
    if a != b: bad += 1; print("mismatch", case, ids, bos, eos, a, b)
    if bad > 5: break
print("mismatches", bad)

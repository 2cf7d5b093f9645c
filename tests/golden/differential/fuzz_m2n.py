import sys, types, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/reference")   # this container only
for _m in ("torchaudio", "pretty_midi", "librosa", "madmom", "mido"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
from etude.data.extractor import AMTAPC_Extractor as RefExt
from etude_amd import _lib
from etude_amd.extractor import NOTE_DTYPE
lib = _lib.lib()
class Cfg: pass
def ref_m2n(on, off, mpe, vel, th):
    self = types.SimpleNamespace()
    inp = types.SimpleNamespace(hop_sample=256, sr=16000); midi = types.SimpleNamespace(note_min=21)
    self.config = {"input": {"hop_sample": 256, "sr": 16000}, "midi": {"note_min": 21, "num_note": on.shape[1]}}
    return RefExt._mpe2note(self, a_onset=on, a_offset=off, a_mpe=mpe, a_velocity=vel, thred_onset=th[0], thred_offset=th[1], thred_mpe=th[2])
def mine_m2n(on, off, mpe, vel, th):
    T, nn = on.shape
    cap = 4 * T * nn // 2 + 64
    out = np.empty(cap, dtype=NOTE_DTYPE); k = C.c_longlong()
    on = np.ascontiguousarray(on, np.float32); off = np.ascontiguousarray(off, np.float32); mpe = np.ascontiguousarray(mpe, np.float32); vel = np.ascontiguousarray(vel, np.int8)
    _lib.check(lib.etd_mpe2note(on.ctypes.data, off.ctypes.data, mpe.ctypes.data, vel.ctypes.data, T, nn, th[0], th[1], th[2], 256, 16000, 21, out.ctypes.data, cap, C.byref(k)), "m2n")
    return out[:k.value]
bad = 0; N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for case in range(N):
    r = np.random.default_rng(7000 + case)
    T = int(r.integers(5, 400)); nn = int(r.choice([4, 12, 88]))
    def field(smooth):
        x = r.random((T, nn)).astype(np.float32)
        if smooth:
            k = int(r.integers(2, 9))
            x = np.apply_along_axis(lambda v: np.convolve(v, np.ones(k) / k, mode="same"), 0, x).astype(np.float32)
        x = (x - x.min()) / (x.max() - x.min() + 1e-9)
        if r.random() < 0.5: x = np.round(x, int(r.integers(1, 3))).astype(np.float32)     # plateaus / exact ties
        return x.astype(np.float32)
    on, off, mpe = field(r.random() < 0.7), field(r.random() < 0.7), field(True)
    vel = r.integers(0, 128, (T, nn)).astype(np.int8)
    th = (float(r.choice([0.3, 0.5, 0.7])), float(r.choice([0.3, 0.5, 0.7])), float(r.choice([0.3, 0.5, 0.6])))
    try: ref = ref_m2n(on, off, mpe, vel, th)
    except Exception as e: ref = ("EXC", repr(e)[:80])
    try: got = mine_m2n(on, off, mpe, vel, th)
    except Exception as e: got = ("EXC", repr(e)[:80])
    if isinstance(ref, tuple) or isinstance(got, tuple):
        if not (isinstance(ref, tuple) and isinstance(got, tuple)): bad += 1; print("EXC mismatch", case, ref if isinstance(ref, tuple) else len(ref), got if isinstance(got, tuple) else len(got))
        continue
    rl = [(n["pitch"], float(n["onset"]), float(n["offset"]), int(n["velocity"])) for n in ref]
    gl = [(int(n["pitch"]), float(n["onset"]), float(n["offset"]), int(n["velocity"])) for n in got]
    if rl != gl:
        bad += 1; print("MISMATCH case", case, "T", T, "nn", nn, len(rl), len(gl))
        for i, (a, b) in enumerate(zip(rl, gl)):
            if a != b: print("  first diff", i, a, b); break
print("cases", N, "mismatches", bad)

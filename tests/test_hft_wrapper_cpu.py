"""HFT_Transformer wrapper, CPU side: the oracle's restatement of `_transcript_stride` / `_transcript`
against vectors captured from the reference class (etude/models/hft_transformer.py), its second `_mpe2note` copy against
the oracle and the C ABI, and the checkpoint loader against a plain-pickled model object."""
import ctypes as C
import json

import numpy as np

from etude_amd import _lib, synth
from oracle import hft, mpe2note
from tests._util import TINY_EXT, hft_dims, torch_sd


def test_oracle_transcript_stride_and_plain_tiny(golden_dir):
    g = np.load(golden_dir / "hft_wrapper_tiny.npz")
    d = hft_dims(TINY_EXT)
    sd = torch_sd(synth.extractor_state_dict(13, TINY_EXT))
    so = hft.transcript_stride(sd, g["feature"], d, n_offset=4, min_value=-80.0)
    to = hft.transcript(sd, g["feature"], d, min_value=-80.0)
    assert so[0].shape == g["stride0"].shape == (32, 12)        # 27 frames -> 4 half-windows of 8
    for i in range(8):
        for got, want in ((so[i], g[f"stride{i}"]), (to[i], g[f"plain{i}"])):
            if got.dtype == np.int8:
                assert (got == want).mean() > 0.99
            else:
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)


def test_wrapper_mpe2note_copy_equals_oracle_and_capi(golden_dir):
    c = json.loads((golden_dir / "hft_wrapper_mpe2note.json").read_text())
    on, off, mp = (np.ascontiguousarray(c[k], np.float32) for k in ("onset", "offset", "mpe"))
    vel = np.ascontiguousarray(c["velocity"], np.int8)
    assert mpe2note.mpe2note(on, off, mp, vel, *c["thr"]) == c["notes"]
    lib = _lib.lib()
    T, nn = on.shape
    buf = (_lib.Note * (T * nn))()
    n = C.c_longlong()
    _lib.check(lib.etd_mpe2note(on.ctypes.data, off.ctypes.data, mp.ctypes.data, vel.ctypes.data, T, nn, c["thr"][0], c["thr"][1], c["thr"][2],
                                256, 16000, 21, buf, T * nn, C.byref(n)), "etd_mpe2note")
    got = [{"pitch": int(b.pitch), "onset": float(b.onset), "offset": float(b.offset), "velocity": int(b.velocity)} for b in buf[: n.value]]
    assert got == c["notes"]


def test_checkpoint_loader_reads_a_pickled_model_object(golden_dir):
    from etude_amd.hft_transformer import load_hft_state
    want = np.load(golden_dir / "hft_tiny_model_state.npz")
    got = load_hft_state(golden_dir / "hft_tiny_model.pkl")
    assert set(got) >= set(want.files)
    for k in want.files:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]), k
    extra = set(got) - set(want.files)
    assert all(k.endswith("scale") or "scale" in k for k in extra), extra      # registered sqrt(d) buffers, unused by the kernels


def test_checkpoint_loader_rejects_foreign_classes(tmp_path):
    import pickle

    import pytest

    from etude_amd.hft_transformer import load_hft_state
    p = tmp_path / "evil.pkl"
    p.write_bytes(pickle.dumps(print))
    with pytest.raises(Exception):
        load_hft_state(p)

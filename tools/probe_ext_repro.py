#!/usr/bin/env python3
"""Is the Extract stage's output the same alone and beside decoding engines?  The extractor transcribes the same 4 windows R times
alone, then R times while two engines run batched prefills and decode steps; prints how many distinct output digests each phase saw.
usage: probe_ext_repro.py [reps=40]"""
import hashlib
import sys
import threading
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.config import ExtractorConfig  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402
from etude_amd.extractor import AMTAPC_Extractor  # noqa: E402

if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    S, T0 = 54, 340
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=1088)]
    decs.append(decs[0].clone())
    ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(7), "cuda", max_windows=4)
    xs = torch.from_numpy(synth.window_features(5, 4)).to(dev)
    est = torch.cuda.Stream(device=dev)
    slots = np.arange(S, dtype=np.int32)
    tg = np.tile(np.asarray([2, 1, 1, 1], np.int32), S)
    eos = np.full(S, -1, np.int32)
    stop = [False]

    def engine(e):
        torch.cuda.set_device(0)
        dec = decs[e]
        i = 0
        while not stop[0]:
            rng = np.random.default_rng(1000 * e + i); i += 1
            Ts = np.asarray([T0 - (s % 9) for s in range(S)], np.int32)
            M = int(Ts.sum())
            ids, cls, a4 = rng.integers(6, 154, M).astype(np.int32), rng.integers(1, 3, M).astype(np.int32), rng.integers(0, 3, (4, M)).astype(np.int32)
            lim = np.full(S, 200, np.int32)
            _lib.check(lib.etd_decoder_begin_bars(dec._h, S, slots.ctypes.data, Ts.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tg.ctypes.data,
                                                  eos.ctypes.data, lim.ctypes.data, dec._stream()), "begin_bars")
            _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, 60, dec._stream()), "step")
            dec._ts.synchronize()

    def phase(label):
        seen = {}
        with torch.cuda.stream(est):
            for _ in range(R):
                outs = ex.transcript_windows(xs)
                est.synchronize()
                h = hashlib.sha256()
                for t in outs:
                    h.update(t.detach().cpu().numpy().tobytes())
                seen[h.hexdigest()[:16]] = seen.get(h.hexdigest()[:16], 0) + 1
        print("%s: %d transcriptions, %d distinct output digests: %s" % (label, R, len(seen), seen), flush=True)
        return seen

    a = phase("alone")
    ths = [threading.Thread(target=engine, args=(e,)) for e in (0, 1)]
    for t in ths:
        t.start()
    b = phase("beside two decoding engines")
    stop[0] = True
    for t in ths:
        t.join()
    c = phase("alone again")
    print("REPRODUCIBLE" if len(a) == 1 and a.keys() == b.keys() == c.keys() else "NOT REPRODUCIBLE", flush=True)

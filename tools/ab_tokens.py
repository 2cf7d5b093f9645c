#!/usr/bin/env python3
"""Greedy token streams of E engines x S streams stepping concurrently -> sha256 per engine (compare two builds / env settings:
identical kernels-in-effect must print identical digests).  usage: ab_tokens.py [streams=54] [ctx0=320] [steps=160] [engines=4]"""
import hashlib
import sys
import threading
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 160
    E = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    tg = np.asarray([2, 1, 1, 1], np.int32)
    decs = []
    for e in range(E):
        decs.append(decs[0].clone() if decs else EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=1024))
    slots = np.arange(S, dtype=np.int32)
    for rep in range(3):
        outs = []
        for e, dec in enumerate(decs):
            rng = np.random.default_rng(100 * rep + e)
            st = dec._stream()
            for s in range(S):
                T = ctx0 - (s % 7)
                ids = rng.integers(6, 154, T).astype(np.int32); cls = rng.integers(1, 3, T).astype(np.int32); a4 = rng.integers(0, 3, (4, T)).astype(np.int32)
                _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, T, tg.ctypes.data, -1, steps + 8, st), "begin_bar")
        torch.cuda.synchronize(dev)

        def run(dec):
            torch.cuda.set_device(0)
            _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, steps, dec._stream()), "step")
            torch.cuda.current_stream().synchronize()

        th = [threading.Thread(target=run, args=(d,)) for d in decs]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize(dev)
        for e, dec in enumerate(decs):
            h = hashlib.sha256()
            tot = 0
            for s in range(S):
                import ctypes as C
                buf = np.zeros(steps + 16, np.int32); n = C.c_int()
                _lib.check(lib.etd_decoder_read_tokens(dec._h, s, buf.ctypes.data, len(buf), C.byref(n), dec._stream()), "read")
                toks = buf[: n.value]
                h.update(np.asarray(toks, np.int32).tobytes()); tot += len(toks)
            print(f"rep {rep} engine {e}: {tot} tokens sha {h.hexdigest()[:16]}")

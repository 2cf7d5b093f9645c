#!/usr/bin/env python3
"""How well do independent decoder engines overlap?  E engines x S streams, decode steps only (hipGraph replays), ctx ~340."""
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 72
    ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    import os
    MAXCTX = int(os.environ.get("ETD_BENCH_MAXCTX", "1024"))       # KV positions per stream: sets the stride between (slot, head) regions of the caches
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    tg = np.asarray([2, 1, 1, 1], np.int32)
    decs = []
    mask_mode = sys.argv[4] if len(sys.argv) > 4 else "none"      # none | thirds | interleave | halves | halves8 : CU masks per engine stream (hipExtStreamCreateWithCUMask)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    for e in range(4):
        dec = decs[0].clone() if decs else EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=MAXCTX)
        if mask_mode != "none":
            bits = [0] * 256
            if mask_mode == "thirds":
                lo, hi = [(0, 88), (88, 176), (176, 256), (0, 256)][e]
                for i in range(lo, hi):
                    bits[i] = 1
            elif mask_mode == "halves":              # every CU hosts exactly two of the four engines, each engine has half the chip
                for i in range(256):
                    bits[i] = 1 if ((i + e) % 4) < 2 else 0
            elif mask_mode == "halves8":             # the same in groups of 8 CUs
                for i in range(256):
                    bits[i] = 1 if (((i // 8) + e) % 4) < 2 else 0
            else:                                   # every third CU
                for i in range(256):
                    bits[i] = 1 if (e == 3 or i % 3 == e) else 0
            words = (C.c_uint32 * 8)(*[sum(bits[w * 32 + b] << b for b in range(32)) for w in range(8)])
            hs = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(hs), 8, words)
            assert rc == 0, rc
            dec._ts = torch.cuda.ExternalStream(hs.value, device=dev)
        decs.append(dec)
    slots = np.arange(S, dtype=np.int32)
    prompts = [(rng.integers(6, 154, ctx0).astype(np.int32), rng.integers(1, 3, ctx0).astype(np.int32), rng.integers(0, 3, (4, ctx0)).astype(np.int32)) for _ in range(S)]

    def reset():
        """every stream of every engine back to a ctx0-token context (each timed run then covers the same contexts)"""
        for dec in decs:
            st = dec._stream()
            for s, (ids, cls, a4) in enumerate(prompts):
                _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, ctx0, tg.ctypes.data, -1, min(1000, MAXCTX - ctx0), st), "begin_bar")
            _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, 4, st), "step")
        torch.cuda.synchronize(dev)

    reset()

    def run(dec):
        torch.cuda.set_device(0)
        _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, steps, dec._stream()), "step")
        torch.cuda.current_stream().synchronize()

    for E in (1, 2, 3, 4, 1):
        reset()
        th = [threading.Thread(target=run, args=(decs[i],)) for i in range(E)]
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t
        print(f"E={E}: {1e3 * dt / steps:.4f} ms per step-round, {E * S * steps / dt / 1e3:.1f} k tok/s aggregate, {E * steps / dt / 1e3:.3f} engine-steps/ms")

#!/usr/bin/env python3
"""What does a CU mask cost ONE decoder engine?  54 streams x ctx 320 on a stream created with hipExtStreamCreateWithCUMask:
step time (hipGraph replays) and per-kernel averages (eager + HIP events), for several mask shapes; then 4 engines, each on its
own quarter, stepping concurrently."""
import ctypes as C
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits, dev):
    words = (C.c_uint32 * 8)(*[sum(bits[w * 32 + b] << b for b in range(32)) for w in range(8)])
    hs = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(hs), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(hs.value, device=dev)


def mask(mode, e):
    if mode == "all":
        return [1] * 256
    if mode == "q_contig":          # bits 64 e .. 64 e + 63
        return [1 if i // 64 == e else 0 for i in range(256)]
    if mode == "q_mod4":            # every fourth bit
        return [1 if i % 4 == e else 0 for i in range(256)]
    if mode == "q_blk8":            # blocks of 8 bits, every fourth block
        return [1 if (i // 8) % 4 == e else 0 for i in range(256)]
    if mode == "q_blk32":
        return [1 if (i // 32) % 4 == e else 0 for i in range(256)]
    if mode == "half_contig":
        return [1 if i // 128 == e % 2 else 0 for i in range(256)]
    raise ValueError(mode)


if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    steps = 96
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    tg = np.asarray([2, 1, 1, 1], np.int32)
    decs = []
    for e in range(4):
        decs.append(decs[0].clone() if decs else EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=1024))
    slots = np.arange(S, dtype=np.int32)
    prompts = [(rng.integers(6, 154, ctx0).astype(np.int32), rng.integers(1, 3, ctx0).astype(np.int32), rng.integers(0, 3, (4, ctx0)).astype(np.int32)) for _ in range(S)]
    plain = [d._ts for d in decs]

    def reset(ds):
        for dec in ds:
            st = dec._stream()
            for s, (ids, cls, a4) in enumerate(prompts):
                _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, ctx0, tg.ctypes.data, -1, min(1000, 1024 - ctx0), st), "begin_bar")
            _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, 4, st), "step")
        torch.cuda.synchronize(dev)

    def run(dec):
        torch.cuda.set_device(0)
        _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, steps, dec._stream()), "step")
        dec._ts.synchronize()

    for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("all", "q_contig", "q_mod4", "q_blk8", "q_blk32", "half_contig")):
        for e in range(4):
            decs[e]._ts = plain[e] if mode == "all" else masked_stream(mask(mode, e), dev)
        # the graphs were captured on another stream: replaying them on this one is fine (a graph launch takes the stream it is given)
        res = []
        for E in (1, 4):
            reset(decs[:E])
            th = [threading.Thread(target=run, args=(decs[i],)) for i in range(E)]
            t = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t
            res.append(f"E={E}: {1e3 * dt / steps:.4f} ms/step-round = {E * steps / dt / 1e3:.2f} engine-steps/ms")
        reset(decs[:1])
        _lib.prof_reset(); _lib.prof_enable(True)
        _lib.check(lib.etd_decoder_step(decs[0]._h, slots.ctypes.data, S, 8, decs[0]._stream()), "step")
        torch.cuda.synchronize(dev)
        _lib.prof_enable(False)
        prof = _lib.prof_report()
        ks = "  ".join(f"{k} {1e3 * v['ms'] / max(1, v['launches']):.1f}us" for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:4])
        print(f"{mode:12s} " + " | ".join(res) + " | solo kernels: " + ks, flush=True)

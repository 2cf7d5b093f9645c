#!/usr/bin/env python3
"""Is the decode step host-bound?  E engine threads each enqueue `steps` hipGraph replays of a decode step (S streams, ctx0);
per thread: time until etd_decoder_step RETURNS (everything enqueued) vs time until the stream has drained."""
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
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 96
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

    def reset():
        for dec in decs:
            st = dec._stream()
            for s, (ids, cls, a4) in enumerate(prompts):
                _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, ctx0, tg.ctypes.data, -1, min(1000, 1024 - ctx0), st), "begin_bar")
            _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, 4, st), "step")
        torch.cuda.synchronize(dev)

    res = {}

    def run(i, dec, gate):
        torch.cuda.set_device(0)
        gate.wait()
        t0 = time.perf_counter()
        _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, steps, dec._stream()), "step")
        t1 = time.perf_counter()
        dec._ts.synchronize()
        t2 = time.perf_counter()
        res[i] = (t1 - t0, t2 - t0)

    for E in (1, 2, 4, 1):
        reset()
        gate = threading.Barrier(E)
        th = [threading.Thread(target=run, args=(i, decs[i], gate)) for i in range(E)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        enq = max(res[i][0] for i in range(E)); tot = max(res[i][1] for i in range(E))
        print(f"S={S} ctx={ctx0} engines={E}: enqueue {1e3 * enq / steps:.4f} ms/step per thread, drained {1e3 * tot / steps:.4f} ms/step -> {E * steps / tot / 1e3:.2f} engine-steps/ms", flush=True)

#!/usr/bin/env python3
"""Do the MFMA-bound batched prefill and the HBM-bound decode steps OVERLAP when they come from two engines?

Engine P loops prefill passes (500 prompts x 513 tokens), engine S loops decode steps (R rows at contexts ~537), first each alone, then both for the same
wall time.  Prints ONE JSON line with the rates alone / together and overlap = rate_P' / rate_P + rate_S' / rate_S: 1.0 = the two time-slice the chip (the
stage costs the sum of its parts), 2.0 = each runs as if alone.

    python tools/bench_overlap_roles.py [--rows 864] [--seconds 3] [--mask-steps N --mask-prefill M]   (CU counts for hipExtStreamCreateWithCUMask streams)"""
import argparse
import ctypes as C
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402


def masked_stream(lo, hi, dev):
    """CUs [lo, hi) of the mask's bit order (bits are dealt over the 8 XCDs first: a run of 8 bits is one CU on every XCD)"""
    hip = C.CDLL("libamdhip64.so")
    bits = [1 if lo <= i < hi else 0 for i in range(256)]
    words = (C.c_uint32 * 8)(*[sum(bits[w * 32 + b] << b for b in range(32)) for w in range(8)])
    hs = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(hs), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(hs.value, device=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=864)
    ap.add_argument("--ctx", type=int, default=513)
    ap.add_argument("--prompts", type=int, default=500)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--mask-steps", type=int, default=0, help="step engine on the first N CUs of the mask order")
    ap.add_argument("--mask-prefill", type=int, default=0, help="prefill engine on the LAST M CUs")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    rng = np.random.default_rng(0)
    M = a.prompts * a.ctx
    S = max(a.rows, a.prompts)
    eS = EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=a.ctx + 600, max_prefill_rows=M)
    eP = eS.clone()
    if a.mask_steps:
        eS._ts = masked_stream(0, a.mask_steps, dev)
    if a.mask_prefill:
        eP._ts = masked_stream(256 - a.mask_prefill, 256, dev)
    ids = rng.integers(6, 154, M).astype(np.int32); cls = rng.integers(1, 3, M).astype(np.int32)
    a4 = np.ascontiguousarray(rng.integers(0, 3, (4, M)).astype(np.int32))
    T = np.full(a.prompts, a.ctx, np.int32)
    tgt = np.ascontiguousarray(np.tile(np.asarray([2, 1, 1, 1], np.int32), (a.prompts, 1)))
    eos = np.full(a.prompts, -1, np.int32); lim = np.full(a.prompts, 560, np.int32)

    def prefill(dec, s0, n):
        sl = np.arange(s0, s0 + n, dtype=np.int32)
        _lib.check(lib.etd_decoder_begin_bars(dec._h, n, sl.ctypes.data, T.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tgt.ctypes.data,
                                              eos.ctypes.data, lim.ctypes.data, dec._stream()), "begin_bars")
    # the step engine's rows: prompts of `ctx` tokens in slots 0 .. rows-1
    for s0 in range(0, a.rows, a.prompts):
        prefill(eS, s0, min(a.prompts, a.rows - s0))
    slots = np.arange(a.rows, dtype=np.int32)
    _lib.check(lib.etd_decoder_step(eS._h, slots.ctypes.data, a.rows, 4, eS._stream()), "step")
    prefill(eP, 0, a.prompts)
    torch.cuda.synchronize(dev)

    stop = threading.Event()
    counts = {"P": 0}
    NCALL, PER = 20, 16              # the step engine's measured stretch: 320 steps from the prompts' end (contexts ctx .. ctx + 320 in every phase)

    def loop_p():
        torch.cuda.set_device(0)
        while not stop.is_set():
            prefill(eP, 0, a.prompts)
            eP._ts.synchronize()
            counts["P"] += 1

    def steps_timed():
        t = time.perf_counter()
        for _ in range(NCALL):
            _lib.check(lib.etd_decoder_step(eS._h, slots.ctypes.data, a.rows, PER, eS._stream()), "step")
        eS._ts.synchronize()
        return time.perf_counter() - t

    def reset_steps():
        for s0 in range(0, a.rows, a.prompts):
            prefill(eS, s0, min(a.prompts, a.rows - s0))
        torch.cuda.synchronize(dev)

    # P alone
    counts["P"] = 0; stop.clear()
    th = threading.Thread(target=loop_p); t = time.perf_counter(); th.start(); time.sleep(a.seconds); stop.set(); th.join()
    p_alone = counts["P"] / (time.perf_counter() - t)
    # S alone
    reset_steps()
    s_alone = NCALL * PER / steps_timed()
    # together: P loops while S does the same 320 steps
    reset_steps()
    counts["P"] = 0; stop.clear()
    th = threading.Thread(target=loop_p); th.start()
    time.sleep(0.1)
    c0 = counts["P"]; t = time.perf_counter()
    dt_s = steps_timed()
    p_both = (counts["P"] - c0) / (time.perf_counter() - t)
    stop.set(); th.join()
    s_both = NCALL * PER / dt_s
    torch.cuda.synchronize(dev)
    out = {"rows": a.rows, "ctx": a.ctx, "mask_steps": a.mask_steps, "mask_prefill": a.mask_prefill,
           "prefill_passes_per_s_alone": round(p_alone, 2), "steps_per_s_alone": round(s_alone, 1),
           "prefill_passes_per_s_together": round(p_both, 2), "steps_per_s_together": round(s_both, 1),
           "overlap": round(p_both / max(p_alone, 1e-9) + s_both / max(s_alone, 1e-9), 3)}
    print(json.dumps(out))
    eP.close(); eS.close()


if __name__ == "__main__":
    main()

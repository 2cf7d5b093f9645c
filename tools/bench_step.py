#!/usr/bin/env python3
"""Decode-step microbenchmark: E engines x R rows at context C, hipGraph-replayed steps.

    python tools/bench_step.py --rows 432 --ctx 537 [--engines 1] [--steps 48] [--pair -1|0|1]

Prints ONE JSON line: ms per step (every engine's rows advanced by one token), the step's SURVEY-8(d) roofline fraction, the
device-stamped span of k_dstep_attn_down (its own duration in this configuration) with ITS roofline fraction, and an eager
HIP-event breakdown of one engine's kernels.  Kernel variants are selected by the library's environment switches
(ETD_NO_ATTN_DOWN, ETD_AD_WAVES, ...) or --pair."""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=432)
    ap.add_argument("--ctx", type=int, default=537)
    ap.add_argument("--engines", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--pair", type=int, default=-1)
    ap.add_argument("--max-ctx", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    max_ctx = a.max_ctx or max(2112, a.ctx + a.steps * 3 + 64)
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=a.rows, max_ctx=max_ctx, max_prefill_rows=max(a.rows * 64, 8192))]
    while len(decs) < a.engines:
        decs.append(decs[0].clone())
    rng = np.random.default_rng(0)
    tg = np.asarray([2, 1, 1, 1], np.int32)
    slots = np.arange(a.rows, dtype=np.int32)
    for dec in decs:
        _lib.check(lib.etd_debug_decoder_force_pair(dec._h, a.pair), "force_pair")
        st = dec._stream()
        per = max(1, 8192 // a.ctx)
        for s0 in range(0, a.rows, per):
            n = min(per, a.rows - s0)
            T = np.full(n, a.ctx, np.int32)
            ids = rng.integers(6, 154, n * a.ctx).astype(np.int32); cls = rng.integers(1, 3, n * a.ctx).astype(np.int32)
            a4 = rng.integers(0, 3, (4, n * a.ctx)).astype(np.int32)
            tgt = np.ascontiguousarray(np.tile(tg, (n, 1))); eos = np.full(n, -1, np.int32); lim = np.full(n, 1000, np.int32)
            sl = np.ascontiguousarray(slots[s0: s0 + n])
            _lib.check(lib.etd_decoder_begin_bars(dec._h, n, sl.ctypes.data, T.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tgt.ctypes.data,
                                                  eos.ctypes.data, lim.ctypes.data, st), "begin_bars")
        _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, a.rows, 4, st), "step")      # capture + warm
    torch.cuda.synchronize(dev)

    def run_all(n_steps):
        errs = []
        gate = threading.Barrier(len(decs) + 1, timeout=120)

        def run(e):
            try:
                torch.cuda.set_device(dev)
                gate.wait()
                _lib.check(lib.etd_decoder_step(decs[e]._h, slots.ctypes.data, a.rows, n_steps, decs[e]._stream()), "step")
                decs[e]._ts.synchronize()
            except Exception as ex:      # noqa: BLE001
                errs.append(ex)
        th = [threading.Thread(target=run, args=(e,)) for e in range(len(decs))]
        for x in th:
            x.start()
        gate.wait()
        t = time.perf_counter()
        for x in th:
            x.join()
        torch.cuda.synchronize(dev)
        if errs:
            raise errs[0]
        return time.perf_counter() - t

    for d in decs:
        d.stats_reset()
    dt = run_all(a.steps)
    st = [d.stats() for d in decs]
    bytes_all = sum(s["kv_bytes"] + s["steps"] * s["weight_bytes_per_step"] for s in st)
    out = {"rows": a.rows, "ctx": a.ctx, "engines": a.engines, "steps": a.steps, "pair": a.pair,
           "ms_per_step": round(1e3 * dt / a.steps, 4), "step_frac": round(bytes_all / dt / 8e12, 4), "tokens_per_s": round(a.rows * a.engines * a.steps / dt, 1)}
    for d in decs:
        d.stamp(True); d.stats_reset()
    run_all(4)                # capture the stamped graphs
    for d in decs:
        d.stats_reset()
    run_all(a.steps)
    st = [d.stats() for d in decs]
    for d in decs:
        d.stamp(False)
    L = sum(s["stamped_launches"] for s in st); S = sum(s["stamped_seconds"] for s in st); B = sum(s["stamped_alg_bytes"] for s in st)
    if L:
        out.update(attn_us=round(1e6 * S / L, 2), attn_mb=round(B / L / 1e6, 1), attn_frac=round(B / S / 8e12, 4))
    _lib.prof_reset(); _lib.prof_enable(True)
    _lib.check(lib.etd_decoder_step(decs[0]._h, slots.ctypes.data, a.rows, 8, decs[0]._stream()), "step")
    torch.cuda.synchronize(dev)
    _lib.prof_enable(False)
    out["event_us_per_launch"] = {k: round(1e3 * v["ms"] / v["launches"], 2) for k, v in sorted(_lib.prof_report().items(), key=lambda kv: -kv[1]["ms"])}
    print(json.dumps(out), flush=True)
    for d in reversed(decs):
        d.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Batched-prefill microbenchmark: ONE prefill pass of the headline's shape (P prompts x T tokens, default 500 x 513 = 256 500 rows: what a 256 k-row
begin_bars pass of bench.py carries), repeated R times on one engine.

    python tools/bench_prefill.py [--prompts 500] [--tokens 513] [--reps 6] [--digest]

Prints ONE JSON line: wall ms per pass (back-to-back passes, one synchronisation at the end), TFLOP/s on the pass's algorithmic FLOPs (50.3 MFLOP per prompt
row + the causal attention), and the eager HIP-event breakdown per kernel of one more pass (ms per launch, TFLOP/s on the kernel's own FLOP count).
--digest adds a sha256 over the first generated token of every prompt and the K/V cache row sums (etd_debug_decoder_kv_rowsums): two builds / switches that must
be equivalent print the same values."""
import argparse
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompts", type=int, default=500)
    ap.add_argument("--tokens", type=int, default=513)
    ap.add_argument("--ragged", type=int, default=0, help="prompt lengths uniform in [tokens - ragged, tokens]")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--digest", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="diagnostic build -DETD_PMLP_STAMP: per-phase shader clocks of k_dmlp_fused's chunks 1 .. 63")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    rng = np.random.default_rng(0)
    T = (np.full(a.prompts, a.tokens, np.int32) if not a.ragged else rng.integers(a.tokens - a.ragged, a.tokens + 1, a.prompts).astype(np.int32))
    M = int(T.sum())
    dec = EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=a.prompts, max_ctx=a.tokens + 64, max_prefill_rows=M)
    st = dec._stream()
    ids = rng.integers(6, 154, M).astype(np.int32); cls = rng.integers(1, 3, M).astype(np.int32)
    a4 = np.ascontiguousarray(rng.integers(0, 3, (4, M)).astype(np.int32))
    tgt = np.ascontiguousarray(np.tile(np.asarray([2, 1, 1, 1], np.int32), (a.prompts, 1)))
    eos = np.full(a.prompts, -1, np.int32); lim = np.full(a.prompts, 48, np.int32); sl = np.arange(a.prompts, dtype=np.int32)

    def one():
        _lib.check(lib.etd_decoder_begin_bars(dec._h, a.prompts, sl.ctypes.data, T.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tgt.ctypes.data,
                                              eos.ctypes.data, lim.ctypes.data, st), "begin_bars")
    one(); one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        one()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / a.reps
    L, H, I = dcfg.num_hidden_layers, dcfg.hidden_size, dcfg.intermediate_size
    gemm = 2.0 * M * (3 * H * H + H * H + 2 * H * I) * L
    attn = float(sum(2.0 * 2.0 * 64 * (t * (t + 1) / 2) * (H // 64) for t in T.tolist())) * L
    out = {"prompts": a.prompts, "rows": M, "ms_per_pass": round(dt * 1e3, 3), "alg_tflop_per_pass": round((gemm + attn) / 1e12, 3),
           "tflops": round((gemm + attn) / dt / 1e12, 1), "frac_of_2500": round((gemm + attn) / dt / 2.5e15, 4), "rows_per_s": round(M / dt, 0)}
    _lib.prof_reset(); _lib.prof_enable(True)
    one()
    torch.cuda.synchronize(dev)
    _lib.prof_enable(False)
    prof = _lib.prof_report()
    out["kernels"] = {k: {"launches": v["launches"], "ms_per_launch": round(v["ms"] / max(1, v["launches"]), 4), "ms": round(v["ms"], 3),
                          "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1) if v.get("flops") else None}
                      for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    out["serial_ms"] = round(sum(v["ms"] for v in prof.values()), 3)
    if a.digest:
        toks = np.zeros((a.prompts, 4), np.int32); cnt = np.zeros(a.prompts, np.int32)
        _lib.check(lib.etd_decoder_read_many(dec._h, a.prompts, sl.ctypes.data, toks.ctypes.data, 4, cnt.ctypes.data, st), "read_many")
        out["first_token_sha256"] = hashlib.sha256(toks[:, 0].tobytes()).hexdigest()[:16]
        n = dcfg.num_hidden_layers * a.prompts * (a.tokens + 64)
        sums = np.zeros(n, np.uint32)
        _lib.check(lib.etd_debug_decoder_kv_rowsums(dec._h, sums.ctypes.data, n, st), "kv_rowsums")
        out["kv_rowsums_sha256"] = hashlib.sha256(sums.tobytes()).hexdigest()[:16]
    if a.stamps:
        import ctypes as C
        if not hasattr(lib, "etd_debug_pmlp_stamps"):
            sys.exit("library built without -DETD_PMLP_STAMP")
        buf = np.zeros(64 * 4 * 8, np.uint64)
        lib.etd_debug_pmlp_stamps(C.c_void_p(buf.ctypes.data))
        raw = buf.reshape(64, 4, 8).astype(np.float64)
        out["pmlp_clk_misc"] = {"chunk64": round(float(raw[:, :, 0].mean()), 0), "token_fragments_loaded": round(float(raw[:, :, 1].mean()), 0),
                                "dma_wait+barrier_per_chunk": round(float(raw[:, :, 2].mean()) / 63.0, 1)}
        out["pmlp_clk_wave_life"] = {n: round(float(raw[:, :, 3 + i].mean()), 0) for i, n in enumerate(["prologue+chunk0", "chunks1-63", "chunks64-72", "residual+hout", "whole"])}
    print(json.dumps(out))
    dec.close()


if __name__ == "__main__":
    main()

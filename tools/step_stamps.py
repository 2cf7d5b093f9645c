#!/usr/bin/env python3
"""Where does a decode-step kernel's time go, alone and with other engines running?  Per-workgroup s_memrealtime stamps
(diagnostic build:  ETD_EXTRA_FLAGS=-DETD_STEP_STAMP python -m etude_amd.build --force) of k_dstep_qkv_up, k_dstep_attn_down
(attention role and down-projection role) and k_resid_ln_rows, collected with 1 and with 4 engines stepping (54 streams each).

usage: step_stamps.py [streams=54] [ctx0=320] [steps=48]"""
import ctypes as C
import os
import sys
import threading
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig  # noqa: E402

PH = {
    (1, 0): ("qkv_up / QKV tile", ["loads land", "MFMA + LDS reduce + barrier", "epilogue (RoPE, stores issued)"]),
    (1, 1): ("qkv_up / up tile", ["loads land", "MFMA + LDS reduce + barrier", "epilogue (GELU, stores issued)"]),
    (2, 0): ("attn_down / attention", ["metadata + q + first K/V block land", "key loop", "merge + barrier", "dense slice", "slab store (+ drain + barrier with the row finish)"]),
    (2, 1): ("attn_down / down GEMM unit", ["4 x (loads + 4 MFMA)", "reduce + epilogue"]),
    (3, 0): ("resid_ln_rows (wave 0 of a workgroup)", ["slab loads land + sum + row store", "statistics", "normalise + stores issued"]),
    (4, 0): ("row finish: attention workgroup that is NOT the last arriver", ["arrive (atomic add returns)"]),
    (4, 1): ("row finish: the last arriver", ["arrive (atomic add returns)", "read 12 slabs, sum, LayerNorm, stores drained"]),
}


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    ctx0 = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    raw = C.CDLL(str(Path(_lib.__file__).with_name("libetude_hip.so")))
    if not hasattr(raw, "etd_debug_step_stamps"):
        sys.exit("library built without -DETD_STEP_STAMP")
    raw.etd_debug_step_stamps.argtypes = [C.c_void_p, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
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

    def run(dec):
        torch.cuda.set_device(0)
        _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, steps, dec._stream()), "step")
        torch.cuda.current_stream().synchronize()

    cap = 4_000_000
    buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    res = {}
    for E in (1, 4):
        reset()
        raw.etd_debug_step_stamps(C.c_void_p(buf.data_ptr()), cap, None)
        th = [threading.Thread(target=run, args=(decs[i],)) for i in range(E)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize(dev)
        n = C.c_ulonglong()
        raw.etd_debug_step_stamps(None, 0, C.byref(n))
        k = min(int(n.value), cap)
        res[E] = buf[: k * 8].cpu().numpy().reshape(k, 8).copy()
        print(f"E={E}: {n.value} records")
    out = os.environ.get("ETD_STAMP_OUT")
    if out:
        np.savez_compressed(out, e1=res[1], e4=res[4])
    for key, (name, phases) in PH.items():
        print(f"== {name}   (us; mean / median / p90 per workgroup)")
        for E in (1, 4):
            r = res[E]
            sel = r[((r[:, 0] & 0xff) == key[0]) & (((r[:, 0] >> 8) & 0xff) == key[1])]
            if not len(sel):
                continue
            t = sel[:, 1:7].astype(np.float64) * 0.01
            line = [f"E={E} n={len(sel):7d}"]
            for i, ph in enumerate(phases):
                d = t[:, i + 1] - t[:, i]
                line.append(f"{ph}: {d.mean():6.2f} / {np.median(d):6.2f} / {np.percentile(d, 90):6.2f}")
            tot = t[:, len(phases)] - t[:, 0]
            line.append(f"workgroup life: {tot.mean():6.2f} / {np.median(tot):6.2f} / {np.percentile(tot, 90):6.2f}")
            print("   " + " | ".join(line))
    # E=1: launches are sequential on one stream, so a launch = a run of records of one kernel id in t0 order
    r = res[1]
    r = r[np.argsort(r[:, 1])]
    kid = r[:, 0] & 0xff
    cuts = np.flatnonzero(np.diff(kid) != 0) + 1
    spans = {1: [], 2: [], 3: []}
    spread = {1: [], 2: [], 3: []}
    for seg in np.split(r, cuts):
        k = int(seg[0, 0] & 0xff)
        last = seg[:, 1:7].max(axis=1)
        spans[k].append((last.max() - seg[:, 1].min()) * 0.01)
        spread[k].append((seg[:, 1].max() - seg[:, 1].min()) * 0.01)
    for k, nm in ((1, "qkv_up"), (2, "attn_down"), (3, "resid_ln_rows")):
        if spans[k]:
            print(f"E=1 {nm}: first workgroup start -> last stamp {np.mean(spans[k]):.2f} us (median {np.median(spans[k]):.2f}); workgroup starts spread over {np.mean(spread[k]):.2f} us; launches {len(spans[k])}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""WHERE does a concurrent aggressor first change a decode step?  One engine replays `steps` decode steps twice from the same
prefilled state -- first alone, then beside an aggressor (the Extract stage restricted to the launchers named in ETD_EXT_ONLY, or
engine 0's batched prefills) -- with the library's step trace on (etd_debug_decoder_trace_*): a hash of every row of every
step kernel's outputs.  Prints the first records that differ: step, layer, buffer, rows (and split-K slabs).
usage: probe_trace.py [steps=300] [aggressor=extractor|prefill|none]"""
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

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    mode = sys.argv[2] if len(sys.argv) > 2 else "extractor"
    S, T0, L = 54, 340, 8
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=1088)]
    decs.append(decs[0].clone())
    slots = np.arange(S, dtype=np.int32)
    tg = np.tile(np.asarray([2, 1, 1, 1], np.int32), S)
    eos = np.full(S, -1, np.int32)

    def begin(dec, seed, limit):
        rng = np.random.default_rng(seed)
        Ts = np.asarray([T0 - (s % 9) for s in range(S)], np.int32)
        M = int(Ts.sum())
        ids, cls, a4 = rng.integers(6, 154, M).astype(np.int32), rng.integers(1, 3, M).astype(np.int32), rng.integers(0, 3, (4, M)).astype(np.int32)
        lim = np.full(S, limit, np.int32)
        _lib.check(lib.etd_decoder_begin_bars(dec._h, S, slots.ctypes.data, Ts.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tg.ctypes.data,
                                              eos.ctypes.data, lim.ctypes.data, dec._stream()), "begin_bars")
        dec._ts.synchronize()

    vic = decs[1]
    LW = 49
    wps = (LW * L + 2) * S
    stop = [False]
    ready = threading.Event()

    def aggressor():
        torch.cuda.set_device(0)
        if mode == "prefill":
            ready.set()
            i = 0
            while not stop[0]:
                begin(decs[0], 1000 + i, 60); i += 1
            return
        from etude_amd.config import ExtractorConfig
        from etude_amd.extractor import AMTAPC_Extractor
        ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(7), "cuda", max_windows=4)
        xs = torch.from_numpy(synth.window_features(5, 4)).to(dev)
        est = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(est):
            ex.transcript_windows(xs); est.synchronize()
            outs = ex._alloc(4 * ex.n_frame)
            argp = [t.data_ptr() for t in outs]
            stv = C.c_void_p(est.cuda_stream)
            ready.set()
            n = 0
            while not stop[0]:
                _lib.check(lib.etd_transcript_windows(ex._h, xs.data_ptr(), 4, *argp, None, None, None, None, stv), "etd_transcript_windows")
                est.synchronize(); n += 1
            print("(x) aggressor: %d calls of etd_transcript_windows (ETD_EXT_ONLY=%s)" % (n, os.environ.get("ETD_EXT_ONLY", "")), flush=True)

    def run(with_aggr):
        begin(vic, 101, steps + 100)
        _lib.check(lib.etd_debug_decoder_trace_begin(vic._h, steps, vic._stream()), "trace_begin")
        th = None
        if with_aggr:
            stop[0] = False; ready.clear()
            th = threading.Thread(target=aggressor); th.start(); ready.wait()
        _lib.check(lib.etd_decoder_step(vic._h, slots.ctypes.data, S, steps, vic._stream()), "step")
        vic._ts.synchronize()
        if th:
            stop[0] = True; th.join()
        torch.cuda.synchronize(dev)
        out = np.zeros(steps * wps, np.uint32); n = C.c_int()
        _lib.check(lib.etd_debug_decoder_trace_read(vic._h, out.ctypes.data, out.size, S, C.byref(n), vic._stream()), "trace_read")
        assert n.value == steps, n.value
        pk = np.zeros((12, S, 512), np.float32)
        _lib.check(lib.etd_debug_decoder_trace_slabs(vic._h, pk.ctypes.data, pk.size, S, vic._stream()), "trace_slabs")
        run.slabs = pk
        ln = np.zeros((8, S, 256, 8), np.float32)
        _lib.check(lib.etd_debug_decoder_trace_lanes(vic._h, ln.ctypes.data, ln.size, S, vic._stream()), "trace_lanes")
        run.lanes = ln
        return out.reshape(steps, wps)

    names = ["Q", "gelu(up)"] + ["slab %d (%s)" % (i, "down k-split" if i < 4 else "head %d" % (i - 4)) for i in range(12)] + ["h_out", "ln1(next)", "ln2(next)"] + ["%s head %d" % (w, h) for w in ("K[pos]", "V[pos]", "K[0..pos]", "V[0..pos]") for h in range(8)]

    def describe(step, a, b):
        d = np.nonzero(a != b)[0]
        out = []
        for off in d[:4000]:
            if off < LW * L * S:
                l, r = divmod(int(off), LW * S); k, row = divmod(r, S)
                out.append((l, k, names[k], row))
            else:
                k, row = divmod(int(off) - LW * L * S, S)
                out.append((L, 100 + k, "next h" if k == 0 else "next token", row))
        return out

    Wd = synth.decoder_state_dict(1, {})["transformer.layers.0.attention.dense.weight"]
    Wd = torch.as_tensor(np.asarray(Wd)).to(torch.bfloat16).to(torch.float64).numpy()          # [512][512], bf16-rounded like the library's copy

    hist = {}

    def explain(row, head, a_, b_, sc):
        """host model of the (row, head) attention workgroup: which slot (wave, j) of the 32 would have to lose its softmax denominator?"""
        T = T0 - (row % 9)
        q = np.zeros((S, 512), np.float32)
        _lib.check(lib.etd_debug_decoder_trace_q(vic._h, q.ctypes.data, q.size, S, vic._stream()), "trace_q")
        kk = np.zeros((T + 1, 64), np.uint16); vv = np.zeros((T + 1, 64), np.uint16)
        _lib.check(lib.etd_debug_decoder_peek_kv(vic._h, 0, row, head, T + 1, kk.ctypes.data, vv.ctypes.data, vic._stream()), "peek_kv")
        K = (kk.astype(np.uint32) << 16).view(np.float32).astype(np.float64); V = (vv.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        sco = K @ q[row, head * 64:(head + 1) * 64].astype(np.float64) * 0.125
        p = np.exp(sco - sco.max()); Ltot = p.sum(); o = (p[:, None] * V).sum(0)

        def dense(ov):
            ob = torch.as_tensor(ov).to(torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
            return Wd[:, head * 64:(head + 1) * 64] @ ob
        ref = dense(o / Ltot)
        print("      host model vs the run alone: max |diff| %.2e (|row| %.2e); largest softmax weight %.4f, newest key's weight %.4f"
              % (np.abs(ref - a_).max(), np.abs(a_).max(), p.max() / Ltot, p[T] / Ltot), flush=True)
        keys = np.arange(T + 1)
        cand = []
        for w in range(4):
            for jj in range(8):
                m_ = ((keys % 64) == w * 8 + jj) | ((keys % 64) == 32 + w * 8 + jj)
                cand.append((abs(Ltot / (Ltot - p[m_].sum()) - sc), w, jj, Ltot / (Ltot - p[m_].sum()), np.abs(dense(o / (Ltot - p[m_].sum())) - b_).max()))
        cand.sort(key=lambda t: t[4])
        exact = [(w, jj) for _, w, jj, s_, d_ in cand if d_ < 5e-8]
        for w, jj in exact:
            hist[(w, jj)] = hist.get((w, jj), 0) + 1.0 / len(exact)
        print("      observed scale %.5f; slots (wave, j) whose lost denominator reproduces the perturbed row exactly: %s; best otherwise: (%d,%d) max |diff| %.2e"
              % (sc, exact, cand[0][1], cand[0][2], cand[0][4]), flush=True)

    A = run(False); PA = run.slabs; LA = run.lanes
    A2 = run(False)
    print("alone vs alone: %d differing records" % int((A != A2).sum()), flush=True)
    for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
        B = run(mode != "none")
        bad = np.nonzero((A != B).any(axis=1))[0]
        if steps == 1:          # the values themselves: layer 0's slabs
            PB = run.slabs
            for sl in range(12):
                for row in range(S):
                    a_, b_ = PA[sl, row], PB[sl, row]
                    if not np.array_equal(a_, b_):
                        dd = np.abs(a_ - b_); i = int(dd.argmax())
                        print("   slab %d row %d: %d of 512 floats differ, max |diff| %.3e at %d (%.6e vs %.6e), |row| max %.3e, NaNs %d"
                              % (sl, row, int((a_ != b_).sum()), dd.max(), i, a_[i], b_[i], np.abs(a_).max(), int(np.isnan(b_).sum())), flush=True)
                        big = np.abs(a_) > 0.2 * np.abs(a_).max()
                        rr = b_[big] / a_[big]
                        # least-squares fit b = s * a: a pure rescaling of the row leaves no residual
                        sc = float((a_ * b_).sum() / (a_ * a_).sum())
                        print("      ratio b/a over the %d large elements: min %.5f median %.5f max %.5f; best scale %.5f leaves residual %.3e (|a| %.3e)"
                              % (int(big.sum()), rr.min(), np.median(rr), rr.max(), sc, float(np.linalg.norm(b_ - sc * a_)), float(np.linalg.norm(a_))), flush=True)
                        if sl >= 4:
                            explain(row, sl - 4, a_, b_, sc)
                            la, lb = LA[sl - 4, row], run.lanes[sl - 4, row]
                            names_ = ["lr after the key loop", "mr after the key loop", "o[0] after the key loop", "lr after stage 8", "lr after stage 16", "lr after stage 32", "o[0] after stage 8", "o[0] after stage 32"]
                            for k_ in (0, 1, 2, 3, 6, 4, 5, 7):
                                dl = np.nonzero(la[:, k_] != lb[:, k_])[0]
                                if len(dl):
                                    t0_ = int(dl[0])
                                    print("      %-24s differs in %3d lanes: %s ... first: thread %d (wave %d, j %d, c %d) %.6e alone, %.6e beside the aggressor"
                                          % (names_[k_], len(dl), dl[:12].tolist(), t0_, t0_ >> 6, (t0_ & 63) >> 3, t0_ & 7, la[t0_, k_], lb[t0_, k_]), flush=True)
        print("rep %d: %d of %d steps differ%s" % (rep, len(bad), steps, "" if not len(bad) else "; first at step %d" % bad[0]), flush=True)
        if len(bad):
            s0 = int(bad[0])
            ev = describe(s0, A[s0], B[s0])
            first = {}
            for l, k, nm, row in ev:
                first.setdefault((l, k, nm), []).append(row)
            order = {0: 0, 1: 1, **{17 + i: 2 + i for i in range(32)}, **{2 + i: 40 + i for i in range(15)}}
            for (l, k, nm), rows in sorted(first.items(), key=lambda kv: (kv[0][0], order.get(kv[0][1], 999)))[:int(os.environ.get("PROBE_LINES", "30"))]:
                print("   step %d layer %d %-22s rows %s" % (s0, l, nm, rows[:20]), flush=True)
    if hist:
        print("lost slots (wave, j) over all exactly explained events: " + ", ".join("(%d,%d): %.1f" % (w, jj, n) for (w, jj), n in sorted(hist.items())), flush=True)
        print("by j: " + ", ".join("j=%d: %.1f" % (jj, sum(n for (w, j2), n in hist.items() if j2 == jj)) for jj in range(8)), flush=True)

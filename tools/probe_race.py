#!/usr/bin/env python3
"""Who is the victim when a batched prefill runs beside other engines' decode steps?  Engine 0 runs R batched prefills (54 prompts x
~340 tokens: the big-tile path) while engines 1..3 replay decode steps; afterwards engine 0 steps ALONE from its last prefill.
Digests: (a) the stepping engines' tokens, (b) engine 0's first tokens of every prefill, (c) engine 0's tokens generated alone
from the KV cache the last concurrent prefill wrote, (d) that KV cache itself.  Run twice and compare."""
import ctypes as C
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
    import os
    NP = int(os.environ.get("PROBE_NPROMPTS", "54"))          # prompts per concurrent prefill of engine 0
    S = 54; T0 = 340; R = int(sys.argv[1]) if len(sys.argv) > 1 else 12; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    quiet = len(sys.argv) > 3 and sys.argv[3] == "quiet"          # engines 1..3 idle: the control
    hog = len(sys.argv) > 3 and sys.argv[3] == "hog"              # instead of engine 0's prefills: a generic compute + stream kernel on a fifth stream
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    decs = []
    own0 = len(sys.argv) > 3 and sys.argv[3] == "own0"            # the prefilling engine is NOT a clone: it has its own weights
    for e in range(4):
        decs.append(decs[1].clone() if (own0 and e > 1) else decs[0].clone() if (decs and not (own0 and e == 1)) else EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision=(os.environ.get("PROBE_PREC0") or os.environ.get("PROBE_PREC", "f16")) if e == 0 else os.environ.get("PROBE_PREC", "f16"), max_streams=S, max_ctx=1088))
    slots = np.arange(S, dtype=np.int32)
    tg = np.tile(np.asarray([2, 1, 1, 1], np.int32), S)
    eos = np.full(S, -1, np.int32); lim = np.full(S, 60, np.int32)

    def batch(seed):
        rng = np.random.default_rng(seed)
        Ts = np.asarray([T0 - (s % 9) for s in range(S)], np.int32)
        M = int(Ts.sum())
        return Ts, rng.integers(6, 154, M).astype(np.int32), rng.integers(1, 3, M).astype(np.int32), rng.integers(0, 3, (4, M)).astype(np.int32)

    def begin(dec, seed):
        Ts, ids, cls, a4 = batch(seed)
        if NP < S:
            M = int(Ts[:NP].sum())
            ids, cls, a4 = ids[:M].copy(), cls[:M].copy(), np.ascontiguousarray(a4[:, :M])
        _lib.check(lib.etd_decoder_begin_bars(dec._h, min(NP, S), slots.ctypes.data, Ts.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tg.ctypes.data,
                                              eos.ctypes.data, lim.ctypes.data, dec._stream()), "begin_bars")

    def tokens(dec, cap=1100):
        h = hashlib.sha256()
        for s in range(S):
            buf = np.zeros(cap, np.int32); n = C.c_int()
            _lib.check(lib.etd_decoder_read_tokens(dec._h, s, buf.ctypes.data, cap, C.byref(n), dec._stream()), "read")
            h.update(buf[: n.value].tobytes())
        return h.hexdigest()[:16]

    lim_step = np.full(S, 700, np.int32)
    for e in (1, 2, 3):                        # the stepping engines start from a sequentially prefetched state
        Ts, ids, cls, a4 = batch(100 + e)
        _lib.check(lib.etd_decoder_begin_bars(decs[e]._h, S, slots.ctypes.data, Ts.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tg.ctypes.data,
                                              eos.ctypes.data, lim_step.ctypes.data, decs[e]._stream()), "begin_bars")
        decs[e]._ts.synchronize()
    torch.cuda.synchronize(dev)
    firsts = hashlib.sha256()

    if os.environ.get("PROBE_PREFILL_NEW_STREAM"):
        decs[0]._ts = torch.cuda.Stream(device=dev)

    def prefiller():
        torch.cuda.set_device(0)
        if os.environ.get("PROBE_LOGITS_T"):          # a reduced aggressor: R single-prompt forwards of T tokens with logits for every row
            T = int(os.environ["PROBE_LOGITS_T"])
            rng = np.random.default_rng(5)
            ids = rng.integers(6, 154, T).astype(np.int32); cls = rng.integers(1, 3, T).astype(np.int32); a4 = rng.integers(0, 3, (4, T)).astype(np.int32)
            for i in range(R):
                decs[0].prefill_logits(ids, cls, a4)
            return
        for i in range(R):
            begin(decs[0], 1000 + i)
            decs[0]._ts.synchronize()
            firsts.update(tokens(decs[0]).encode())

    def stepper(e):
        torch.cuda.set_device(0)
        _lib.check(lib.etd_decoder_step(decs[e]._h, slots.ctypes.data, S, steps, decs[e]._stream()), "step")
        decs[e]._ts.synchronize()

    stop = [False]

    def hogger():
        torch.cuda.set_device(0)
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.hog_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]
        src = torch.ones(64 << 20, dtype=torch.float32, device=dev)
        sink = torch.zeros(16, dtype=torch.float32, device=dev)
        hst = decs[0]._ts if os.environ.get("PROBE_HOG_ON_ENGINE0") else torch.cuda.Stream(device=dev)
        while not stop[0]:
            for _ in range(16):
                nl.hog_launch(2048, src.data_ptr(), sink.data_ptr(), 64, src.numel(), hst.cuda_stream)
            hst.synchronize()

    gemm = len(sys.argv) > 3 and sys.argv[3] == "gemm"             # ... or the library's own GEMM microbenchmark (k_linear on private buffers)

    def gemmer():
        torch.cuda.set_device(0)
        gst = torch.cuda.Stream(device=dev)
        us = C.c_double()
        lib.etd_debug_linear.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        which = os.environ.get("PROBE_KERNEL")
        while not stop[0]:
            if which is not None:
                _lib.check(lib.etd_debug_kernel_loop(int(which), 40, gst.cuda_stream), "debug_kernel_loop")
            else:
                _lib.check(lib.etd_debug_linear(18432, 2048, 512, 60, gst.cuda_stream, C.byref(us)), "debug_linear")

    hogw = len(sys.argv) > 3 and sys.argv[3] == "hogw"            # ... or a kernel that streams WRITES over 1 GiB

    def hogwriter():
        torch.cuda.set_device(0)
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.hogw_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p]
        dst = torch.empty(256 << 20, dtype=torch.float32, device=dev)
        hst = torch.cuda.Stream(device=dev)
        k = 0
        while not stop[0]:
            for _ in range(8):
                k += 1
                nl.hogw_launch(4096, dst.data_ptr(), 64, dst.numel(), float(k), hst.cuda_stream)
            hst.synchronize()

    regn = len(sys.argv) > 3 and sys.argv[3] == "regnoise"        # ... or waves that leave NaNs in every VGPR and AGPR of every SIMD

    def regnoiser():
        torch.cuda.set_device(0)
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.regnoise_launch.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        sink = torch.zeros(16, dtype=torch.int32, device=dev)
        hst = torch.cuda.Stream(device=dev)
        while not stop[0]:
            for _ in range(32):
                nl.regnoise_launch(4096, 0x7fc00000, sink.data_ptr(), hst.cuda_stream)
            hst.synchronize()

    extr = len(sys.argv) > 3 and sys.argv[3] == "extractor"       # ... or the Extract stage's model (k_attn, k_enc_layer, k_proj256, ...)

    def extractor_loop():
        torch.cuda.set_device(0)
        t_c0 = __import__("time").perf_counter()
        from etude_amd.config import ExtractorConfig
        from etude_amd.extractor import AMTAPC_Extractor
        ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(7), "cuda", max_windows=4)
        xs = torch.from_numpy(synth.window_features(5, 4)).to(dev)
        est = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(est):
            ex.transcript_windows(xs)
            est.synchronize()
            print("(x) extractor engine ready after %.1f s" % (__import__("time").perf_counter() - t_c0), flush=True)
            ext_ready.set()                      # the engine exists, its first pass is done: only the steady-state loop runs beside the steppers
            if os.environ.get("PROBE_EXT_IDLE") == "2":     # the extractor exists; what is launched is the prof.hip empty kernel (harmless on its own)
                sink2 = torch.zeros(1024, dtype=torch.int32, device=dev)
                nl_ = 0
                while not stop[0]:
                    _lib.check(lib.etd_debug_empty_launch(*[int(v) for v in os.environ.get("PROBE_EXT_IDLE_GRID", "8,8,4").split(",")], sink2.data_ptr(), C.c_void_p(est.cuda_stream)), "empty_launch")
                    est.synchronize(); nl_ += 1
                print("(x) aggressor: %d empty launches through etd_debug_empty_launch" % nl_, flush=True)
            if os.environ.get("PROBE_EXT_IDLE"):    # the extractor exists and has run once; nothing is launched while the steppers run
                while not stop[0]:
                    __import__("time").sleep(0.01)
            if os.environ.get("PROBE_EXT_CAPI"):    # the C entry point directly, outputs allocated once: no Python wrapper, no torch allocator
                outs = ex._alloc(4 * ex.n_frame)
                argp = [t.data_ptr() for t in outs]
                stv = C.c_void_p(est.cuda_stream)
                nl_ = 0
                while not stop[0]:
                    _lib.check(lib.etd_transcript_windows(ex._h, xs.data_ptr(), 4, *argp, None, None, None, None, stv), "etd_transcript_windows")
                    if not os.environ.get("PROBE_EXT_NOSYNC"):
                        est.synchronize()
                    nl_ += 1
                est.synchronize()
                print("(x) aggressor: %d calls of etd_transcript_windows" % nl_, flush=True)
            while not stop[0]:
                ex.transcript_windows(xs)
                est.synchronize()

    ext_ready = threading.Event()
    if extr:
        ht = threading.Thread(target=extractor_loop); ht.start()
        if os.environ.get("PROBE_EXT_EARLY"):       # (the first version of this probe: steppers start 3 s after the extractor's creation BEGAN)
            __import__("time").sleep(3.0)
        else:
            ext_ready.wait()
        ths = [threading.Thread(target=stepper, args=(e,)) for e in (1, 2, 3)]
        for x in ths:
            x.start()
        for x in ths:
            x.join()
        stop[0] = True; ht.join()
        torch.cuda.synchronize(dev)
        print("(a) stepping engines' tokens:", " ".join(tokens(decs[e]) for e in (1, 2, 3)))
        sys.exit(0)
    burn = len(sys.argv) > 3 and sys.argv[3] == "burn"            # ... or register-only MFMA loops on every CU (power, no memory)

    def burner():
        torch.cuda.set_device(0)
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.burn_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p]
        sink = torch.zeros(16, dtype=torch.float32, device=dev)
        hst = torch.cuda.Stream(device=dev)
        k = 0
        while not stop[0]:
            for _ in range(8):
                k += 1
                nl.burn_launch(512, 2000, sink.data_ptr(), k, hst.cuda_stream)      # 2 workgroups per CU x 8 waves, ~0.5 ms each
            hst.synchronize()

    ldsn = len(sys.argv) > 3 and sys.argv[3].startswith("lds")    # lds<KiB>: workgroups that only fill and read back <KiB> of dynamic LDS

    def ldsnoiser():
        torch.cuda.set_device(0)
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.noise_launch2.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        kib = int(sys.argv[3][3:])
        sink = torch.zeros(16, dtype=torch.int32, device=dev)
        hst = torch.cuda.Stream(device=dev)
        while not stop[0]:
            for _ in range(16):
                nl.noise_launch2(1024, kib * 1024, 0x3f800000, sink.data_ptr(), hst.cuda_stream)
            hst.synchronize()

    sc8 = len(sys.argv) > 3 and sys.argv[3] == "scatter8"         # ... or k_embed's store pattern alone: 8-byte pieces into rows 512 bytes apart

    def scatterer():
        torch.cuda.set_device(0)
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.scatter8_launch.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_ulonglong, C.c_void_p]
        n_rows = 1 << 20                                            # 512 MiB of 512-byte rows
        dst = torch.empty(n_rows * 64, dtype=torch.int64, device=dev)
        hst = torch.cuda.Stream(device=dev)
        k = 0
        while not stop[0]:
            for _ in range(8):
                k += 1
                nl.scatter8_launch(2048, dst.data_ptr(), n_rows, 64, k, hst.cuda_stream)
            hst.synchronize()

    mset = len(sys.argv) > 3 and sys.argv[3] in ("memset", "h2d", "d2h", "d2d")   # ... or nothing but runtime fill / copy operations on a stream of their own

    def copier():
        torch.cuda.set_device(0)
        hst = torch.cuda.Stream(device=dev)
        nbytes = int(os.environ.get("PROBE_COPY_BYTES", str(1 << 20)))
        dbuf = torch.empty(nbytes, dtype=torch.uint8, device=dev); dbuf2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        hbuf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        mode = sys.argv[3]
        with torch.cuda.stream(hst):
            while not stop[0]:
                for _ in range(32):
                    if mode == "memset":
                        dbuf.zero_()
                    elif mode == "h2d":
                        dbuf.copy_(hbuf, non_blocking=True)
                    elif mode == "d2h":
                        hbuf.copy_(dbuf, non_blocking=True)
                    else:
                        dbuf2.copy_(dbuf, non_blocking=True)
                hst.synchronize()

    evt = len(sys.argv) > 3 and sys.argv[3] in ("events", "alloc", "emptykernel")    # ... or only event records / only allocator traffic / only empty launches

    def eventer():
        torch.cuda.set_device(0)
        hst = torch.cuda.Stream(device=dev)
        mode = sys.argv[3]
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.noise_launch2.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        sink = torch.zeros(16, dtype=torch.int32, device=dev)
        with torch.cuda.stream(hst):
            while not stop[0]:
                if mode == "events":
                    for _ in range(64):
                        e_ = torch.cuda.Event(); e_.record(hst)
                    hst.synchronize()
                elif mode == "alloc":
                    for _ in range(16):
                        ts = [torch.empty((2048, 88), dtype=torch.float32, device=dev) for _ in range(4)]
                        del ts
                    hst.synchronize()
                elif os.environ.get("PROBE_EMPTY_INLIB"):    # the same empty kernel, but compiled into and launched from libetude_hip.so
                    gx_, gy_, gz_ = [int(x) for x in os.environ["PROBE_EMPTY_INLIB"].split(",")]
                    _lib.check(lib.etd_debug_empty_launch(gx_, gy_, gz_, sink.data_ptr(), C.c_void_p(hst.cuda_stream)), "empty_launch")
                    hst.synchronize()
                elif os.environ.get("PROBE_EMPTY"):      # "which,gx,gy,gz": an empty kernel (0 plain, 1 82 KiB static LDS, 2 296 registers) on that grid
                    w_, gx_, gy_, gz_ = [int(x) for x in os.environ["PROBE_EMPTY"].split(",")]
                    nl.empty_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
                    nl.empty_launch(w_, gx_, gy_, gz_, sink.data_ptr(), hst.cuda_stream)
                    hst.synchronize()
                else:                       # one small launch, then a stream synchronize -- the cadence of the k_embed-only aggressor
                    nl.noise_launch2(256, 82 * 1024, 0x3f800000, sink.data_ptr(), hst.cuda_stream)
                    hst.synchronize()

    if evt:
        ht = threading.Thread(target=eventer); ht.start()
        ths = [threading.Thread(target=stepper, args=(e,)) for e in (1, 2, 3)]
        for x in ths:
            x.start()
        for x in ths:
            x.join()
        stop[0] = True; ht.join()
        torch.cuda.synchronize(dev)
        print("(a) stepping engines' tokens:", " ".join(tokens(decs[e]) for e in (1, 2, 3)))
        sys.exit(0)
    if mset:
        ht = threading.Thread(target=copier); ht.start()
        ths = [threading.Thread(target=stepper, args=(e,)) for e in (1, 2, 3)]
        for x in ths:
            x.start()
        for x in ths:
            x.join()
        stop[0] = True; ht.join()
        torch.cuda.synchronize(dev)
        print("(a) stepping engines' tokens:", " ".join(tokens(decs[e]) for e in (1, 2, 3)))
        sys.exit(0)
    if hog or gemm or hogw or regn or burn or ldsn or sc8:
        ht = threading.Thread(target=gemmer if gemm else hogwriter if hogw else regnoiser if regn else burner if burn else ldsnoiser if ldsn else scatterer if sc8 else hogger); ht.start()
        ths = [threading.Thread(target=stepper, args=(e,)) for e in (1, 2, 3)]
        for x in ths:
            x.start()
        for x in ths:
            x.join()
        stop[0] = True; ht.join()
        torch.cuda.synchronize(dev)
        print("(a) stepping engines' tokens:", " ".join(tokens(decs[e]) for e in (1, 2, 3)))
        if os.environ.get("PROBE_KVSUMS"):
            rs = np.zeros(8 * S * 1088, np.uint32)
            _lib.check(lib.etd_debug_decoder_kv_rowsums(decs[2]._h, rs.ctypes.data, rs.size, decs[2]._stream()), "kv_rowsums")
            np.save(os.environ["PROBE_KVSUMS"], rs.reshape(8, S, 1088))
        for e in (1, 2, 3):
            v = (C.c_ulonglong * 48)()
            _lib.check(lib.etd_debug_decoder_checksum(decs[e]._h, v, 48, decs[e]._stream()), "checksum")
            print("(m) engine", e, "per allocation:", " ".join("%x" % (x & 0xffffffff) for x in list(v)[:40]))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "oob":                # engines 1..3 idle: does a prefill of engine 0 change a single word of THEIR memory?
        def csum(dec):
            v = (C.c_ulonglong * 1)()
            _lib.check(lib.etd_debug_decoder_checksum(dec._h, v, 1, dec._stream()), "checksum")
            return v[0]
        before = [csum(decs[e]) for e in (1, 2, 3)]
        again = [csum(decs[e]) for e in (1, 2, 3)]
        prefiller()
        torch.cuda.synchronize(dev)
        after = [csum(decs[e]) for e in (1, 2, 3)]
        print("(o) checksums of engines 1..3 before:", before, "repeat:", again, "after", R, "prefills of engine 0:", after, "->", "UNCHANGED" if before == after == again else "CHANGED")
        sys.exit(0)
    role = os.environ.get("PROBE_ROLE", "")                      # two PROCESSES: one runs only the prefills, the other only the steps
    if role == "prefiller":
        prefiller()
        print("(p) prefiller process done")
        sys.exit(0)
    th = ([] if role == "stepper" else [threading.Thread(target=prefiller)]) + ([] if quiet else [threading.Thread(target=stepper, args=(e,)) for e in (1, 2, 3)])
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize(dev)
    print("(a) stepping engines' tokens:", " ".join(tokens(decs[e]) for e in (1, 2, 3)))
    def csum2(dec):
        v = (C.c_ulonglong * 48)()
        _lib.check(lib.etd_debug_decoder_checksum(dec._h, v, 48, dec._stream()), "checksum")
        return ["%x" % (x & 0xffffffff) for x in v]
    for e in (1, 2, 3):
        print("(m) engine", e, "per allocation:", " ".join(csum2(decs[e])[:40]))
    kvd = os.environ.get("PROBE_KVSUMS")
    if kvd:                                    # per (layer, slot, position) sums of engine 2's K/V cache -> npy
        rs = np.zeros(8 * S * 1088, np.uint32)
        _lib.check(lib.etd_debug_decoder_kv_rowsums(decs[2]._h, rs.ctypes.data, rs.size, decs[2]._stream()), "kv_rowsums")
        np.save(kvd, rs.reshape(8, S, 1088))
    dump = os.environ.get("PROBE_DUMP")
    if dump:                                   # all token streams of the stepping engines -> npy (compare two runs offline)
        allt = np.zeros((3, S, 1100), np.int32)
        for ei, e in enumerate((1, 2, 3)):
            for s_ in range(S):
                buf = np.zeros(1100, np.int32); n = C.c_int()
                _lib.check(lib.etd_decoder_read_tokens(decs[e]._h, s_, buf.ctypes.data, 1100, C.byref(n), decs[e]._stream()), "read")
                allt[ei, s_, : n.value] = buf[: n.value]
        np.save(dump, allt)
    print("(b) engine 0, first tokens of its", R, "concurrent prefills:", firsts.hexdigest()[:16])
    d0 = decs[0]
    # (d) KV cache of engine 0 as the last concurrent prefill left it -- through the debug accessor if the library has one, else skipped
    _lib.check(lib.etd_decoder_step(d0._h, slots.ctypes.data, S, 48, d0._stream()), "step")
    d0._ts.synchronize()
    print("(c) engine 0 stepping alone from its last concurrent prefill:", tokens(d0))

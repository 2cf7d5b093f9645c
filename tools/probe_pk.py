#!/usr/bin/env python3
"""Does packed-FP32 arithmetic stay correct beside another stream's MFMA kernels?  A self-checking kernel
(tools/ubench/pk_check.hip; build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC pk_check.hip -o libpk_check.so) runs
alone, then beside the Extract stage restricted to ETD_EXT_ONLY's launchers.  usage: probe_pk.py [seconds=3] [n_wg=432] [iters=40]; PROBE_PK_ASYNC=1|2|3 runs ubench/pk_async_check.hip instead
(crossed packed ops while ds_bpermute (1) / global_load (2) / both (3) results of the same wave are in flight)"""
import ctypes as C
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib, synth  # noqa: E402

if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    n_wg = int(sys.argv[2]) if len(sys.argv) > 2 else 432
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    amode = int(os.environ.get("PROBE_PK_ASYNC", "0"))
    xl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / ("libpk_async_check.so" if amode else "libpk_check.so")))
    if not amode:
        xl.pk_check.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if amode:
        xl.pk_async_fill.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]
        xl.pk_async_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        gsrc = torch.empty(64 << 20, dtype=torch.int32, device=dev)
    errs = torch.zeros(8, dtype=torch.int64, device=dev)
    lanes = torch.zeros(64, dtype=torch.int64, device=dev)
    first = torch.zeros(8, dtype=torch.int32, device=dev)
    vst = torch.cuda.Stream(device=dev)
    if amode:
        assert xl.pk_async_fill(gsrc.data_ptr(), gsrc.numel(), vst.cuda_stream) == 0
        vst.synchronize()
    stop = [False]
    ready = threading.Event()

    def aggressor():
        torch.cuda.set_device(0)
        from etude_amd.config import ExtractorConfig
        from etude_amd.extractor import AMTAPC_Extractor
        ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(7), "cuda", max_windows=4)
        xs = torch.from_numpy(synth.window_features(5, 4)).to(dev)
        est = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(est):
            ex.transcript_windows(xs); est.synchronize()
            outs = ex._alloc(4 * ex.n_frame)
            argp = [t.data_ptr() for t in outs]
            ready.set()
            n = 0
            while not stop[0]:
                _lib.check(lib.etd_transcript_windows(ex._h, xs.data_ptr(), 4, *argp, None, None, None, None, C.c_void_p(est.cuda_stream)), "etd_transcript_windows")
                est.synchronize(); n += 1
            print("(x) aggressor: %d calls of etd_transcript_windows (ETD_EXT_ONLY=%s)" % (n, os.environ.get("ETD_EXT_ONLY", "")), flush=True)

    def victim(label):
        errs.zero_(); first.zero_(); lanes.zero_(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(20):
                if amode:
                    assert xl.pk_async_check(n_wg, iters, amode, gsrc.data_ptr(), gsrc.numel() // 4, errs.data_ptr(), lanes.data_ptr(), first.data_ptr(), vst.cuda_stream) == 0
                else:
                    assert xl.pk_check(n_wg, iters, errs.data_ptr(), lanes.data_ptr(), first.data_ptr(), vst.cuda_stream) == 0
            vst.synchronize(); n += 20
        e = errs.cpu().numpy(); f = first.cpu().numpy().astype(np.uint32); ln = lanes.cpu().numpy()
        if amode:
            print("%s: %d launches of %d workgroups x %d iterations (async mode %d); wrong results: crossed packed sum %d, exchanged values %d, loaded values %d"
                  % (label, n, n_wg, iters, amode, e[0], e[1], e[2]), flush=True)
        else:
            print("%s: %d launches of %d workgroups x %d iterations x 8; wrong results: v_pk_mul_f32 %d, v_pk_fma_f32 %d, v_pk_fma_f32 with crossed op_sel %d"
                  % (label, n, n_wg, iters, e[0], e[1], e[2]), flush=True)
        if ln.sum():
            print("   wrong results by lane group of 8: %s" % [int(ln[8 * g:8 * g + 8].sum()) for g in range(8)], flush=True)
        if f[7]:
            fl = lambda u: float(np.asarray([u], np.uint32).view(np.float32)[0])
            print("   first crossed-op_sel error: workgroup %d thread %d iteration %d: lo %.9g (expected %.9g), hi %.9g (expected %.9g)" % (f[0], f[1], f[2], fl(f[3]), fl(f[4]), fl(f[5]), fl(f[6])), flush=True)

    victim("alone")
    th = threading.Thread(target=aggressor); th.start(); ready.wait()
    victim("beside the aggressor")
    victim("beside the aggressor (again)")
    stop[0] = True; th.join()
    victim("alone again")

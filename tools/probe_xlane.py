#!/usr/bin/env python3
"""Do cross-lane exchanges and small LDS hand-offs stay correct beside another stream's kernels?  A self-checking kernel
(tools/ubench/xlane_check.hip; build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC xlane_check.hip -o libxlane_check.so) runs
alone, then beside the Extract stage restricted to ETD_EXT_ONLY's launchers.  usage: probe_xlane.py [seconds=3] [mode=15] [n_wg=432] [iters=40]"""
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
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    n_wg = int(sys.argv[3]) if len(sys.argv) > 3 else 432
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = _lib.lib()
    xl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "libxlane_check.so"))
    xl.xlane_fill.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]
    xl.xlane_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    gw = 64 << 20
    gsrc = torch.empty(gw, dtype=torch.int32, device=dev)
    errs = torch.zeros(8, dtype=torch.int64, device=dev)
    first = torch.zeros(8, dtype=torch.int32, device=dev)
    vst = torch.cuda.Stream(device=dev)
    assert xl.xlane_fill(gsrc.data_ptr(), gw, vst.cuda_stream) == 0
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
        errs.zero_(); first.zero_(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(20):
                assert xl.xlane_check(n_wg, iters, mode, gsrc.data_ptr(), gw, errs.data_ptr(), first.data_ptr(), vst.cuda_stream) == 0
            vst.synchronize(); n += 20
        e = errs.cpu().numpy(); f = first.cpu().numpy().astype(np.uint32)
        print("%s: %d launches of %d workgroups x %d iterations; wrong results: ds_bpermute %d, dpp/permlane %d, LDS hand-off %d, global loads %d"
              % (label, n, n_wg, iters, e[0], e[1], e[2], e[3]), flush=True)
        if f[7]:
            print("   first ds_bpermute error: workgroup %d thread %d iteration %d offset %d value %d: got %08x, expected %08x" % (f[0], f[1], f[2], f[3], f[6], f[4], f[5]), flush=True)

    victim("alone")
    th = threading.Thread(target=aggressor); th.start(); ready.wait()
    victim("beside the aggressor")
    victim("beside the aggressor (again)")
    stop[0] = True; th.join()
    victim("alone again")

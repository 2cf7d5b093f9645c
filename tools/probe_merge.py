#!/usr/bin/env python3
"""Does hipcc's own code for the softmax merge (crossed packed-FP32 sums) agree with the scalar form beside another stream's MFMA kernels?  A self-checking kernel
(tools/ubench/merge_check.hip; build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC merge_check.hip -o libmerge_check.so) runs
alone, then beside the Extract stage restricted to ETD_EXT_ONLY's launchers.  usage: probe_merge.py [seconds=3] [n_wg=432] [iters=40]"""
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
    xl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "libmerge_check.so"))
    xl.merge_check.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    gsrc = torch.ones(16 << 20, dtype=torch.float32, device=dev)
    errs = torch.zeros(8, dtype=torch.int64, device=dev)
    lanes = torch.zeros(64, dtype=torch.int64, device=dev)
    first = torch.zeros(8, dtype=torch.int32, device=dev)
    vst = torch.cuda.Stream(device=dev)
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
                assert xl.merge_check(n_wg, iters, gsrc.data_ptr(), gsrc.numel(), errs.data_ptr(), lanes.data_ptr(), first.data_ptr(), vst.cuda_stream) == 0
            vst.synchronize(); n += 20
        e = errs.cpu().numpy(); f = first.cpu().numpy().astype(np.uint32); ln = lanes.cpu().numpy()
        print("%s: %d launches of %d workgroups x %d merges per lane; lanes where the compiled merge and the scalar merge disagree: lr %d, o[] %d"
              % (label, n, n_wg, iters, e[0], e[1]), flush=True)
        if ln.sum():
            print("   disagreements by lane group of 8: %s" % [int(ln[8 * g:8 * g + 8].sum()) for g in range(8)], flush=True)
        if f[7]:
            fl = lambda u: float(np.asarray([u], np.uint32).view(np.float32)[0])
            print("   first lr disagreement: workgroup %d thread %d iteration %d: compiled %.9g, scalar %.9g" % (f[0], f[1], f[2], fl(f[3]), fl(f[4])), flush=True)

    victim("alone")
    th = threading.Thread(target=aggressor); th.start(); ready.wait()
    victim("beside the aggressor")
    victim("beside the aggressor (again)")
    stop[0] = True; th.join()
    victim("alone again")

#!/usr/bin/env python3
"""Which HIP streams share a compute pipe?  Two dependent empty-kernel chains that sit on one pipe run at half speed each
(one queue per pipe is serviced at a time); on different pipes they do not disturb each other.  Runs every pair of N streams."""
import ctypes as C
import os
import sys
import threading
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib  # noqa: E402

torch.cuda.set_device(0)
lib = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.Stream() for _ in range(N)]


def run(i, out, k):
    torch.cuda.set_device(0)
    e, g = C.c_double(), C.c_double()
    _lib.check(lib.etd_debug_boundary_cost(64, 100, 0, C.c_void_p(streams[i].cuda_stream), C.byref(e), C.byref(g)), "boundary_cost")
    out[k] = g.value


solo = [None]
run(0, solo, 0)
print(f"solo: {solo[0]:.2f} us per boundary")
for i in range(N):
    row = []
    for j in range(N):
        if j <= i:
            row.append("  .  ")
            continue
        out = [None, None]
        th = [threading.Thread(target=run, args=(i, out, 0)), threading.Thread(target=run, args=(j, out, 1))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        row.append(f"{max(out):5.2f}")
    print(f"stream {i}: " + " ".join(row))

#!/usr/bin/env python3
"""Do kernel boundaries of independent queues serialise?  E host threads, each replaying a hipGraph of dependent EMPTY kernels
on its own stream (etd_debug_boundary_cost); prints the per-boundary time each chain sees and the aggregate kernel rate."""
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
streams = [torch.cuda.Stream() for _ in range(6)]
nodes, iters = 64, 200


def run(i, out):
    torch.cuda.set_device(0)
    e, g = C.c_double(), C.c_double()
    _lib.check(lib.etd_debug_boundary_cost(nodes, iters, 0, C.c_void_p(streams[i].cuda_stream), C.byref(e), C.byref(g)), "boundary_cost")
    out[i] = (e.value, g.value)


for E in (1, 2, 3, 4, 6):
    out = [None] * E
    th = [threading.Thread(target=run, args=(i, out)) for i in range(E)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    g = [o[1] for o in out]
    print(f"E={E}: graph replay {min(g):.2f}-{max(g):.2f} us per boundary on each chain -> {sum(1.0 / x for x in g) * 1e3:.0f} k kernels/s aggregate   (eager {out[0][0]:.2f} us)")

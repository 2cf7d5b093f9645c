#!/usr/bin/env python3
"""GPU-box micro-measurements that decide design questions (kernel-boundary cost, ...)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib  # noqa: E402

lib = _lib.lib()
st = torch.cuda.Stream()
for big in (0, 1):
    for n in (8, 52):
        e, g = C.c_double(), C.c_double()
        _lib.check(lib.etd_debug_boundary_cost(n, 200, big, C.c_void_p(st.cuda_stream), C.byref(e), C.byref(g)), "boundary")
        print(f"empty kernel x{n} big_args={big}: eager {e.value:.2f} us/kernel, graph {g.value:.2f} us/kernel")

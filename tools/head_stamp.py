"""Phase timing inside k_dstep_head (the decode step's last kernel) from s_memtime stamps.

Needs a diagnostic build:  ETD_EXTRA_FLAGS=-DETD_HEAD_STAMP python -m etude_amd.build --force   (each stamp costs ~900 clk,
so rebuild without the flag afterwards).  Prints, for workgroups 0 and 1, the shader-clock count of each phase:
row/LN loads + first reduction, rest of the LayerNorm, barrier wait, logits MFMAs, token choice + state update, next embedding.
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
import bench
from etude_amd import synth
from etude_amd.decoder import EtudeDecoderConfig

dev = torch.device("cuda:0"); torch.cuda.set_device(0)
r = bench.decoder_stream_bench(EtudeDecoderConfig(**synth.decoder_dims()), dev, n_streams=54, ctx0=320, steps=32)
print(r["ms_per_step"], r["kernel_ms_per_step"])
lib = C.CDLL('/root/repo/etude_amd/libetude_hip.so')
if not hasattr(lib, "etd_debug_head_stamps"):
    sys.exit("library built without -DETD_HEAD_STAMP")
buf = np.zeros(8 * 16, np.int64)
lib.etd_debug_head_stamps(C.c_void_p(buf.ctypes.data))
b = buf.reshape(8, 16)
for g in range(2):
    t = b[g]
    print("wg", g, "clk: loads+sum", t[6] - t[0], "LN rest", t[7] - t[6], "barrier", t[1] - t[7], "logits", t[2] - t[1], "choice+state", t[3] - t[2],
          "next embedding", t[4] - t[3], "total", t[4] - t[0])

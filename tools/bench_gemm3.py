#!/usr/bin/env python3
"""k_gemm3 / k_gemm3_s on the exact-parity mode's GEMM shapes through the test hook (weights re-packed per call: timed with device events around the launch only is not
possible through the hook, so each shape runs R calls and the first call's upload is amortised; the hook synchronises -- use for relative comparisons).

    python tools/bench_gemm3.py            # prints ms and fp32-equivalent TFLOP/s per shape"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from etude_amd import _lib  # noqa: E402


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    shapes = [("dec qkv", 131072, 1536, 512), ("dec dense", 131072, 512, 512), ("dec up", 131072, 2048, 512), ("dec down", 131072, 512, 2048),
              ("ext qkv", 131072, 768, 256), ("ext o", 131072, 256, 256), ("ext f1", 131072, 512, 256), ("ext f2", 131072, 256, 512),
              ("step qkv", 1728, 1536, 512), ("step dense", 1728, 512, 512), ("step up", 1728, 2048, 512), ("step down", 1728, 512, 2048),
              ("step qkv", 576, 1536, 512), ("step dense", 576, 512, 512), ("step up", 576, 2048, 512), ("step down", 576, 512, 2048),
              ("small qkv", 54, 1536, 512), ("small down", 54, 512, 2048)]
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        shapes = [s_ for s_ in shapes if s_[0].startswith("step")]
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        shapes = [s_ for s_ in shapes if s_[0].startswith("dec")]
    rng = np.random.default_rng(0)
    for name, M, N, K in shapes:
        x = torch.randn((M, K), dtype=torch.float32, device=dev)
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32); b = np.zeros(N, np.float32)
        call = lambda: _lib.check(lib.etd_debug_gemm3(x.data_ptr(), M, K, w.ctypes.data, b.ctypes.data, N, 8.0, 0, y.data_ptr(), None, None, st), "gemm3")   # noqa: E731
        call()
        # the hook uploads the packed planes (host work + H2D) and synchronises: time the device part with the library's event profiler instead
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5):
            call()
        _lib.prof_enable(False)
        rep = _lib.prof_report()
        k = [v for kk, v in rep.items() if kk.startswith("k_gemm3")]
        ms = sum(v["ms"] for v in k) / max(1, sum(v["launches"] for v in k))
        print(f"{name:12s} M {M:7d} N {N:5d} K {K:5d}: {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s fp32-equivalent ({3 * 2.0 * M * N * K / ms / 1e9 / 2500:5.3f} of the f16 MFMA peak), "
              f"{(M * K * 4 + M * N * 4) / ms / 1e6:7.1f} GB/s of fp32 rows")


if __name__ == "__main__":
    main()

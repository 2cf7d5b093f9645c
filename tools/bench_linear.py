"""GEMM microbenchmark: k_linear (128-token x 256-feature tiles, plain bf16 epilogue) on the shapes the extractor and the
decoder prefill use.  `ETD_LIN_STAMP=1 python tools/bench_linear.py M N K` prints the in-kernel phase stamps of one launch."""
import ctypes as C
import sys

import torch

sys.path.insert(0, '/root/repo')
from etude_amd import _lib

torch.cuda.set_device(0)
st = torch.cuda.Stream()
lib = _lib.lib()
shapes = [tuple(int(x) for x in sys.argv[1:4])] if len(sys.argv) >= 4 else [
    (45056, 768, 256), (45056, 512, 256), (45056, 256, 512),                 # extractor, 16 windows: qkv / ffn1 / ffn2
    (17800, 1536, 512), (17800, 2048, 512), (17800, 512, 2560),              # decoder prefill, 54 prompts x ~330 tokens: qkv / up / [down|dense]
    (4096, 512, 2560), (65536, 512, 2560)]
for (M, N, K) in shapes:
    us = C.c_double()
    _lib.check(lib.etd_debug_linear(M, N, K, 20, C.c_void_p(st.cuda_stream), C.byref(us)), "etd_debug_linear")
    tf = 2.0 * M * N * K / us.value * 1e-6
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us.value:8.1f} us  {tf:7.1f} TFLOP/s  ({tf / 2500 * 100:.1f} % of bf16 MFMA peak)")

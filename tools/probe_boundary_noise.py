"""Does kernel-boundary traffic of OTHER queues (cache invalidates / write-backs at every boundary) slow a decode chain?  One engine\ndecodes while 0, 1 or 3 host threads replay graphs of empty dependent kernels on their own streams.  Measured: 0.205 -> 0.215 ->\n0.234 ms per step with ~0.5 M boundaries/s per noise chain -- real engines issue ~60 k/s each, so this is not what makes four\nengines slow each other down."""
import ctypes as C, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from etude_amd import _lib, synth
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
lib = _lib.lib()
S, ctx0, steps = 54, 320, 96
dec = EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()), synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=S, max_ctx=1024)
rng = np.random.default_rng(0); tg = np.asarray([2, 1, 1, 1], np.int32); slots = np.arange(S, dtype=np.int32)
prompts = [(rng.integers(6, 154, ctx0).astype(np.int32), rng.integers(1, 3, ctx0).astype(np.int32), rng.integers(0, 3, (4, ctx0)).astype(np.int32)) for _ in range(S)]
def reset():
    st = dec._stream()
    for s, (ids, cls, a4) in enumerate(prompts):
        _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, ctx0, tg.ctypes.data, -1, min(1000, 1024 - ctx0), st), "bb")
    _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, 4, st), "step"); torch.cuda.synchronize(dev)
streams = [torch.cuda.Stream() for _ in range(3)]
stop = False
def noise(i, big):
    torch.cuda.set_device(0)
    e, g = C.c_double(), C.c_double()
    while not stop:
        lib.etd_debug_boundary_cost(64, 20, big, C.c_void_p(streams[i].cuda_stream), C.byref(e), C.byref(g))
for n_noise in (0, 1, 3):
    reset(); stop = False
    th = [threading.Thread(target=noise, args=(i, 0)) for i in range(n_noise)]
    for t in th: t.start()
    time.sleep(0.2)
    t0 = time.perf_counter()
    _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, S, steps, dec._stream()), "step"); dec._ts.synchronize()
    dt = time.perf_counter() - t0
    stop = True
    for t in th: t.join()
    print(f"decode step with {n_noise} chains of empty kernels running beside it: {1e3*dt/steps:.4f} ms per step")

#!/usr/bin/env python3
"""Are the greedy token streams of the headline decode workload reproducible run to run?  Same jobs, R repetitions, E engines:
prints a digest per repetition and, where two repetitions differ, the first (job, bar, position) that does."""
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from etude_amd import synth  # noqa: E402
from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines  # noqa: E402

if __name__ == "__main__":
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 216
    n_bars = int(sys.argv[3]) if len(sys.argv) > 3 else 92
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    prec = sys.argv[5] if len(sys.argv) > 5 else "f16"
    bar_tokens = int(sys.argv[6]) if len(sys.argv) > 6 else 48          # 1: every bar is a prefill and nothing else
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    per = (n_jobs + E - 1) // E
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision=prec, max_streams=per)]
    import os
    if os.environ.get("PROBE_NOCLONE"):          # every engine with its own copy of the weights
        decs += [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision=prec, max_streams=per) for _ in range(E - 1)]
    else:
        decs += [decs[0].clone() for _ in range(E - 1)]
    if os.environ.get("PROBE_SAMEJOBS"):         # every engine gets the same jobs: engines can be compared with each other
        pass
    vocab = bench.make_vocab()
    grid = [(p, r, s_) for p in range(3) for r in range(3) for s_ in range(3)]
    jobs = []
    for k in range(n_jobs):
        bars = synth.song_bars(seed=1234 + k // 27, n_bars=n_bars)
        p, r, s_ = grid[k % 27]
        jobs.append((bars, [synth.attrs(p, r, s_, 2)] * len(bars)))
    outs = []
    noise = len(sys.argv) > 7 and sys.argv[7] == "noise"        # reps alternate: without / with NaN-poisoning workgroups on a second stream
    stop = [False]
    if noise:
        import ctypes as C
        import threading
        nl = C.CDLL(str(Path(__file__).resolve().parent / "ubench" / "liblds_noise.so"))
        nl.noise_launch.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        sink = torch.zeros(16, dtype=torch.int32, device=dev)
        nst = torch.cuda.Stream(device=dev)

        def noisy():
            torch.cuda.set_device(0)
            while not stop[0]:
                for _ in range(64):
                    nl.noise_launch(256, 0x7fc00000, sink.data_ptr(), nst.cuda_stream)       # quiet NaN in every LDS word
                nst.synchronize()
    for rep in range(reps):
        th = None
        if noise and rep % 2 == 1:
            stop[0] = False
            th = threading.Thread(target=noisy)
            th.start()
        out, _ = run_engines(decs, jobs, vocab, force_bar_tokens=bar_tokens)()
        if th is not None:
            stop[0] = True
            th.join()
        torch.cuda.synchronize(dev)
        flat = np.asarray([t for job in out for bar in job for t in bar], np.int32)
        print(f"E={E} jobs={n_jobs} bars={n_bars} {prec} rep {rep}: {hashlib.sha256(flat.tobytes()).hexdigest()[:16]}  ({flat.size} tokens)", flush=True)
        outs.append(out)
    for rep in range(1, reps):
        if outs[rep] != outs[0]:
            nd = 0; first = None
            for j, (a, b) in enumerate(zip(outs[0], outs[rep])):
                if a != b:
                    nd += 1
                    if first is None:
                        for bi, (ba, bb) in enumerate(zip(a, b)):
                            if list(ba) != list(bb):
                                pos = next(i for i, (x, y) in enumerate(zip(ba, bb)) if x != y) if len(ba) == len(bb) else -1
                                first = (j, bi, pos)
                                break
            print(f"  rep {rep} differs from rep 0 in {nd} of {len(outs[0])} jobs; first difference at (job, bar, position) = {first}", flush=True)

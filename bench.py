#!/usr/bin/env python3
"""Benchmark of Etude's two hot paths on MI355X: audio-seconds/s transcribed + decoder tokens/s.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = ONE PASS OVER THE BATCH north_star NAMES: 64 x 3-minute 44.1 kHz stereo clips (BASELINE.json configs[4]), sharded 64 / N
per rank -- at N = 8 exactly configs[4], at N = 1 the whole batch on one MI355X ("strong" scaling: the batch is fixed).  Per rank the
step is infer.py's stage sequence for its clips (etude_amd/pipeline.py):

    per clip   wav (already resident in HBM) -> channel mean, resample, STFT / log-mel, hFT-Transformer over 22 windows, device note
               picking -> the note list extract() writes; volume contour (analyze_volume); TinyREMITokenizer on the synthetic
               tempo.json (stage 2, beat detection, is out of scope) -> vocab ids -> condition bars
    per job    (clip, attribute tuple) for the 27 tuples {0,1,2}^3, overlap bin 2: greedy generate() over the clip's OWN ~92 condition
               bars -- all jobs as concurrent device streams of one decoder engine (1728 rows per decode-step launch) -- then decode_to_notes with
               the clip's volume map (MIDI file writing is left out: file I/O)
    ranks work on different clips with no data-path collective; ONE final gather of the token ids (RCCL).

--bar-tokens (48) tokens are generated per bar and Bar_EOS does not end a bar: seeded synthetic weights carry no musical EOS statistics,
so bar lengths would otherwise be an artefact of the seed (the reference's own chain ends in degenerate 400-token bars on these
weights).  Everything else -- prompt assembly, 4-pair history, truncation to the last 512 prompt tokens, KV reset per bar -- is
generate()'s.  value = audio seconds taken through the WHOLE chain per wall second, all ranks.  Next to it:
  extract_audio_s_per_s / decoder_tokens_per_s   the per-stage rates the metric names
  roofline      the dominant kernel (k_dstep_attn_down) in the TIMED configuration: device-side span of every launch of one extra
                decode stage with all engines running (etd_decoder_stamp), algorithmic bytes from the library's exact counters;
                decode_stage = all decode-step bytes of the timed steps / the decode stage's wall time
  cpu_baseline  the CPU oracle on this node's host cores, bounded sample (rank 0, N = 1 only)
  extras        outside the timed region: configs[2] (extractor only), configs[1] (one clip), configs[3] (128 streams at ctx 512 and 3.5 k); parity_mode (the chain with the fp32
                extractor + fp32 decoder on the largest of 64 / 32 / 16 / 8 clips the budget allows -- what exact parity costs -- with its own roofline and, as
                parity_mode.oracle_check, the CPU oracle's bars of cpu_baseline() compared id by id with that pass's job for the same clip / tuple / bars); ragged_bars (the
                reference's stopping rule -- Bar_EOS, 512 / 25 600 limits -- under continuous batching, and the reference's own 13 217 ids of tests/golden/clip_ctx.npz in the
                exact-parity mode); n8_share (one rank's share of the N = 8 run); bar_tokens_144 (the decode stage at 144 forced tokens per bar)
Harness budget: the driver runs `--steps 20 --warmup 5` under a 600 s wall-clock limit; bench.py plans against ETD_BENCH_BUDGET_S (default 560 s, counted from process
start).  The first warm-up step is always a full step (and the estimate).  The other warm-up steps are full steps only if that leaves room for everything that follows the
timed steps (~130 s: 24 stamped bars, serial event pass, extras incl. the exact-parity pass, CPU baseline); otherwise they run 4 bars per job (same launches, same widths:
everything is allocated, captured and cached by then).  Behind the timed steps the optional parts are sized / shed by the time left: extras.parity_mode (clips 64 -> 32 -> 16 -> 8, then bars), bar_tokens_144 (clips), ragged_bars,
n8_share, the configs[1..3] extras, the serial pass, the stamped stage 24 -> 8 bars; the CPU baseline goes last.  Only if K full steps + that minimum cannot fit does the batch per step shrink
(32 / 16 / 8 clips per rank) -- all of it written into config.workload / config.warmup_step.  The K timed steps are always full steps of the stated batch.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

# With the HIP runtime's default of 4 hardware queues, a fifth stream in the process (engines + torch's default stream + e.g.
# RCCL's at N > 1) makes streams share a queue and costs 25 % of the throughput; 8 queues remove that cliff and allow a fourth
# engine (+2 %).  Five or more concurrently submitting engines halve the throughput whatever the queue count.  Must be set before HIP
# initialises; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 / f16 MFMA (MI355X_MICROARCH.md: the F16 forms take the same cycles)
PEAK_HBM_GBS = 8000.0         # HBM3E spec
T_START = time.perf_counter()
# cost model of extras.parity_mode (seconds, measured on one MI355X: profiles/r05_parity_full.json, one engine x 27 * clips streams from 512 jobs up): fp32 engines' set-up +
# warm-up pass, then per clip the fp32 extract, the fp32 decode of its 27 jobs x all bars (+ its share of the 8 stamped bars); the 16-bit decode of the same jobs
# (the per-bar divergence) runs on the first 8 clips only
PARITY_FIXED_S = float(os.environ.get("ETD_PARITY_FIXED_S", "9"))
PARITY_EXTRACT_S_PER_CLIP = float(os.environ.get("ETD_PARITY_EXTRACT_S", "0.14"))
PARITY_DECODE_S_PER_CLIP = float(os.environ.get("ETD_PARITY_DECODE_S", "0.75"))
PARITY_DECODE_S_PER_CLIP_SMALL = float(os.environ.get("ETD_PARITY_DECODE_SMALL_S", "1.5"))      # 8 clips: 4 x 54 streams, the latency-bound regime
PARITY_F16_S_PER_CLIP = float(os.environ.get("ETD_PARITY_F16_S", "0.45"))
PARITY_F16_CLIPS = 8


def parity_cost(clips: int, bars: int, nb_mean: float) -> float:
    """estimated wall seconds of parity_mode_extras(clips, bars)"""
    dec = PARITY_DECODE_S_PER_CLIP if clips * 27 >= 512 else PARITY_DECODE_S_PER_CLIP_SMALL
    frac = (bars or nb_mean) / nb_mean
    return PARITY_FIXED_S + clips * (PARITY_EXTRACT_S_PER_CLIP + dec * (frac + 8.0 / nb_mean)) + min(clips, PARITY_F16_CLIPS) * PARITY_F16_S_PER_CLIP * frac


def since_process_start() -> float:
    """wall seconds since this PROCESS started (interpreter start-up and `import torch` included: the harness's clock runs from there)"""
    try:
        import psutil
        return time.time() - psutil.Process().create_time()
    except Exception:
        return time.perf_counter() - T_START + 15.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - T_START:7.1f}s]", *a, file=sys.stderr, flush=True)


def count_gpus_without_hip() -> int:
    """GPUs of this node as the kernel driver lists them (KFD topology: nodes with SIMDs), WITHOUT initialising HIP: the launcher
    process must never touch the GPU (a process that has cannot start other programs on this pool, and its children inherit nothing
    useful from it).  Honours ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when they list indices."""
    n = 0
    base = Path("/sys/class/kfd/kfd/topology/nodes")
    if base.is_dir():
        for nd in base.iterdir():
            try:
                props = dict(l.split(None, 1) for l in (nd / "properties").read_text().splitlines() if " " in l)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except Exception:
                pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""])) if n else len([x for x in v.split(",") if x.strip() != ""])
    return n


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run, one per GPU, over
    RCCL on 127.0.0.1.  The parent never touches the GPU (GPUs are counted from the KFD topology in sysfs); the ranks are FRESH child
    processes, rank 0's ONE JSON line is relayed and their exit status becomes ours.  Fails loudly -- exit 3, no JSON line -- when the
    node has fewer than N GPUs."""
    import socket
    import subprocess
    have = count_gpus_without_hip()
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible on this node; not reporting a {n}-GPU number", file=sys.stderr)
        return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("ETD_FORCE_SPAWN", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def make_vocab():
    from etude_amd import synth
    from etude_amd.vocab import Vocab
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def cpu_baseline(bars, seconds_budget: float = 25.0, clip_seconds: float = 180.0, windows_per_clip: int = 22, attr_tuples: int = 27, bars_per_job: int = 92,
                 bar_tokens: int = 48):
    """The CPU oracle (a restatement of the reference, pinned by golden vectors) on this node's host cores, on a bounded sample of
    the SAME workload: 2 extractor windows; the first bars of one clip's own condition bars through the oracle's generate loop
    (prompt pass + `bar_tokens` forced tokens per bar).  `bars`: that clip's condition bars (lists of ids)."""
    import torch
    from etude_amd import synth
    from oracle import hft, neox
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(avail, 32)          # torch-CPU GEMMs of this size stop scaling (and regress) beyond ~32 threads
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v) for k, v in synth.extractor_state_dict(0).items()}
    x = torch.from_numpy(synth.window_features(5, 1))
    d = hft.HftDims()
    t0 = time.time()
    nwin = 0
    while nwin < 2 and (nwin == 0 or time.time() - t0 < seconds_budget * 0.5):
        hft.model_forward(sd, x, d)
        nwin += 1
    t_ext = time.time() - t0
    tsd = {k: torch.from_numpy(v) for k, v in synth.decoder_state_dict(1, {}).items()}
    dthreads = min(avail, 8)        # batch-1 token loop: small ops, more threads only add sync cost
    torch.set_num_threads(dthreads)
    # bars 0..4 of the clip: bar 4 is the first with a full 4-pair history (a steady-state prompt, truncated to 512 + Bar_BOS like
    # every later bar); the sample's LAST bar is timed on its own and stands for the steady state
    nb = min(5, len(bars))
    t0 = time.time()
    out = neox.generate_ids(tsd, neox.NeoxDims(), 4, 5, bars[: nb - 1], [synth.attrs()] * (nb - 1), max_bar_token_limit=512, force_bar_tokens=bar_tokens)
    t_ramp = time.time() - t0
    # the steady-state bar: history = the four bars just generated
    hist_x, hist_y = bars[nb - 5: nb - 1], out[nb - 5: nb - 1]
    user_keys = sorted(synth.attrs().keys())
    history = [(xb, yb, synth.attrs()) for xb, yb in zip(hist_x, hist_y)]
    toks, cls, al = neox.build_bar_prompt(history, bars[nb - 1], synth.attrs(), user_keys, 4, 5, neox.NeoxDims(), 512, 0.5)
    t0 = time.time()
    ids_t, cls_t = torch.tensor([toks]), torch.tensor([cls])
    at_t = {neox.ATTR_KEY_MAP[k]: torch.tensor([al[k]]) for k in user_keys}
    kv = None
    ntok = 0
    steady = [4]                 # (Bar_BOS: bar_bos_id 4 of the generate_ids call above)
    with torch.no_grad():
        for _ in range(bar_tokens):
            logits, kv = neox.forward_logits(tsd, neox.NeoxDims(), ids_t, cls_t, at_t, kv)
            nxt = int(torch.argmax(logits[:, -1, :], dim=-1).item())
            steady.append(nxt)
            ntok += 1
            ids_t, cls_t = torch.tensor([[nxt]]), torch.tensor([[2]])
            at_t = {neox.ATTR_KEY_MAP[k]: torch.tensor([[synth.attrs()[k]]]) for k in user_keys}
    t_bar = time.time() - t0
    assert ntok == bar_tokens and all(len(b) == bar_tokens + 1 for b in out), "cpu_baseline: the sample bars must run the full forced length"
    # the headline's unit: audio seconds taken through BOTH stages (extract once, decode for every attribute tuple) per wall second,
    # extrapolated from the bounded sample: extract = windows_per_clip x (time per window); decode = tuples x bars x (time of a steady-state bar)
    ext_s_per_clip = (t_ext / nwin) * windows_per_clip
    dec_s_per_clip = attr_tuples * bars_per_job * t_bar
    return {"value": round(clip_seconds / (ext_s_per_clip + dec_s_per_clip), 4),
            "unit": "audio-s/s (each clip extracted and decoded for every attribute tuple; extrapolated from the sample)", "cores": cores, "kind": "port",
            "sample": f"oracle hFT forward on {nwin} window(s) of 512 frames (8.192 s audio each) in {t_ext:.1f}s on {cores} threads; "
                      f"oracle generate on one clip's own condition bars: {nb - 1} ramp-up bars in {t_ramp:.1f}s, then ONE steady-state bar "
                      f"(prompt of {len(toks)} tokens + {bar_tokens} forced tokens) in {t_bar:.1f}s on {dthreads} threads",
            "extract_audio_s_per_s": round(nwin * 8.192 / t_ext, 4),
            "decoder_tokens_per_s": round(bar_tokens / t_bar, 2), "decoder_cores": dthreads,
            "_oracle_ids": [list(b) for b in out] + [steady]}      # (popped by the caller: the checker leg compares them with the device engines' ids)

def default_engines(n_jobs: int, world: int = 1) -> int:
    """decoder engines for `n_jobs` jobs on one rank when --engines is not given: whatever measured fastest (round 5, profiles/r05_engines.txt: 64 clips on one GPU 567 /
    574 / 588 / 581 audio-s/s with 1 / 2 / 3 / 4 engines, same token digest): three chains of launches overlap each other's small kernels and bar boundaries while a
    launch still covers most of the chip; small per-rank batches (216 jobs at N = 8) stay on four (the latency-bound regime)"""
    return 3 if (n_jobs >= 1024 or (world == 1 and n_jobs >= 512)) else 4


def union_seconds(pairs, tick_s: float = 1e-8) -> float:
    """length of the union of [start, end) intervals given in device ticks (uint64 [n, 2], any order): the time during which at least one of them was running"""
    iv = np.asarray(pairs)
    if iv.size == 0:
        return 0.0
    iv = iv.reshape(-1, 2)
    iv = iv[np.argsort(iv[:, 0], kind="stable")].astype(np.int64)
    cur_s, cur_e = int(iv[0, 0]), int(iv[0, 1])
    tot = 0
    for a_, b_ in iv[1:]:
        if a_ > cur_e:
            tot += cur_e - cur_s
            cur_s, cur_e = int(a_), int(b_)
        elif b_ > cur_e:
            cur_e = int(b_)
    tot += cur_e - cur_s
    return tot * tick_s


def n8_share_extras(args, dev, exs, wavs, grid, vocab, headline_value):
    """extras.n8_share: what ONE rank of the N = 8 run does -- 8 of the batch's clips x 27 tuples = 216 jobs on four engines of 54 streams (`default_engines`), one
    timed pass after a 2-bar warm-up pass -- measured here on the one GPU of an N = 1 run, so that the line carries the scaling expectation (8 x this figure against the
    N = 1 value) whether or not a node is available to the driver.  The driver computes the measured efficiency itself from its own N = 1, 2, 4, 8 runs."""
    import torch
    from etude_amd import synth
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from etude_amd.pipeline import ClipBatchPipeline, synthetic_tempo
    n_clips = 8
    n_jobs = n_clips * len(grid)
    n_eng = default_engines(n_jobs, 8)
    per_eng = (n_jobs + n_eng - 1) // n_eng
    decs = []; pipe = None
    try:
        decs = [EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()), synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=per_eng,
                             max_prefill_rows=min(262144, per_eng * 520))]
        decs += [decs[0].clone() for _ in range(n_eng - 1)]
        pipe = ClipBatchPipeline(exs[:2], decs, vocab, synthetic_tempo(), grid, 44100, force_bar_tokens=args.bar_tokens)
        sub = wavs[:n_clips]
        conds = pipe.extract_stage(sub)
        pipe.decode_stage(conds, max_bars=2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r = pipe.run(sub)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        v = args.seconds * n_clips / dt
        return {"workload": f"one rank's share of the N = 8 run on this GPU: {n_clips} clips x {len(grid)} tuples = {n_jobs} jobs, {n_eng} engines x {per_eng} streams, the whole chain, one pass",
                "s_per_step": round(dt, 3), "audio_s_per_s": round(v, 2), "extract_s": round(r["t_extract"], 3), "decode_s": round(r["t_decode"], 3),
                "projected_n8_audio_s_per_s": round(8 * v, 1), "projected_n8_over_n1": round(8 * v / headline_value, 2),
                "note": "projection = 8 x this figure (ranks share nothing but the final gather of token ids); the small per-rank batch is latency-bound (25 dependent launches per "
                        "decode step at 54 rows), which is why 8 ranks are projected at less than 8 x the N = 1 value"}
    finally:
        for d in reversed(decs):
            try:
                d.close()
            except Exception:      # noqa: BLE001
                pass
        if pipe is not None:
            try:
                pipe.close()
            except Exception:      # noqa: BLE001
                pass


def bar_divergence(ra, rb):
    """per-bar comparison of two result lists [(flat ids, bar lengths)] of the same jobs (tests/test_gpu_full_configs.py: a bar can only be compared
    while the two histories are still equal): -> (bars identical, comparable bars, jobs identical end to end)"""
    same = comparable = jobs_same = 0
    for (fa, la), (fb, lb) in zip(ra, rb):
        oa, ob = np.concatenate([[0], np.cumsum(la)]), np.concatenate([[0], np.cumsum(lb)])
        nb = min(len(la), len(lb))
        k = 0
        while k < nb and la[k] == lb[k] and np.array_equal(fa[oa[k]:oa[k + 1]], fb[ob[k]:ob[k + 1]]):
            k += 1
        same += k
        comparable += min(k + 1, nb)
        jobs_same += int(k == nb and len(la) == len(lb))
    return same, comparable, jobs_same


DEFAULT_TUPLE_INDEX = 13      # (polyphony, rhythm, sustain) = (1, 1, 1) in pipeline.attr_grid(27): infer.py's CLI defaults, the tuple cpu_baseline() decodes


def first_bars(flat, lens, n):
    """the first n bars of one job's (flat ids, bar lengths) as id lists"""
    ends = np.cumsum(np.asarray(lens, np.int64))
    return [np.asarray(flat[e - l: e]).tolist() for l, e in zip(np.asarray(lens).tolist()[:n], ends.tolist()[:n])]


def parity_mode_extras(args, dev, wavs, grid, vocab, n_clips, max_bars, f16_engines, time_left, headline_decode_s_per_clip=None):
    """extras.parity_mode: what exact parity costs.  north_star's "identical token-id sequences under greedy decode" holds in the fp32 mode (the reference
    runs fp32: etude_decoder.py:333); the headline is timed in the 16-bit serving mode (IEEE-half operands).  The SAME chain (extract .. notes) on the first `n_clips` clips of this rank with the fp32
    extractor and fp32 decoder engines (every dense contraction at fp32 grade on the f16 matrix cores: csrc/gemm3.h) -- from 512 jobs up ONE engine x 27 * n_clips streams, the layout
    tests/test_gpu_decoder_parity.py holds to the oracle at 1 728 rows -- ONE timed pass after a 2-bar warm-up pass; then 8 stamped bars (4 of them steady-state) for the roofline of the
    fp32 attention launches; then the 16-bit decoder on the SAME condition bars (the fp32 extractor's; the first 8 clips) for the per-bar divergence of the two decoders.
    Returns (dict for the JSON line, oracle-check material or None): the first five condition bars of clip 0 and the ids the fp32 (and 16-bit) engines generated for them with the
    default attribute tuple, INSIDE this batch -- what cpu_baseline()'s oracle bars are compared with.  Whatever happens, every engine and the extractor are closed on the way out."""
    import torch
    from etude_amd import synth
    from etude_amd.config import ExtractorConfig
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from etude_amd.extractor import AMTAPC_Extractor
    from etude_amd.pipeline import ClipBatchPipeline, synthetic_tempo
    t_in = time.perf_counter()
    n_jobs = n_clips * len(grid)
    n_eng = 1 if n_jobs >= 512 else (4 if n_jobs >= 64 else 1)
    per_eng = (n_jobs + n_eng - 1) // n_eng
    ex32 = None; d32 = []; pipe32 = None; pipe16 = None
    check = None
    try:
        ex32 = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(0), dev, max_windows=4, precision="fp32")
        d32 = [EtudeDecoder(EtudeDecoderConfig(**synth.decoder_dims()), synth.decoder_state_dict(1, {}), dev, precision="fp32", max_streams=per_eng,
                            max_prefill_rows=min(131072, per_eng * 520))]
        d32 += [d32[0].clone() for _ in range(n_eng - 1)]
        pipe32 = ClipBatchPipeline([ex32], d32, vocab, synthetic_tempo(), grid, 44100, force_bar_tokens=args.bar_tokens)
        sub = wavs[:n_clips]
        conds = pipe32.extract_stage(sub)
        tw = time.perf_counter()
        pipe32.decode_stage(conds, max_bars=2)                     # allocations, graph captures
        torch.cuda.synchronize(dev)
        # the warm-up pass measured this box: 2 bars (the cheap, short-context ones) -> a bound on the full pass; shed bars if the time left is short of it
        per_bar = (time.perf_counter() - tw) / 2.0
        nb_mean = float(np.mean([len(cd.bars) for cd in conds]))
        if not max_bars and 2.2 * per_bar * nb_mean > time_left() - 8.0:
            max_bars = int(max(4, min(nb_mean, (time_left() - 8.0) / (2.2 * per_bar))))
        for d in d32:
            d.stats_reset()
        t0 = time.perf_counter()
        conds = pipe32.extract_stage(sub)
        t1 = time.perf_counter()
        res32, st32 = pipe32.decode_stage(conds, max_bars=max_bars)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        pipe32.notes_stage(conds, res32)
        t3 = time.perf_counter()
        ntok = sum(s["tokens"] for s in st32)
        frac = 1.0
        if max_bars:
            frac = min(1.0, max_bars / max(1.0, nb_mean))
        out = {"workload": f"the headline's chain on {n_clips} of its clips x {len(grid)} tuples = {n_jobs} jobs in the EXACT-PARITY mode: fp32 extractor (etd_ext_cfg.precision 1) "
                           f"+ fp32 decoder (fp32 weights, KV cache and activations; every dense contraction as a two-plane f16 split, three MFMAs per product, fp32 accumulate: "
                           f"csrc/gemm3.h; {n_eng} engine(s) x {per_eng} streams), {args.bar_tokens} tokens per bar"
                           + (f"; ONLY THE FIRST {max_bars} BARS of every job were decoded to stay inside the harness budget (audio_s_per_s scales the decode stage to all bars)" if max_bars else ""),
               "clips": n_clips, "jobs": n_jobs, "engines": n_eng, "streams_per_engine": per_eng,
               "extract_s": round(t1 - t0, 3), "decode_s": round(t2 - t1, 3), "notes_s": round(t3 - t2, 3),
               "audio_s_per_s": round(args.seconds * n_clips / ((t1 - t0) + (t2 - t1) / frac + (t3 - t2) / frac), 2),
               "extract_audio_s_per_s": round(args.seconds * n_clips / (t1 - t0), 1), "decoder_tokens_per_s": round(ntok / (t2 - t1), 1),
               "tokens_sha256": hashlib.sha256(np.concatenate([r[0] for r in res32]).astype(np.int32).tobytes()).hexdigest()[:16],
               "full_batch": "profiles/r06_parity_full.json: the same mode on the whole 64-clip batch (one engine x 1728 streams), a gpurun of tools/bench_parity.py --clips 64 --stamp"}
        if headline_decode_s_per_clip and n_jobs >= 512:
            # both stages at >= 512 rows per launch (throughput regime): the 16-bit decode stage of the timed steps against this pass, per clip
            out["fp32_over_f16_decode_time"] = round((t2 - t1) / frac / n_clips / headline_decode_s_per_clip, 2)
        if len(grid) > DEFAULT_TUPLE_INDEX and len(conds[0].bars) >= 5 and (not max_bars or max_bars >= 5):
            check = {"bars": [conds[0].bars.bar(i) for i in range(5)], "fp32": first_bars(*res32[DEFAULT_TUPLE_INDEX], 5),
                     "where": f"job (clip 0, tuple (1, 1, 1)) of the exact-parity batch above: row {DEFAULT_TUPLE_INDEX} of {per_eng} rows per launch on engine 0 of {n_eng}"}
        # roofline of the fp32 attention launches (k_dattn<float>: 4-byte K / V, no weights in the launch), steady-state bars, device stamps as for the headline
        if time_left() > 12.0 + 8.0 * per_bar * 1.5 and nb_mean >= 8:
            for d in d32:
                d.stamp(True, skip_steps=4 * (args.bar_tokens - 1))
                d.stats_reset()
            pipe32.decode_stage(conds, max_bars=8)
            torch.cuda.synchronize(dev)
            s2 = [d.stats() for d in d32]
            for d in d32:
                d.stamp(False)
            launches = sum(s["stamped_launches"] for s in s2); secs = sum(s["stamped_seconds"] for s in s2); byts = sum(s["stamped_alg_bytes"] for s in s2)
            if launches > 0 and secs > 0:
                out["roofline"] = {"kernel": "k_dattn<float>", "bound": "hbm", "achieved": round(byts / secs / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(byts / secs / 1e9 / PEAK_HBM_GBS, 4), "launches": int(launches), "avg_launch_ms": round(1e3 * secs / launches, 5),
                                   "alg_bytes_per_launch": byts / launches, "rows_per_launch": per_eng, "traffic": None,
                                   "note": "device stamps of every attention launch of bars 4-7 (steady-state contexts), all engines running; algorithmic bytes = fp32 K + V rows of every (row, head) context"}
        for d in reversed(d32):
            d.close()
        d32 = []
        pipe32.close(); pipe32 = None
        # the 16-bit decoder on the same condition bars (the first 8 clips of the pass): the headline's engines when they hold enough streams, else nothing (no new allocations this late)
        n16 = min(n_clips, PARITY_F16_CLIPS)
        j16 = n16 * len(grid)
        if time_left() > 6.0 + n16 * PARITY_F16_S_PER_CLIP * frac * 1.5 and f16_engines and sum(d.max_streams for d in f16_engines) >= j16:
            pipe16 = ClipBatchPipeline([ex32], f16_engines, vocab, synthetic_tempo(), grid, 44100, force_bar_tokens=args.bar_tokens)
            t4 = time.perf_counter()
            res16, st16 = pipe16.decode_stage(conds[:n16], max_bars=max_bars)
            torch.cuda.synchronize(dev)
            t5 = time.perf_counter()
            same, comparable, jobs_same = bar_divergence(res32[:j16], res16)
            out["f16_same_conditions"] = {"clips": n16, "decode_s": round(t5 - t4, 3), "decoder_tokens_per_s": round(sum(s["tokens"] for s in st16) / (t5 - t4), 1),
                                           "bars_identical": same, "bars_comparable": comparable, "bar_divergence_rate": round(1.0 - same / max(1, comparable), 5),
                                           "jobs_identical_end_to_end": jobs_same, "jobs": j16,
                                           "note": f"the headline's 16-bit decoder engines (IEEE-half weights and KV cache; rounds 1-4: bf16, 8.5 % of the bars diverged) on the fp32 extractor's condition bars of the first {n16} clips; a bar is comparable while both histories are still equal"}
            if check is not None:
                check["f16"] = first_bars(*res16[DEFAULT_TUPLE_INDEX], 5)
        out["wall_s"] = round(time.perf_counter() - t_in, 2)
        return out, check
    finally:
        for d in reversed(d32):
            try:
                d.close()
            except Exception:      # noqa: BLE001
                pass
        for p_ in (pipe32, pipe16):
            if p_ is not None:
                try:
                    p_.close()
                except Exception:      # noqa: BLE001
                    pass
        if ex32 is not None:
            try:
                ex32.close()
            except Exception:      # noqa: BLE001
                pass


def ragged_bars_extras(args, dev, conds, grid, vocab, time_left):
    """extras.ragged_bars: the reference's OWN stopping rule under load (etude_decoder.py:300-352): a bar ends at Bar_EOS or after max_bar_token_limit = 512 tokens, the song at
    max_output_tokens = 25 600.  The benchmark weights never emit Bar_EOS at a musically meaningful rate, so this uses the context weights of the goldens
    (synth.decoder_state_dict_ctx: on the reference's configs[1] song 81 of 92 bars end in Bar_EOS, mean ~144 tokens): 8 of the batch's clips x 27 tuples on 4 engines x 54 streams of
    the 16-bit serving mode, ragged bars under continuous batching (a stream that ends its bar waits for the engine's next begin_bars pass; the scheduler admits the next bar of every
    finished stream together).  Reported: tokens/s, bars ended by EOS, tokens per bar, mean active rows per decode step (scheduler occupancy = that / streams).
    Then, in the EXACT-PARITY mode, the reference's own song: tests/golden/clip_ctx.npz -- the (1,1,1) job of the configs[1] clip (seed 1234, the reference's condition bars from
    clip_full.npz), 13 217 ids generated by the reference's generate(): `clip_ctx_ids_identical`."""
    import torch
    from etude_amd import synth
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines
    n_clips = min(8, len(conds))
    n_jobs = n_clips * len(grid)
    n_eng = default_engines(n_jobs, 8)
    per_eng = (n_jobs + n_eng - 1) // n_eng
    decs = []
    out = {}
    try:
        sd = synth.decoder_state_dict_ctx(1)
        dcfg = EtudeDecoderConfig(**synth.decoder_dims())
        decs = [EtudeDecoder(dcfg, sd, dev, precision="f16", max_streams=per_eng, max_prefill_rows=min(262144, per_eng * 520))]
        decs += [decs[0].clone() for _ in range(n_eng - 1)]
        a4s = [np.asarray([a[k] for k in ("pitch_overlap_bin", "polyphony_bin", "sustain_bin", "rhythm_intensity_bin")], np.int32) for a in grid]
        jobs = [(cd.bars, np.tile(a4, (len(cd.bars), 1))) for cd in conds[:n_clips] for a4 in a4s]
        kw = dict(max_output_tokens=25600, max_bar_token_limit=512, temperature=0.0, as_arrays=True)
        short = [(type(j[0])(j[0].ids[: j[0].offsets[2]], j[0].offsets[:3]), j[1][:2]) for j in jobs]
        run_engines(decs, short, vocab, **kw)()                  # allocations, graph captures
        torch.cuda.synchronize(dev)
        for d in decs:
            d.stats_reset()
        t0 = time.perf_counter()
        res, st = run_engines(decs, jobs, vocab, **kw)()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        dst = [d.stats() for d in decs]
        eos = vocab.get_bar_eos_id()
        lens = np.concatenate([r[1] for r in res]) - 1                         # generated tokens per bar (Bar_BOS is the prompt's)
        n_bars = int(lens.size)
        ends = [np.cumsum(r[1]) for r in res]
        n_eos = int(sum(int((r[0][e - 1] == eos).sum()) for r, e in zip(res, ends)))
        ntok = int(sum(s["tokens"] for s in st))
        steps = sum(s["steps"] for s in dst); row_steps = sum(s["row_steps"] for s in dst)
        out = {"workload": f"{n_clips} of the batch's clips x {len(grid)} tuples = {n_jobs} jobs, context weights (synth.decoder_state_dict_ctx), the reference's stopping rule: Bar_EOS ends a bar, "
                           f"max_bar_token_limit 512, max_output_tokens 25600; 16-bit serving mode, {n_eng} engines x {per_eng} streams, each clip's own condition bars, one pass after a 2-bar warm-up pass",
               "decode_s": round(dt, 3), "tokens": ntok, "decoder_tokens_per_s": round(ntok / dt, 1), "bars": n_bars, "bars_ended_by_eos": n_eos,
               "tokens_per_bar_mean": round(float(lens.mean()), 1), "tokens_per_bar_max": int(lens.max()), "jobs_at_max_output_tokens": int(sum(int(r[1].sum() - len(r[1]) >= 25600) for r in res)),
               "decode_steps": int(steps), "mean_active_rows_per_step": round(row_steps / max(1.0, steps), 2), "scheduler_occupancy": round(row_steps / max(1.0, steps) / per_eng, 4),
               "audio_s_per_s_decode_only": round(args.seconds * n_clips / dt, 1)}
        for d in reversed(decs):
            d.close()
        decs = []
        # ---- the reference's own ids for the configs[1] song under this rule, exact-parity mode
        gdir = ROOT / "tests" / "golden"
        if time_left() > 14.0 and (gdir / "clip_ctx.npz").exists() and (gdir / "clip_full.npz").exists():
            gc, gf = np.load(gdir / "clip_ctx.npz"), np.load(gdir / "clip_full.npz")
            flat, bl = gf["bar_ids"].tolist(), gf["bar_lens"].tolist()
            bars, p_ = [], 0
            for l in bl:
                bars.append(flat[p_: p_ + l]); p_ += l
            d32 = EtudeDecoder(dcfg, sd, dev, precision="fp32", max_streams=1)
            decs = [d32]
            t0 = time.perf_counter()
            got = d32.generate_ids(vocab, bars, [synth.attrs(1, 1, 1, 2)] * len(bars), temperature=0.0)
            dt32 = time.perf_counter() - t0
            ids = [t for b_ in got for t in b_]
            ref = gc["gen_ids"].tolist()
            out["clip_ctx_fp32"] = {"what": "tests/golden/clip_ctx.npz: the reference's generate() (default limits, Bar_EOS ends bars) on the configs[1] clip's 92 condition bars, tuple (1,1,1,2), "
                                            "context weights; here the exact-parity engine, one stream",
                                    "ids": len(ids), "reference_ids": len(ref), "clip_ctx_ids_identical": bool(ids == ref), "decode_s": round(dt32, 2),
                                    "decoder_tokens_per_s": round((len(ids) - len(got)) / dt32, 1)}
        return out
    finally:
        for d in reversed(decs):
            try:
                d.close()
            except Exception:      # noqa: BLE001
                pass


def bar_tokens_sensitivity(args, dev, pipe, conds, n_clips, bar_tokens, headline_tokens_per_s):
    """extras.bar_tokens_144: the headline depends on the 48 generated tokens per bar SURVEY 8(d) sanctioned; the reference's goldens average 144-279 tokens per bar.  One decode
    stage of the first `n_clips` clips' jobs on the headline's engines at `bar_tokens` forced tokens per bar (contexts grow to 513 + bar_tokens)."""
    import torch
    old = pipe.force_bar_tokens
    try:
        pipe.force_bar_tokens = bar_tokens
        sub = conds[:n_clips]
        pipe.decode_stage(sub, max_bars=2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res, st = pipe.decode_stage(sub)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        ntok = sum(s["tokens"] for s in st)
        return {"workload": f"the decode stage of {n_clips} clips x {len(pipe.attrs)} tuples on the headline's {len(pipe.decs)} engines at {bar_tokens} forced tokens per bar instead of {old} "
                            "(extract and notes stages unchanged)", "clips": n_clips, "bar_tokens": bar_tokens, "decode_s": round(dt, 3), "tokens": int(ntok),
                "decoder_tokens_per_s": round(ntok / dt, 1), "vs_headline_decoder_tokens_per_s": round(ntok / dt / headline_tokens_per_s, 3),
                "decode_s_per_clip": round(dt / n_clips, 4)}
    finally:
        pipe.force_bar_tokens = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=180.0, help="clip length")
    ap.add_argument("--batch-clips", type=int, default=64, help="clips in the batch (all ranks together); each rank takes batch / N")
    ap.add_argument("--clips", type=int, default=0, help="clips per rank (overrides --batch-clips / N)")
    ap.add_argument("--attr-grid", type=int, default=27, help="attribute tuples per clip: 1 -> (1,1,1); 27 -> {0,1,2}^3")
    ap.add_argument("--ext-engines", type=int, default=int(os.environ.get("ETD_EXT_ENGINES", "0")),
                    help="extractor instances that transcribe different clips at the same time (own stream + host thread each); 0 = three beside up to three decoder engines, "
                         "two beside four (hardware queues).  Measured on the 64-clip extract stage: 2 -> 4 850-5 050, 3 -> 5 040-5 240, 4 -> 5 080-5 220 audio-s/s (round 6, gpurun r6_env: "
                         "three instead of two beside the three decoder engines is +3.9 % of the extract stage = +0.4 % of the step)")
    ap.add_argument("--engines", type=int, default=int(os.environ.get("ETD_ENGINES", "0")),
                    help="decoder engines (own HIP stream + KV cache each, driven from host threads); 0 = by the jobs per rank, whatever measured fastest: THREE engines "
                         "(576 streams each) from 1024 jobs up (and from 512 on a single GPU), four below (216 jobs per rank at N = 8: the latency-bound regime, where four "
                         "chains of short kernels overlap).  Measured on the round-5 tree, 64 clips, audio-s/s with 1 / 2 / 3 / 4 engines: 567 / 574 / 588 / 581 "
                         "(profiles/r05_engines.txt); the roofline of concurrent engines is reported on the UNION of their attention launches (roofline.frac)")
    ap.add_argument("--max-streams", type=int, default=2048, help="streams per engine (the fused decode step takes up to 2048 rows per launch)")
    ap.add_argument("--bar-tokens", type=int, default=48, help="tokens generated per bar (Bar_EOS does not end a bar)")
    ap.add_argument("--synthetic-bars", action="store_true", help="rounds 1-2 workload: ~8-notes/bar synthetic condition bars instead of the clip's own (A/B only)")
    ap.add_argument("--max-bars", type=int, default=0, help="diagnostics / profiling passes only: decode just the first N bars of every job (stated in config.workload)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("ETD_BENCH_BUDGET_S", "560")))
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--parity-clips", type=int, default=int(os.environ.get("ETD_PARITY_CLIPS", "64")),
                    help="extras.parity_mode: at most this many clips are taken through the chain in the exact-parity mode (fp32 extractor + fp32 decoder): the largest of "
                         "64 / 32 / 16 / 8 that the harness budget has room for (then fewer bars per job); 0 = skip")
    ap.add_argument("--parity-bars", type=int, default=0, help="extras.parity_mode: decode only the first N bars of every job (0 = by the time left)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the short HIP-event pass (PMC profiling runs: its eager launches would be counted with the step's)")
    ap.add_argument("--no-stamp", action="store_true", help="skip the stamped decode stage (roofline then comes from the serial event pass)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the N ranks ourselves, before anything in this process touches the GPU
    if (args.gpus > 1 or os.environ.get("ETD_FORCE_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a mislabelled number", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    if torch.cuda.device_count() < max(1, min(world, local + 1)):
        print(f"bench.py: rank {rank} needs GPU {local} but only {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        sys.exit(3)

    # Keep stdout clean for the ONE JSON line: RCCL prints its version banner to stdout from C code, so fd 1 is
    # pointed at stderr for the whole run and the result is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    use_dist = world > 1 or launched or os.environ.get("ETD_FORCE_DIST") == "1"      # a launcher-started single rank exercises the RCCL path too
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=dev)

    from etude_amd import _lib, parallel, synth
    from etude_amd.config import ExtractorConfig
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, PackedBars
    from etude_amd.extractor import AMTAPC_Extractor
    from etude_amd.pipeline import ClipBatchPipeline, ClipConditions, attr_grid, synthetic_tempo

    if args.clips > 0:
        clips = args.clips
    else:
        if args.batch_clips % world:
            print(f"bench.py: a batch of {args.batch_clips} clips does not split evenly over {world} ranks", file=sys.stderr)
            sys.exit(2)
        clips = args.batch_clips // world
    cfg = ExtractorConfig()
    wb = int(os.environ.get("ETD_WB", "4"))
    n_jobs0 = (args.clips if args.clips > 0 else args.batch_clips // world) * args.attr_grid
    # streams of the process: decoder engines + extractor instances + torch's default stream must stay within the 8 hardware queues (top of this file)
    want_eng = args.engines if args.engines > 0 else default_engines(n_jobs0, world)
    n_ext = args.ext_engines if args.ext_engines > 0 else (3 if want_eng <= 3 else 2)
    exs = [AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), dev, max_windows=wb) for _ in range(n_ext)]
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    grid = attr_grid(args.attr_grid)
    vocab = make_vocab()

    def build_engines(n_clips):
        n_jobs = n_clips * len(grid)
        want = args.engines if args.engines > 0 else default_engines(n_jobs, world)      # by measurement, not by the look of a per-launch figure
        n_eng = max(1, min(want, n_jobs))
        per_eng = min(args.max_streams, (n_jobs + n_eng - 1) // n_eng)
        # one batched-prefill pass carries up to 256 k prompt rows (~500 prompts at the 512-token truncation): a bar boundary of 1728 streams is then
        # four passes, and the host assembles / stages pass k + 1 while the GPU runs pass k (one 886 k-row pass left the queue empty for ~6 ms per bar)
        decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=per_eng, max_prefill_rows=min(262144, per_eng * 520))]
        decs += [decs[0].clone() for _ in range(n_eng - 1)]            # engines share one weight set (own KV caches and state)
        return decs, n_jobs, per_eng

    def my_clip_ids(n_clips):
        """global indices of this rank's clips: parallel.shard over the batch (rank r of R takes clips r, r + R, ...: tests/test_parallel_gloo.py), so that
        parallel.unshard of the gathered results restores the batch order whatever N is"""
        return parallel.shard(list(range(n_clips * world)), rank, world)

    def make_wavs(n_clips):
        # "64 clips as config 2 with seeds 0..63" -- clip c of the batch is synth.clip_audio's construction with seed c, evaluated on the
        # GPU (etude_amd/synth.py: clip_audio_device), resident in HBM before anything is timed
        return [synth.clip_audio_device(seed=ci, seconds=args.seconds, device=dev) for ci in my_clip_ids(n_clips)]

    decs, n_jobs, per_eng = build_engines(clips)
    wavs = make_wavs(clips)
    pipe = ClipBatchPipeline(exs, decs, vocab, synthetic_tempo(), grid, 44100, force_bar_tokens=args.bar_tokens)
    log(f"setup done: {clips} clip(s) on this rank, {n_jobs} decode jobs on {len(decs)} engine(s) x {per_eng} streams")

    def synthetic_conditions(conds):
        """rounds 1-2 workload (A/B): replace each clip's own bars by the synthetic ~8-notes/bar song"""
        out = []
        for c, cd in enumerate(conds):
            out.append(ClipConditions(cd.notes, cd.volume, PackedBars.from_lists(synth.song_bars(seed=1234 + my_clip_ids(clips)[c], n_bars=92)), cd.tokenizer))
        return out

    state = {}

    def step(max_bars=0):
        """One pass over this rank's clips.  Returns (t_extract, t_decode, t_notes, tokens)."""
        t0 = time.perf_counter()
        conds = pipe.extract_stage(wavs)
        if args.synthetic_bars:
            conds = synthetic_conditions(conds)
        t1 = time.perf_counter()
        results, stats = pipe.decode_stage(conds, max_bars=max_bars or args.max_bars)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        notes = pipe.notes_stage(conds, results)
        t3 = time.perf_counter()
        state.update(conds=conds, results=results, notes=notes)
        return t1 - t0, t2 - t1, t3 - t2, sum(s["tokens"] for s in stats)

    def agree(*vals):
        """the same decision on every rank: MAX over ranks"""
        if not use_dist:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    # ---- warm-up, with the harness budget in view (module docstring)
    warm_mode = "full step"
    shrunk_from = None
    batch64 = None
    # What follows the timed steps, in the order it is SHED when the run is late (time_left() below): extras.parity_mode (bars, then clips, then all of it), the extras
    # (configs[1..3], ~12 s), the serial event pass (~4 s), the stamped stage 24 -> 8 bars; the CPU baseline (~25 s) and an 8-bar stamped stage (~2 s) are what the
    # line cannot do without.  The batch per step shrinks only if K full steps + that minimum do not fit.
    OVERHEAD_S = 34.0            # the minimum: 8 stamped bars + the CPU baseline + JSON + teardown
    if args.warmup > 0:
        tw = time.perf_counter()
        a, b, c_, ntok1 = step()
        t_first, since = agree(time.perf_counter() - tw, since_process_start())
        log(f"first warm-up step: {t_first:.2f}s (extract {a:.2f} decode {b:.2f} notes {c_:.2f}); {since:.0f}s since the process started")
        rest = args.warmup - 1
        short = lambda t: 0.12 * t + 0.5                      # noqa: E731  (a 4-bar step: the whole extract stage + 4 of 92 bars)
        # full warm-up steps only if they leave room for EVERYTHING that follows the timed steps (24 stamped bars, serial pass, extras incl. the exact-parity pass,
        # CPU baseline: ~90 s); otherwise the warm-up steps shrink first -- they are untimed and everything is allocated, captured and cached after the first one
        FULL_POST_S = 0.27 * t_first + 4.0 + 14.0 + parity_cost(max(1, min(args.parity_clips, 32, clips)), 0, 92.0) + 16.0 + 11.0 + 30.0
        if since + (rest + args.steps) * t_first + max(OVERHEAD_S, FULL_POST_S if rest > 0 else 0.0) > args.budget_s:
            if since + args.steps * t_first + rest * short(t_first) + OVERHEAD_S <= args.budget_s:
                warm_mode = "first warm-up step full, the others 4 bars per job (W + K full steps exceed the harness budget)"
                for _ in range(rest):
                    step(max_bars=4)
                rest = 0
            elif clips > 8 and args.clips == 0:
                # K full steps of this batch do not fit: keep the measured full step as extras.batch64 and time the largest share that does
                batch64 = {"workload": f"the {clips}-clip share of the 64-clip batch on this rank, ONE full step (the first warm-up step)", "s_per_step": round(t_first, 3),
                           "audio_s_per_s": round(args.seconds * clips * world / t_first, 2), "decode_s": round(b, 3), "extract_s": round(a, 3),
                           "decoder_tokens_per_s": round(ntok1 / b, 1)}
                shrunk_from = clips
                new_clips = 8
                for cand in (32, 16, 8):
                    if cand < clips and since + (args.steps + 1) * t_first * cand / clips * 1.1 + rest * short(t_first * cand / clips) + OVERHEAD_S <= args.budget_s:
                        new_clips = cand
                        break
                log(f"{args.steps} steps of {t_first:.1f}s do not fit the {args.budget_s:.0f}s budget: timing {new_clips} clips per rank instead")
                for d in reversed(decs):
                    d.close()
                pipe.close()
                del wavs[new_clips:]
                clips = new_clips
                decs, n_jobs, per_eng = build_engines(clips)
                pipe = ClipBatchPipeline(exs, decs, vocab, synthetic_tempo(), grid, 44100, force_bar_tokens=args.bar_tokens)
                step()
                warm_mode = "one full step of the shrunk batch, the others 4 bars per job"
                for _ in range(max(0, rest - 1)):
                    step(max_bars=4)
                rest = 0
        for _ in range(rest):
            step()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for d in decs:
        d.stats_reset()
    barrier()
    t0 = time.perf_counter()
    t_ext = t_dec = t_notes = 0.0
    n_tok = 0
    for i in range(args.steps):
        a, b, c_, d_ = step()
        t_ext += a; t_dec += b; t_notes += c_; n_tok += d_
        log(f"timed step {i + 1}/{args.steps}: {a + b + c_:.2f}s (extract {a:.2f} decode {b:.2f} notes {c_:.2f})")      # (a line per step: harnesses take minutes of silence for a hang)
    results = state["results"]
    gathered_jobs = len(results)
    # digest of every token this rank generated in the last timed step (job order): two builds / switches whose kernels must be
    # equivalent print the same value under the real four-engine load
    tok_digest = hashlib.sha256(np.concatenate([r[0] for r in results]).astype(np.int32).tobytes()).hexdigest()[:16]
    per_rank_ids = [[np.asarray(r[0], np.int32) for r in results]]
    if use_dist:
        # the path's only exchange: ONE final gather of the small variable-length results (token ids of every job)
        per_rank_ids = parallel.gather_int_arrays(per_rank_ids[0], device=dev, force=True)
        gathered_jobs = sum(len(x) for x in per_rank_ids)
    barrier()
    elapsed = time.perf_counter() - t0
    # digest of the WHOLE batch in global clip order (clip c = its 27 jobs in tuple order): parallel.unshard of the per-rank clip lists.  A job's ids do not
    # depend on which jobs share its launches (the batch-invariance the GPU suite asserts), so an N-rank run must print the value the 1-rank run prints.
    tok_digest_all = parallel.digest_in_global_clip_order(per_rank_ids, len(grid))
    timed_stats = [d.stats() for d in decs]

    tmax = torch.tensor([elapsed, t_ext, t_dec, t_notes], dtype=torch.float64, device=dev)
    tsum = torch.tensor([float(n_tok), sum(s["kv_bytes"] + s["steps"] * s["weight_bytes_per_step"] for s in timed_stats)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    elapsed, t_ext, t_dec, t_notes = [float(x) for x in tmax.tolist()]
    n_tok_all, dec_bytes_all = [float(x) for x in tsum.tolist()]

    conds = state["conds"]
    nbars = [len(cd.bars) for cd in conds]
    xlen = float(np.mean([cd.bars.ids.size / max(1, len(cd.bars)) for cd in conds]))
    audio_s = args.seconds * clips * args.steps * world
    result = {
        "metric": "audio-sec/s transcribed + decoder tokens/s, 3-min clip batch",
        "value": round(audio_s / elapsed, 3), "unit": "audio-s/s (each clip extracted, tokenized and decoded for every attribute tuple)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "strong" if args.clips == 0 else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[4], the batch north_star names: {clips * world} x 3-min 44.1 kHz stereo clips ({clips} per rank), per clip the chain of infer.py "
                                f"(extract wav->notes, volume contour, tokenizer on the synthetic tempo.json -> the clip's OWN condition bars: {int(np.mean(nbars))} bars of ~{xlen:.0f} ids), "
                                f"greedy generate() for {clips}x{args.attr_grid} (clip, attribute tuple) jobs per rank with {args.bar_tokens} generated tokens per bar (Bar_EOS does not end a bar: "
                                "synthetic weights carry no musical EOS statistics), overlap bin 2, decode_to_notes with the clip's volume map; 16-bit operands / fp32 accumulate -- IEEE half (f16), which on MI355X has the storage, the layout and the MFMA rate of the bf16 that BASELINE.json's configs name and 3 more significant bits (extractor 5x closer to the fp32 reference, greedy bars 5x less divergent: profiles/r05_ext_f16.txt, r05_dec_f16.txt); seeded synthetic weights"
                                + (f" -- ONLY THE FIRST {args.max_bars} BARS of every job are decoded (--max-bars: a profiling pass, not a throughput figure)" if args.max_bars else "")
                                + (" -- CONDITION BARS REPLACED by the synthetic ~8-notes/bar song (--synthetic-bars, A/B)" if args.synthetic_bars else "")
                                + (f" -- BATCH SHRUNK from {shrunk_from} to {clips} clips per rank to fit the harness budget of {args.budget_s:.0f}s (the full batch, one step: extras.batch64)" if shrunk_from else "")),
                   "batch_clips": clips * world, "clips_per_gpu": clips, "attr_tuples_per_clip": args.attr_grid, "decode_jobs_per_gpu": n_jobs, "decoder_streams_per_engine": per_eng,
                   "decoder_engines": len(decs), "extractor_engines": len(exs), "clip_seconds": args.seconds,
                   "windows_per_clip": int(np.ceil((1 + int(np.ceil(160 * wavs[0].shape[1] / 441)) // 256) / 512)),
                   "bars": int(np.mean(nbars)), "condition_ids_per_bar": round(xlen, 1), "bar_tokens": args.bar_tokens, "parallelism": f"clip-sharded x{world}",
                   "stage_order": "extract stage for all clips, then decode stage for all jobs, then notes", "warmup_step": warm_mode},
        "extract_audio_s_per_s": round(audio_s / t_ext, 2),
        "decoder_tokens_per_s": round(n_tok_all / t_dec, 2),
        "notes_stage_s_per_step": round(t_notes / args.steps, 4),
        "decoder_tokens_per_step": n_tok / args.steps, "notes_per_clip": float(np.mean([cd.notes.size for cd in conds])),
        "cover_notes_per_job": float(np.mean([n.size for n in state["notes"]])), "jobs_gathered": gathered_jobs, "tokens_sha256_rank0": tok_digest, "tokens_sha256_all": tok_digest_all,
        "build_id": _lib.lib().etd_build_id().decode(),
    }

    time_left = lambda: args.budget_s - since_process_start()                       # noqa: E731  (the ONE JSON line matters more than its optional parts)
    CPU_RESERVE_S = 30.0 if (rank == 0 and world == 1 and not args.no_cpu_baseline) else 4.0
    late = lambda margin: time_left() < margin                                       # noqa: E731
    # ---- roofline of the dominant kernel, in the timed configuration: one more decode stage over the same conditions with every
    # engine stamping its k_dstep_attn_down launches on the device (first workgroup's start .. last workgroup's end)
    roof = {"kernel": "k_dstep_attn_down", "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "traffic": None}
    if not args.no_stamp:
        # short runs (profiling: tools/profile.sh) stamp every bar of the extra stage, the harness's 25-step run 24 bars (its budget is tight)
        step_dec_s = t_dec / max(1, args.steps)
        nb_mean = max(1.0, float(np.mean(nbars)))
        stamp_bars = args.max_bars or (0 if args.steps + args.warmup <= 4 and time_left() > CPU_RESERVE_S + step_dec_s + 30.0 else
                                       (24 if time_left() > CPU_RESERVE_S + step_dec_s * 24 / nb_mean + 22.0 else 8))
        skip_bars = 4 if (stamp_bars == 0 or stamp_bars >= 8) else 0    # the bars in which the 4-pair history (and with it the context) is still growing
        for d in decs:
            d.stamp(True, skip_steps=skip_bars * (args.bar_tokens - 1))
            d.stats_reset()
        pipe.decode_stage(conds, max_bars=stamp_bars)
        torch.cuda.synchronize(dev)
        st = [d.stats() for d in decs]
        for d in decs:
            d.stamp(False)
        launches = sum(s["stamped_launches"] for s in st)
        secs = sum(s["stamped_seconds"] for s in st)
        byts = sum(s["stamped_alg_bytes"] for s in st)
        # concurrent engines share the HBM: a launch of one engine then reads its bytes at a fraction of the peak by construction, and the statistic that says how well
        # the memory system is used is bytes of ALL launches / the time during which ANY of them was running -- the union of the per-launch (start, end) stamps of
        # every engine (one device clock).  With one engine the union is the sum and both figures coincide.
        logs = [d.stamp_log() for d in decs]
        iv = np.concatenate([l for l in logs if len(l)]) if any(len(l) for l in logs) else np.zeros((0, 2), np.uint64)
        union_s = union_seconds(iv)
        if launches > 0 and secs > 0:
            per_launch = byts / secs / 1e9
            # (a log holds the first 131 072 launches of an engine: should a run exceed that, the union covers the logged share and so do the bytes)
            ach = byts * (len(iv) / launches) / union_s / 1e9 if union_s > 0 else per_launch
            roof.update(achieved=round(ach, 1), frac=round(ach / PEAK_HBM_GBS, 4), launches=int(launches), avg_launch_ms=round(1e3 * secs / launches, 5),
                        alg_bytes_per_launch=byts / launches, engines=len(decs), launches_in_union=int(len(iv)),
                        frac_per_launch=round(per_launch / PEAK_HBM_GBS, 4), union_busy_s=round(union_s, 4), sum_of_spans_s=round(secs, 4),
                        frac_source=("device stamps (etd_decoder_stamp / etd_decoder_stamp_log): s_memrealtime of the first workgroup's start and the last workgroup's end of EVERY k_dstep_attn_down launch "
                                     f"of one extra decode stage over the same jobs (bars {skip_bars} .. {(stamp_bars or int(np.mean(nbars))) - 1}: the steady-state prompt size, 96 % of a job's bars) with all {len(decs)} "
                                     "engine(s) running.  frac = algorithmic bytes of all stamped launches / the UNION of their (start, end) intervals across engines (one device clock) / peak: what the "
                                     "memory system delivers while any attention launch runs; frac_per_launch = the same bytes / the SUM of the spans (a launch that shares the HBM with another engine's "
                                     "reads its bytes at a fraction of the peak by construction; with one engine the two coincide -- and that figure is what a rocprofv3 kernel trace averages); "
                                     "algorithmic bytes = K+V rows of every (row, head) context + the down / dense weights a launch streams, counted exactly by the library"))
    result["roofline"] = roof
    tp = ROOT / "profiles" / "traffic.json"
    if tp.exists():
        try:
            tj = json.loads(tp.read_text())
            sj = tj.get("k_dstep_attn_down_steady")
            if sj and roof.get("alg_bytes_per_launch"):
                # PMC bytes of the SAME attention form at the SAME rows per launch and contexts as the stamped launches (steady-state bars), so traffic / alg_bytes_per_launch
                # reads directly; a table measured on another launch shape (other engine count / batch) is not this run's traffic and is left out
                if sj.get("rows_per_launch") == per_eng and sj.get("engines") == len(decs):
                    roof["traffic"] = sj["bytes_per_launch"]
                    roof["traffic_over_algorithmic"] = round(sj["bytes_per_launch"] / roof["alg_bytes_per_launch"], 4)
                    roof["traffic_source"] = ("static: profiles/traffic.json -- rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch of " + sj["form"] + f" ({sj['engines']} engine(s) x "
                                              f"{sj['rows_per_launch']} rows, library build {sj.get('build_id', '?')}) over the steady-state bars (4..7) of separate 8-bar profiling passes of this "
                                              "command, committed with the tree; NOT measured in this run")
                else:
                    roof["traffic_source"] = (f"none: profiles/traffic.json holds {sj.get('engines', '?')} engine(s) x {sj.get('rows_per_launch', '?')} rows per launch, this run "
                                              f"{len(decs)} x {per_eng} (tools/profile.sh measures the default layout)")
        except Exception:
            pass
    # the whole decode stage as ONE figure that needs no per-kernel timing: SURVEY 8(d)'s step bytes (weights once per engine-step + K/V of
    # every row's context over all layers), exact counts of the TIMED steps, over the decode stage's wall time in those steps
    if t_dec > 0:
        roof["decode_stage"] = {"alg_bytes_per_step": dec_bytes_all / args.steps, "stage_s_per_step": round(t_dec / args.steps, 4),
                                "achieved": round(dec_bytes_all / t_dec / 1e9, 1), "unit": "GB/s", "frac": round(dec_bytes_all / t_dec / 1e9 / PEAK_HBM_GBS, 4),
                                "note": "algorithmic decode-step bytes (SURVEY 8d: W per engine-step + 16 KiB x ctx per row-step) of the timed steps / wall time of their decode stage "
                                        "(the batched prefill of every bar is in the time, not in the bytes)"}

    # ---- per-kernel HIP-event breakdown: a SHORT serial pass (4 bars per job, stages and engines one at a time, launches eager with an
    # event pair each -- event records cannot sit inside hipGraph replays).  Indicative: serial durations, not the timed configuration's.
    try:
        if args.no_serial_pass:
            raise RuntimeError("skipped (--no-serial-pass)")
        if late(CPU_RESERVE_S + 8.0):
            raise RuntimeError("skipped: the run is close to its harness budget")
        _lib.prof_reset()
        _lib.prof_enable(True)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(pipe.ex_streams[0]):
            pipe.conditions_of(wavs[0], 0, 0)
        pipe.ex_streams[0].synchronize()
        pipe.decode_stage(conds, max_bars=4, one_at_a_time=True)
        torch.cuda.synchronize(dev)
        _lib.prof_enable(False)
        prof = _lib.prof_report()
        result["kernel_ms_serial_pass"] = {"what": "one clip's extract + 4 bars of every decode job, engines one after the other, HIP events around every launch",
                                           "ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                                           "launches": {k: v["launches"] for k, v in prof.items()}}
        p = prof.get("k_dstep_attn_down")
        if p and p["ms"] > 0:
            roof["frac_serial"] = round(p["bytes"] / (p["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
            if "frac" not in roof:
                roof.update(achieved=round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1), frac=roof["frac_serial"], launches=p["launches"], avg_launch_ms=round(p["ms"] / p["launches"], 5),
                            alg_bytes_per_launch=p["bytes"] / p["launches"], frac_source="serial event pass (no device stamps in this run)")
    except Exception as e:      # a diagnostics failure must not take the headline down
        _lib.prof_enable(False)
        result["kernel_ms_serial_pass"] = {"error": repr(e)}
    if batch64:
        result.setdefault("extras", {})["batch64"] = batch64

    # ---- extras outside the timed region
    if not args.no_extras and rank == 0 and late(CPU_RESERVE_S + 16.0):
        result.setdefault("extras", {})["skipped"] = "the run is close to its harness budget"
    elif not args.no_extras and rank == 0:
        extras = result.setdefault("extras", {})
        ex = exs[0]
        try:
            xs = torch.from_numpy(synth.window_features(5, 16)).to(dev)          # configs[2]: 16 windows
            ex.transcript_windows(xs)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            reps = 3
            for _ in range(reps):
                ex.transcript_windows(xs)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t) / (reps * 16)
            tf = ex.window_flops / dt / 1e12
            extras["extractor_only"] = {"workload": "BASELINE configs[2]: 16 x 512-frame windows (8.192 s each), hFT-Transformer only",
                                        "ms_per_window": round(dt * 1e3, 3), "audio_s_per_s": round(8.192 / dt, 1),
                                        "alg_gflop_per_window": round(ex.window_flops / 1e9, 1),
                                        "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)}}
        except Exception as e:      # extras must never take the headline down
            extras["extractor_only"] = {"error": repr(e)}
        try:
            # configs[1]: ONE 3-min clip, the whole chain with the default attributes (1/1/1, overlap 2) on one engine
            one = ClipBatchPipeline(exs[:1], decs[:1], vocab, synthetic_tempo(), attr_grid(1), 44100, force_bar_tokens=args.bar_tokens)
            torch.cuda.synchronize(dev)
            r1 = one.run(wavs[:1])
            extras["single_clip"] = {"workload": f"BASELINE configs[1]: one 3-min 44.1 kHz clip, extract (wav -> {r1['conditions'][0].notes.size} notes) + tokenize + greedy decode of "
                                                 f"{len(r1['conditions'][0].bars)} bars x {args.bar_tokens} tokens + notes, attributes 1/1/1, f16 operands",
                                     "extract_s": round(r1["t_extract"], 4), "decode_s": round(r1["t_decode"], 4), "notes_s": round(r1["t_notes"], 4),
                                     "wall_s": round(r1["t_extract"] + r1["t_decode"] + r1["t_notes"], 4),
                                     "audio_s_per_s": round(args.seconds / (r1["t_extract"] + r1["t_decode"] + r1["t_notes"]), 1),
                                     "decoder_tokens_per_s": round(r1["tokens"] / r1["t_decode"], 1)}
            one.close()
        except Exception as e:
            extras["single_clip"] = {"error": repr(e)}
        for key, c0 in (("decoder_streams", 512), ("decoder_streams_4k", 3500)):     # reference-faithful context / 4k stress
            try:
                extras[key] = decoder_stream_bench(dcfg, dev, ctx0=c0, streams=[d._ts for d in decs])
            except Exception as e:
                extras[key] = {"error": repr(e)}

    # ---- extras.parity_mode: the cost of exact parity (fp32 extractor + fp32 decoder) on the LARGEST of 64 / 32 / 16 / 8 clips the time left allows (one engine x 27 * clips streams)
    oracle_check = None
    nb_mean = float(np.mean(nbars))
    RAGGED_S, N8_S, BT144_S = 16.0, 11.0, 6.0          # what the extras behind this one need (reserved while sizing it)
    if not args.no_extras and rank == 0 and args.parity_clips > 0:
        extras = result.setdefault("extras", {})
        want = [c for c in (64, 32, 16, 8, 4, 2, 1) if c <= min(args.parity_clips, len(wavs))] or [min(args.parity_clips, len(wavs))]
        cands = [(want[0], args.parity_bars)] if args.parity_bars else [(c, 0) for c in want] + [(want[-1], 24), (want[-1], 8)]
        room = lambda: time_left() - CPU_RESERVE_S - (RAGGED_S + N8_S if world == 1 else 0.0) - 5.0      # noqa: E731
        pick = next(((c, b) for c, b in cands if parity_cost(c, b, nb_mean) < room()), None)
        if pick is None:
            extras["parity_mode"] = {"skipped": f"{time_left():.0f} s of the harness budget left: not enough for the smallest exact-parity pass ({parity_cost(*cands[-1], nb_mean):.0f} s estimated); "
                                                "python bench.py --steps 1 --warmup 1 runs it whole (profiles/)"}
        else:
            try:
                log(f"extras.parity_mode: {pick[0]} clips" + (f", {pick[1]} bars" if pick[1] else "") + f" (estimated {parity_cost(*pick, nb_mean):.0f} s, {time_left():.0f} s left)")
                extras["parity_mode"], oracle_check = parity_mode_extras(args, dev, wavs, grid, vocab, pick[0], pick[1], decs, lambda: time_left() - CPU_RESERVE_S,
                                                                               headline_decode_s_per_clip=(t_dec / max(1, args.steps) / clips) if not args.max_bars else None)
            except Exception as e:      # extras must never take the headline down
                extras["parity_mode"] = {"error": repr(e)}

    # ---- extras.ragged_bars: the reference's own stopping rule (Bar_EOS ends a bar; 512 / 25 600 limits) under continuous batching, + the reference's ids for clip_ctx.npz
    if not args.no_extras and rank == 0 and world == 1 and clips >= 8 and time_left() > CPU_RESERVE_S + N8_S + RAGGED_S:
        try:
            log(f"extras.ragged_bars ({time_left():.0f} s left)")
            result.setdefault("extras", {})["ragged_bars"] = ragged_bars_extras(args, dev, conds, grid, vocab, lambda: time_left() - CPU_RESERVE_S - N8_S)
        except Exception as e:      # extras must never take the headline down
            result.setdefault("extras", {})["ragged_bars"] = {"error": repr(e)}

    # ---- extras.n8_share: the per-rank batch of the N = 8 run, measured on this GPU (the scaling expectation the N = 1 line carries)
    if not args.no_extras and rank == 0 and world == 1 and clips >= 8 and time_left() > CPU_RESERVE_S + N8_S:
        try:
            log(f"extras.n8_share ({time_left():.0f} s left)")
            result.setdefault("extras", {})["n8_share"] = n8_share_extras(args, dev, exs, wavs, grid, vocab, result["value"])
        except Exception as e:      # extras must never take the headline down
            result.setdefault("extras", {})["n8_share"] = {"error": repr(e)}

    # ---- extras.bar_tokens_144: the headline's sensitivity to the 48 forced tokens per bar (as many clips as the time left allows; a 64-clip decode stage at 144 tokens is ~45 s)
    if not args.no_extras and rank == 0 and world == 1 and args.bar_tokens != 144:
        per_clip = 3.2 * (t_dec / max(1, args.steps)) / max(1, clips)          # ~3 x the tokens at longer contexts
        nc = next((c for c in (64, 32, 16, 8) if c <= clips and 1.15 * c * per_clip + 3.0 < time_left() - CPU_RESERVE_S), 0)
        if nc:
            try:
                log(f"extras.bar_tokens_144: {nc} clips ({time_left():.0f} s left)")
                result.setdefault("extras", {})["bar_tokens_144"] = bar_tokens_sensitivity(args, dev, pipe, conds, nc, 144, result["decoder_tokens_per_s"])
            except Exception as e:      # extras must never take the headline down
                result.setdefault("extras", {})["bar_tokens_144"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and late(6.0):
        result["cpu_baseline"] = {"skipped": "the run is within 6 s of its harness budget"}
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            # the oracle decodes the first five condition bars of clip 0 with the default tuple: the exact-parity pass's bars (the fp32 extractor's) when it ran -- its ids for the
            # same job are then compared with the oracle's -- else the headline's
            if oracle_check is not None:
                bars0 = oracle_check["bars"]
            else:
                cd0 = conds[0]
                bars0 = [cd0.bars.bar(i) for i in range(min(5, len(cd0.bars)))]
                if args.synthetic_bars:
                    bars0 = synth.song_bars(seed=1234, n_bars=5)
            cb = cpu_baseline(bars0, clip_seconds=args.seconds, windows_per_clip=result["config"]["windows_per_clip"], attr_tuples=args.attr_grid,
                              bars_per_job=int(np.mean(nbars)), bar_tokens=args.bar_tokens)
            oracle_ids = cb.pop("_oracle_ids", None)
            result["cpu_baseline"] = cb
            # ---- the CHECKER leg: the oracle's greedy ids of those bars against the engines' ids for the same clip / tuple / bars
            if oracle_ids is not None and not args.no_extras:
                cmp_ = lambda got: sum(int(list(a_) == list(b_)) for a_, b_ in zip(got, oracle_ids))      # noqa: E731
                chk = {"bars": len(oracle_ids), "tokens_per_bar": args.bar_tokens,
                       "what": "cpu_baseline()'s oracle bars (oracle.neox: bars 0-3 through generate_ids, bar 4 -- the first with a full 4-pair history, prompt at the 512-token truncation -- "
                               "through forward_logits) against the ids the device engines generated for the SAME clip 0 / tuple (1,1,1,2) / condition bars"}
                if oracle_check is not None:
                    chk["oracle_bars_identical"] = f"{cmp_(oracle_check['fp32'])}/{len(oracle_ids)}"
                    chk["fp32_engine"] = oracle_check["where"]
                    if "f16" in oracle_check:
                        chk["oracle_bars_identical_f16"] = f"{cmp_(oracle_check['f16'])}/{len(oracle_ids)}"
                    result.setdefault("extras", {}).setdefault("parity_mode", {})["oracle_check"] = chk
                else:
                    # no exact-parity pass in this run: a one-stream fp32 engine on the headline's bars (not the wide launch shape; stated)
                    from etude_amd.decoder import EtudeDecoder as _ED
                    d1 = _ED(dcfg, synth.decoder_state_dict(1, {}), dev, precision="fp32", max_streams=1)
                    try:
                        got = d1.generate_many([(bars0, [synth.attrs()] * len(bars0))], vocab, force_bar_tokens=args.bar_tokens)[0]
                    finally:
                        d1.close()
                    chk["oracle_bars_identical"] = f"{cmp_(got)}/{len(oracle_ids)}"
                    chk["fp32_engine"] = "a ONE-stream fp32 engine on the headline's (16-bit extractor's) bars: the exact-parity batch pass did not fit this run"
                    if not args.synthetic_bars and len(results) > DEFAULT_TUPLE_INDEX:
                        chk["oracle_bars_identical_f16"] = f"{cmp_(first_bars(*results[DEFAULT_TUPLE_INDEX], len(oracle_ids)))}/{len(oracle_ids)}"
                    result.setdefault("extras", {})["oracle_check"] = chk
        except Exception as e:
            result["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    log("done")
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def decoder_stream_bench(dcfg, dev, n_streams: int = 128, ctx0: int = 512, steps: int = 512, engines: int = 2, streams=None):
    """BASELINE configs[3]: 128 concurrent streams, each prefilled to ctx0 then `steps` (512: SURVEY 8(d) config 4) greedy decode steps in a
    4 096-position KV ring (EOS suppressed so every stream runs the full length -- throughput does not depend on the token values).
    The streams are dealt over `engines` decoder engines (own stream, KV cache and captured graphs, shared weights) that
    step concurrently from one host thread each, like the headline's decode stage: one step of the figure below = every one
    of the n_streams streams advanced by one token.
    `streams`: torch streams the engines run on.  Inside bench.py these are the headline engines' own (idle by then): a process
    that already holds seven streams gets hardware queues for two NEW ones that may share a compute pipe, and two dependent
    kernel chains on one pipe run one after the other (0.50 instead of 0.34 ms per step at ctx 512, LABNOTES.md)."""
    import threading
    import torch
    from etude_amd import _lib, synth
    from etude_amd.decoder import EtudeDecoder
    engines = max(1, min(engines, n_streams))
    per = [n_streams // engines + (1 if e < n_streams % engines else 0) for e in range(engines)]
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="f16", max_streams=max(per), max_ctx=4096)]
    while len(decs) < engines:
        decs.append(decs[0].clone())
    if streams:
        for dec, ts in zip(decs, streams):
            dec._ts = ts
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    tg = np.asarray([2, 1, 1, 1], np.int32)
    slots = [np.arange(n, dtype=np.int32) for n in per]
    for e, dec in enumerate(decs):
        st = dec._stream()
        grp = max(1, dec.max_prefill_rows // ctx0)                # prompts per batched prefill pass
        for s0 in range(0, per[e], grp):
            n = min(grp, per[e] - s0)
            ids = rng.integers(6, 154, n * ctx0).astype(np.int32)
            cls = rng.integers(1, 3, n * ctx0).astype(np.int32)
            a4 = rng.integers(0, 3, (4, n * ctx0)).astype(np.int32)
            T = np.full(n, ctx0, np.int32); tgt = np.ascontiguousarray(np.tile(tg, (n, 1))); eos = np.full(n, -1, np.int32)
            lim = np.full(n, min(1000, 4096 - ctx0), np.int32); sl = np.ascontiguousarray(slots[e][s0: s0 + n])
            _lib.check(lib.etd_decoder_begin_bars(dec._h, n, sl.ctypes.data, T.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, tgt.ctypes.data,
                                                  eos.ctypes.data, lim.ctypes.data, st), "begin_bars")
        _lib.check(lib.etd_decoder_step(dec._h, slots[e].ctypes.data, per[e], 4, st), "step")
    torch.cuda.synchronize(dev)
    errs = []
    for dec in decs:
        dec.stats_reset()

    gate = threading.Barrier(engines + 1, timeout=120)

    def run(e):
        try:
            torch.cuda.set_device(dev)
            gate.wait()
            _lib.check(lib.etd_decoder_step(decs[e]._h, slots[e].ctypes.data, per[e], steps, decs[e]._stream()), "step")     # hipGraph replays
            decs[e]._ts.synchronize()
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
    th = [threading.Thread(target=run, args=(e,)) for e in range(engines)]
    for x in th:
        x.start()
    gate.wait()                                          # the engine threads exist and are bound to the device: start the clock
    t = time.perf_counter()
    for x in th:
        x.join()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t
    if errs:
        raise errs[0]
    st = [dec.stats() for dec in decs]
    bytes_all = sum(s["kv_bytes"] + s["steps"] * s["weight_bytes_per_step"] for s in st)     # exact: every engine streams the weight set once per step
    gbs = bytes_all / dt / 1e9
    out = {"workload": f"BASELINE configs[3]: {n_streams} streams on {engines} engine(s), f16 weights+KV, ctx {ctx0 + 4}->{ctx0 + 4 + steps}, {steps} greedy steps, EOS suppressed",
           "engines": engines, "tokens_per_s": round(n_streams * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4),
           "alg_bytes_per_step": bytes_all / steps,
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}}
    for dec in reversed(decs):
        dec.close()
    return out


if __name__ == "__main__":
    main()

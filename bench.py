#!/usr/bin/env python3
"""Benchmark of Etude's two hot paths on MI355X: audio-seconds/s transcribed + decoder tokens/s.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = each rank's share of BASELINE.json configs[4] (the configuration the metric is quoted on: "3-min clip
batch", 64 clips sharded 8 per GPU, attribute grid {0,1,2}^3).  Per rank: --clips (8) 3-minute 44.1 kHz stereo
clips, already resident in HBM, go through the whole Extract hot path (channel mean, resample, STFT/log-mel,
hFT-Transformer over 22 windows, D2H of the frame outputs, note picking -> the note list extract() writes); then
the Decode hot path generates a cover for each (clip, attribute tuple) job -- --attr-grid (27) tuples per clip,
~92 condition bars each -- as concurrent device streams (continuous batching on four decoder engines), greedy.
Ranks work on different clips (seed 1234 + clip index) with no data-path collective: weak scaling; at N=8 the
job is exactly configs[4].  `--clips 1 --attr-grid 1` is configs[1] (one clip, attributes 1/1/1).

value = audio seconds taken through BOTH stages per wall second, whole job (all ranks).  The per-stage
numbers the metric names are reported next to it (extract_audio_s_per_s, decoder_tokens_per_s), plus
  roofline      the dominant kernel of the timed region (HIP-event timed inside the library)
  cpu_baseline  the CPU oracle timed on this node's host cores on a bounded sample (rank 0, N=1 only)
  extras        extractor-only (configs[2]) and 128-stream decoder (configs[3]) measurements taken OUTSIDE the
                timed region, each with its own roofline fraction.
Weights are seeded synthetic tensors of the reference's architecture (no checkpoints / network); the
condition bars are the synthetic ~8-notes/bar song of SURVEY.md 8(d) config 1 because the Structuralize
stage and the tokenizer are outside the hot path.
"""
from __future__ import annotations

import argparse
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
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0         # HBM3E spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_vocab():
    from etude_amd import synth
    from etude_amd.vocab import Vocab
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def cpu_baseline(seconds_budget: float = 25.0, clip_seconds: float = 180.0, windows_per_clip: int = 22, attr_tuples: int = 27, bars_per_job: int = 92,
                 bar_tokens: int = 48):
    """The CPU oracle (a restatement of the reference, pinned by golden vectors) on this node's host cores."""
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
    bars = synth.song_bars(seed=1234, n_bars=2)
    t0 = time.time()
    # synthetic weights reach Bar_EOS after ~7 tokens; id 5 (Bar_EOS) is pushed out of reach so that both bars run the full 24
    # tokens, like the GPU workload (Bar_EOS suppressed)
    tsd = dict(tsd); tsd["lm_head.weight"] = tsd["lm_head.weight"].clone(); tsd["lm_head.weight"][5] = 0
    out = neox.generate_ids(tsd, neox.NeoxDims(), 4, 5, bars, [synth.attrs()] * 2, max_bar_token_limit=24)
    t_dec = time.time() - t0
    ntok = sum(len(b) - 1 for b in out)
    # the headline's unit: audio seconds taken through BOTH stages (extract once, decode for every attribute tuple) per wall second.
    # Extrapolated from the bounded sample: extract = windows_per_clip x (time per window); decode = tuples x bars x (time per bar of
    # bar_tokens tokens, prompt pass included).
    ext_s_per_clip = (t_ext / nwin) * windows_per_clip
    dec_s_per_clip = attr_tuples * bars_per_job * (t_dec / 2.0) * (bar_tokens / 24.0)
    return {"value": round(clip_seconds / (ext_s_per_clip + dec_s_per_clip), 4),
            "unit": "audio-s/s (each clip extracted and decoded for every attribute tuple; extrapolated from the sample)", "cores": cores, "kind": "port",
            "sample": f"oracle hFT forward on {nwin} window(s) of 512 frames (8.192 s audio each) in {t_ext:.1f}s on {cores} threads; "
                      f"oracle greedy generate on 2 bars of 24 tokens ({ntok} tokens) in {t_dec:.1f}s on {dthreads} threads",
            "extract_audio_s_per_s": round(nwin * 8.192 / t_ext, 4),
            "decoder_tokens_per_s": round(ntok / t_dec, 2), "decoder_cores": dthreads}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=180.0, help="clip length")
    ap.add_argument("--clips", type=int, default=8, help="clips per rank")
    ap.add_argument("--attr-grid", type=int, default=27, help="attribute tuples per clip: 1 -> (1,1,1); 27 -> {0,1,2}^3")
    ap.add_argument("--streams", type=int, default=256, help="concurrent decoder streams (capped at the number of jobs)")
    ap.add_argument("--ext-engines", type=int, default=int(os.environ.get("ETD_EXT_ENGINES", "2")),
                    help="extractor instances that transcribe different clips at the same time (own stream + host thread each)")
    ap.add_argument("--engines", type=int, default=int(os.environ.get("ETD_ENGINES", "4")),
                    help="independent decoder engines (own HIP stream + KV cache each) driven from host threads: the short dependent kernels of one engine's decode step overlap the other's")
    ap.add_argument("--bars", type=int, default=92)
    ap.add_argument("--bar-tokens", type=int, default=48, help="tokens generated per bar (Bar_EOS suppressed)")
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap the stages across clips (jobs of clip c are admitted when its extraction is done) instead of running them back to back; "
                         "measured 4 %% slower on one GPU: both stages are GPU-bound, the overlap only adds contention and a ragged ramp-up")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the N ranks ourselves.  Nothing in this process has touched the GPU yet
    # (torch.cuda.device_count() does not initialise HIP), the ranks are fresh child processes of torch.distributed.run, their
    # stdout (rank 0's ONE JSON line) is relayed and their exit status becomes ours.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a mislabelled number", file=sys.stderr)
        sys.exit(2)
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
    use_dist = world > 1 or os.environ.get("ETD_FORCE_DIST") == "1"      # ETD_FORCE_DIST: exercise the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=dev)

    from etude_amd import _lib, parallel, synth
    from etude_amd.config import ExtractorConfig
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from etude_amd.extractor import AMTAPC_Extractor

    cfg = ExtractorConfig()
    ex = AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), dev, max_windows=int(os.environ.get("ETD_WB", "4")))
    exs = [ex] + [AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), dev, max_windows=int(os.environ.get("ETD_WB", "4"))) for _ in range(max(1, args.ext_engines) - 1)]
    ex_streams = [torch.cuda.Stream(device=dev) for _ in exs]
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    n_jobs = args.clips * args.attr_grid
    n_eng = max(1, min(args.engines, n_jobs))
    per_eng = (min(args.streams, n_jobs) + n_eng - 1) // n_eng
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="bf16", max_streams=per_eng)]
    decs += [decs[0].clone() for _ in range(n_eng - 1)]            # engines share one weight set (own KV caches and state)
    vocab = make_vocab()
    clip_ids = [rank * args.clips + c for c in range(args.clips)]                      # global clip index = rank-major shard
    base = synth.clip_audio(seed=1234, seconds=args.seconds)
    wavs = []
    for ci in clip_ids:                                                                 # distinct clips, resident in HBM
        rng = np.random.default_rng(1234 + ci)
        w = np.roll(base, int(rng.integers(0, base.shape[1])), axis=1) * np.float32(rng.uniform(0.6, 1.0))
        wavs.append(torch.from_numpy(np.ascontiguousarray(w)).to(dev))
    if args.attr_grid == 1:
        grid = [(1, 1, 1)]
    else:
        grid = [(p, r, s_) for p in range(3) for r in range(3) for s_ in range(3)][: args.attr_grid]
    jobs = []
    for ci in clip_ids:
        bars = synth.song_bars(seed=1234 + ci, n_bars=args.bars)
        for (p, r, s_) in grid:
            jobs.append((bars, [synth.attrs(p, r, s_, 2)] * len(bars)))
    inf = cfg.infer

    # the extractor stays on torch's default stream: engines + default must fit the runtime's hardware queues (4 unless
    # GPU_MAX_HW_QUEUES says otherwise) or streams start sharing a queue and serialise against each other
    ext_stream = torch.cuda.Stream(device=dev) if os.environ.get("ETD_EXT_STREAM") == "1" else torch.cuda.default_stream(dev)
    job_clip = [k // len(grid) for k in range(n_jobs)]              # job -> local clip whose stages must have finished

    def step(profiling=False):
        """One pass over this rank's clips: extract every clip (wav -> notes on the host), then decode all (clip, attribute
        tuple) jobs on the engines.  With --pipeline the stages overlap across clips instead (infer.py:82-198 order kept per
        clip: the engines admit the jobs of clip c once the extractor, on the main thread, has delivered clip c)."""
        t0 = time.perf_counter()
        ready = np.zeros(len(wavs), np.int32)
        serial = (not args.pipeline) or profiling
        if serial:
            ready[:] = 1
        n_notes = 0
        bg = None if serial else decode_jobs_async(decs, jobs, vocab, args.bar_tokens, (ready, job_clip))
        try:
            if len(exs) > 1 and not profiling:
                # several extractor instances, one host thread and stream each, clips dealt round-robin
                import threading
                cnt = [0] * len(exs); errs = []

                def run_ex(i):
                    try:
                        torch.cuda.set_device(dev)
                        with torch.cuda.stream(ex_streams[i]):
                            for c in range(i, len(wavs), len(exs)):
                                cnt[i] += len(exs[i].extract_notes(wavs[c], 44100, inf.min_duration))
                                ready[c] = 1
                        ex_streams[i].synchronize()
                    except Exception as e:      # noqa: BLE001
                        errs.append(e)
                th = [threading.Thread(target=run_ex, args=(i,)) for i in range(len(exs))]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                if errs:
                    raise errs[0]
                n_notes += sum(cnt)
            else:
                with torch.cuda.stream(ext_stream):
                    for c, wav in enumerate(wavs):
                        notes = ex.extract_notes(wav, 44100, inf.min_duration)      # device wav -> the note list extract() writes (host)
                        n_notes += len(notes)
                        ready[c] = 1
                ext_stream.synchronize()
        finally:
            ready[:] = 1                                                         # never leave a scheduler waiting
        t1 = time.perf_counter()
        if bg is None:
            bg = decode_jobs_async(decs, jobs, vocab, args.bar_tokens, (ready, job_clip), one_at_a_time=profiling)
        out, ntok = bg()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        return t1 - t0, (t2 - t1) if serial else (t2 - t0), ntok, n_notes / len(wavs), out

    for _ in range(args.warmup):
        step()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    t_ext = t_dec = 0.0
    n_tok = n_notes = 0
    for _ in range(args.steps):
        a, b, c, d, out = step()
        t_ext += a; t_dec += b; n_tok += c; n_notes = d
    gathered_jobs = len(out)
    # digest of every token this rank generated in the last timed step (job order): two builds / switches whose kernels must be
    # equivalent print the same value under the real four-engine load
    import hashlib
    tok_digest = hashlib.sha256(np.asarray([t for job in out for bar in job for t in bar], np.int32).tobytes()).hexdigest()[:16]
    if use_dist:
        # the path's only exchange: ONE final gather of the small variable-length results (token ids of every job)
        g = parallel.gather_int_arrays([np.asarray([t for bar in job for t in bar], np.int32) for job in out], device=dev, force=True)
        gathered_jobs = sum(len(x) for x in g)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event timing: one extra step over the same inputs with an event pair around every launch.  It sits
    # outside the K timed steps because event records cannot be placed inside the hipGraph replays that the
    # production decode loop uses (with the profiler on the library launches the same kernels eagerly), and it runs the
    # stages and the decoder engines one after another: an event pair on one stream also counts the time its kernel
    # queues behind the other engines' kernels, which is not that kernel's duration.
    _lib.prof_reset()
    _lib.prof_enable(True)
    step(profiling=True)
    _lib.prof_enable(False)
    prof = _lib.prof_report()
    prof_steps = 1

    tmax = torch.tensor([elapsed, t_ext, t_dec], dtype=torch.float64, device=dev)
    tsum = torch.tensor([float(n_tok)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    elapsed, t_ext, t_dec = [float(x) for x in tmax.tolist()]
    n_tok_all = float(tsum.item())

    audio_s = args.seconds * args.clips * args.steps * world
    result = {
        "metric": "audio-sec/s transcribed + decoder tokens/s, 3-min clip batch",
        "value": round(audio_s / elapsed, 3), "unit": "audio-s/s (each clip extracted and decoded for every attribute tuple)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4] share per rank: {args.clips} x 3-min 44.1 kHz stereo clips, full extract (wav->notes) each, + greedy decode of "
                               f"{args.clips}x{args.attr_grid} (clip, attribute tuple) jobs, {args.bars} synthetic condition bars x {args.bar_tokens} generated tokens each "
                               "(Bar_EOS suppressed: synthetic weights carry no musical EOS statistics), overlap bin 2, bf16 compute / fp32 accumulate; synthetic seeded weights",
                   "clips_per_gpu": args.clips, "attr_tuples_per_clip": args.attr_grid, "decode_jobs_per_gpu": n_jobs, "decoder_streams": per_eng * n_eng, "decoder_engines": n_eng, "extractor_engines": len(exs),
                   "clip_seconds": args.seconds, "windows_per_clip": int(np.ceil((1 + int(np.ceil(160 * wavs[0].shape[1] / 441)) // 256) / 512)),
                   "bars": args.bars, "bar_tokens": args.bar_tokens, "parallelism": f"clip-sharded x{world}",
                   "stage_overlap": "pipelined per clip (jobs of clip c admitted when its extraction is done; extraction of c+1 overlaps)" if args.pipeline else "stages back to back"},
        "extract_audio_s_per_s": round(audio_s / t_ext, 2),
        "decoder_tokens_per_s": round(n_tok_all / t_dec, 2),
        "decoder_tokens_per_step": n_tok / args.steps, "notes_per_clip": n_notes, "jobs_gathered": gathered_jobs, "tokens_sha256_rank0": tok_digest,
    }

    # ---- roofline of the dominant kernel (HIP events inside the library, on the stream the kernel runs on)
    MFMA_BOUND = {"k_linear", "k_linear_ln", "k_attn", "k_embed", "k_heads", "k_dgemm"}   # dense contractions; the rest stream weights / KV / audio
    if prof:
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        name, p = dom
        avg_ms = p["ms"] / max(1, p["launches"])
        if name in MFMA_BOUND:
            ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
            result["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None, "launches": p["launches"],
                                  "avg_launch_ms": round(avg_ms, 5), "alg_flops_per_launch": p["flops"] / max(1, p["launches"])}
        else:
            ach = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
            result["roofline"] = {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "launches": p["launches"], "avg_launch_ms": round(avg_ms, 5),
                                  "alg_bytes_per_launch": p["bytes"] / max(1, p["launches"])}
        tp = ROOT / "profiles" / "traffic.json"
        if tp.exists():
            try:
                result["roofline"]["traffic"] = json.loads(tp.read_text()).get(name)
                result["roofline"]["traffic_source"] = ("static: profiles/traffic.json -- rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch from separate "
                                                        "profiling passes of this same command, committed with the tree; NOT measured in this run")
            except Exception:
                pass
        result["roofline"]["frac_source"] = ("serial event pass: HIP events around every launch of the library during ONE extra step over the same inputs right after "
                                             "the timed region, stages and engines one at a time (event records cannot sit inside the hipGraph replays of the timed "
                                             "steps; with four engines running, an event pair also counts queueing behind the other engines' kernels).  The rocprofv3 "
                                             "kernel-trace average of the timed configuration is in profiles/ (tools/profile.sh).")
        result["kernel_ms_per_step"] = {k: round(v["ms"] / prof_steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        # the whole decode stage as ONE figure that needs no per-kernel timing: algorithmic bytes of every decode-step launch of the
        # step (host-side counts: K/V of every (row, head) + the weights each launch streams) over the stage's wall time in the TIMED steps
        step_k = ("k_dstep_attn_down", "k_dstep_qkv_up", "k_resid_ln_rows", "k_dstep_head", "k_dattn", "k_dgemm_s", "k_ln_rows")
        dec_bytes = sum(prof[k]["bytes"] for k in step_k if k in prof) / prof_steps
        t_dec_step = t_dec / args.steps
        if t_dec_step > 0:
            result["roofline"]["decode_stage"] = {"alg_bytes_per_step": dec_bytes, "stage_s_per_step": round(t_dec_step, 4),
                                                  "achieved": round(dec_bytes / t_dec_step / 1e9, 1), "unit": "GB/s",
                                                  "frac": round(dec_bytes / t_dec_step / 1e9 / PEAK_HBM_GBS, 4),
                                                  "note": "sum of algorithmic decode-step bytes / wall time of the decode stage in the timed steps (prefill passes included in the time, not in the bytes)"}

    # ---- extras outside the timed region
    if not args.no_extras and rank == 0:
        extras = {}
        try:
            xs = torch.from_numpy(synth.window_features(5, 16)).to(dev)          # configs[2]: 16 windows
            # (dealing the windows over both extractor instances, as the headline's extract stage does with clips, measured 1.64 against
            # 1.60 ms per window here: two 8-window halves are two batches each and gain nothing from each other -- tools/runs/r2_run49.sh)
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
            # configs[1]: ONE 3-min clip, full extract + greedy decode with the default attributes (1/1/1, overlap 2) on one engine
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            with torch.cuda.stream(ext_stream):
                n1 = len(ex.extract_notes(wavs[0], 44100, inf.min_duration))
            ext_stream.synchronize()
            t_e = time.perf_counter() - t
            t = time.perf_counter()
            st1 = {}
            decs[0].generate_many([(jobs[0][0], [synth.attrs(1, 1, 1, 2)] * len(jobs[0][0]))], vocab, stats=st1, force_bar_tokens=args.bar_tokens)
            torch.cuda.synchronize(dev)
            t_d = time.perf_counter() - t
            extras["single_clip"] = {"workload": f"BASELINE configs[1]: one 3-min 44.1 kHz clip, extract (wav -> {n1} notes) + greedy decode of {len(jobs[0][0])} bars x {args.bar_tokens} tokens, attributes 1/1/1, bf16",
                                     "extract_s": round(t_e, 4), "decode_s": round(t_d, 4), "wall_s": round(t_e + t_d, 4),
                                     "audio_s_per_s": round(args.seconds / (t_e + t_d), 1), "decoder_tokens_per_s": round(st1.get("tokens", 0) / t_d, 1)}
        except Exception as e:
            extras["single_clip"] = {"error": repr(e)}
        for key, c0 in (("decoder_streams", 512), ("decoder_streams_4k", 3500)):     # reference-faithful context / 4k stress
            try:
                extras[key] = decoder_stream_bench(dcfg, dev, ctx0=c0, streams=[d._ts for d in decs])
            except Exception as e:
                extras[key] = {"error": repr(e)}
        result["extras"] = extras

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(clip_seconds=args.seconds, windows_per_clip=result["config"]["windows_per_clip"], attr_tuples=args.attr_grid,
                                                  bars_per_job=args.bars, bar_tokens=args.bar_tokens)
        except Exception as e:
            result["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` (no launcher): run N ranks of this script under torch.distributed.run, one per GPU, over
    RCCL on 127.0.0.1.  Fails loudly -- non-zero exit, no JSON line -- when the node has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible on this node; not reporting a {n}-GPU number", file=sys.stderr)
        return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    log("bench.py: starting", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def decode_jobs_async(decs, jobs, vocab, bar_tokens, ready, one_at_a_time=False):
    """Greedy-decode all jobs on the engines (etude_amd.decoder.run_engines: jobs dealt round-robin, one host thread per
    engine).  `ready` = (flags, job -> flag index) gates the admission of each job on its clip's upstream stages.  Returns
    a function that joins and yields (results, n_tokens)."""
    from etude_amd.decoder import run_engines
    join = run_engines(decs, jobs, vocab, one_at_a_time=one_at_a_time, ready=ready, force_bar_tokens=bar_tokens,
                       stagger_s=float(os.environ.get("ETD_ENGINE_STAGGER_MS", "0")) * 1e-3)

    def join_tokens():
        out, stats = join()
        return out, sum(s["tokens"] for s in stats)

    return join_tokens


def decoder_stream_bench(dcfg, dev, n_streams: int = 128, ctx0: int = 512, steps: int = 64, engines: int = 2, streams=None):
    """BASELINE configs[3]: 128 concurrent streams, each prefilled to ctx0 then `steps` greedy decode steps
    (EOS suppressed so every stream runs the full length -- throughput does not depend on the token values).
    The streams are dealt over `engines` decoder engines (own stream, KV cache and captured graphs, shared weights) that
    step concurrently from one host thread each, like the headline's decode stage: one step of the figure below = every one
    of the n_streams streams advanced by one token.  Measured (tools/runs/r2_run31.sh): ctx 512: 0.359 / 0.346 / 0.367 ms
    with 1 / 2 / 4 engines; ctx 3.5 k: 1.30 / 1.21 / 1.18 ms.
    `streams`: torch streams the engines run on.  Inside bench.py these are the headline engines' own (idle by then): a process
    that already holds seven streams gets hardware queues for two NEW ones that may share a compute pipe, and two dependent
    kernel chains on one pipe run one after the other (0.50 instead of 0.34 ms per step at ctx 512, tools/runs/r2_run40/41.sh)."""
    import threading
    from etude_amd import _lib, synth
    from etude_amd.decoder import EtudeDecoder
    engines = max(1, min(engines, n_streams))
    per = [n_streams // engines + (1 if e < n_streams % engines else 0) for e in range(engines)]
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict(1, {}), dev, precision="bf16", max_streams=max(per), max_ctx=4096)]
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
        for s in range(per[e]):
            ids = rng.integers(6, 154, ctx0).astype(np.int32)
            cls = rng.integers(1, 3, ctx0).astype(np.int32)
            a4 = rng.integers(0, 3, (4, ctx0)).astype(np.int32)
            _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, ctx0, tg.ctypes.data,
                                                 -1, min(1000, 4096 - ctx0), st), "begin_bar")
        _lib.check(lib.etd_decoder_step(dec._h, slots[e].ctypes.data, per[e], 4, st), "step")
    torch.cuda.synchronize(dev)
    errs = []

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
    _lib.prof_reset(); _lib.prof_enable(True)
    psteps = 8
    for e, dec in enumerate(decs):                       # eager + events, one engine after the other: per-kernel breakdown
        _lib.check(lib.etd_decoder_step(dec._h, slots[e].ctypes.data, per[e], psteps, dec._stream()), "step")
        torch.cuda.synchronize(dev)
    _lib.prof_enable(False)
    prof = _lib.prof_report()
    ctx_mid = ctx0 + 4 + steps // 2
    bytes_step = sum(decs[0].step_bytes(n, ctx_mid) for n in per)       # every engine streams the weight set once per step
    gbs = bytes_step * steps / dt / 1e9
    out = {"workload": f"BASELINE configs[3]: {n_streams} streams on {engines} engine(s), bf16 weights+KV, ctx {ctx0}->{ctx0 + 4 + steps}, {steps} greedy steps, EOS suppressed",
           "engines": engines, "tokens_per_s": round(n_streams * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4),
           "alg_bytes_per_step": bytes_step,
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)},
           "kernel_ms_per_step": {k: round(v["ms"] / psteps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
    for dec in reversed(decs):
        dec.close()
    return out


if __name__ == "__main__":
    main()

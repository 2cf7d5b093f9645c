"""Run-to-run reproducibility of the Decode stage (LABNOTES.md, "packed FP32 with crossed op_sel";
profiles/r02_reproducibility.txt).

One engine gives the same tokens every time (bf16 and fp32), and so do several engines whose prefills and steps overlap, in both
modes.  The bf16 case failed until round 2: the softmax merge of the attention core held a packed-FP32 instruction that misbehaves
when the SIMD is shared with another queue's MFMA waves (csrc/dec_kernels.hip, merge_sum); tests/test_isa_guard.py keeps that
instruction form out of the library, this file checks the behaviour."""
import numpy as np
import pytest
import torch

from etude_amd import synth

pytestmark = pytest.mark.gpu


def _vocab():
    from etude_amd.vocab import Vocab
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def _jobs(n_jobs, n_bars):
    grid = [(p, r, s_) for p in range(3) for r in range(3) for s_ in range(3)]
    jobs = []
    for k in range(n_jobs):
        bars = synth.song_bars(seed=1234 + k // 27, n_bars=n_bars)
        p, r, s_ = grid[k % 27]
        jobs.append((bars, [synth.attrs(p, r, s_, 2)] * len(bars)))
    return jobs


def _run(precision, n_engines, n_jobs, n_bars, reps, bar_tokens=24, stagger_s=0.0):
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines
    assert torch.cuda.is_available()
    cfg = EtudeDecoderConfig(**synth.decoder_dims())
    per = (n_jobs + n_engines - 1) // n_engines
    decs = [EtudeDecoder(cfg, synth.decoder_state_dict(1, {}), "cuda", precision=precision, max_streams=per)]
    decs += [decs[0].clone() for _ in range(n_engines - 1)]
    jobs, v = _jobs(n_jobs, n_bars), _vocab()
    outs = []
    for _ in range(reps):
        out, _stats = run_engines(decs, jobs, v, force_bar_tokens=bar_tokens, stagger_s=stagger_s)()
        torch.cuda.synchronize()
        outs.append(out)
    for d in reversed(decs):
        d.close()
    return outs


@pytest.mark.parametrize("precision", ["f16", "fp32"])
def test_one_engine_is_reproducible(precision):
    outs = _run(precision, 1, 54 if precision == "f16" else 16, 4, 3)
    assert sum(len(b) for job in outs[0] for b in job) > 1000
    assert outs[1] == outs[0] and outs[2] == outs[0]


def test_concurrent_fp32_engines_are_reproducible():
    outs = _run("fp32", 4, 64, 3, 2)
    assert outs[1] == outs[0]


def test_concurrent_f16_engines_are_reproducible():
    """4 engines x 54 jobs x 10 bars of 32 tokens, three times, the engines started 4 ms apart so that every engine's batched prefills
    (54 prompts, the big-tile path) fall into the other engines' decode steps -- the configuration in which 15-40 of the headline's 216
    jobs used to differ between identical runs (a build with -DETD_AD_CROSSED_PK=1 fails this test)."""
    outs = _run("f16", 4, 216, 10, 3, bar_tokens=32, stagger_s=0.004)
    for k in (1, 2):
        differ = sum(1 for a, b in zip(outs[0], outs[k]) if a != b)
        assert differ == 0, f"concurrent bf16 engines: {differ} of {len(outs[0])} jobs differ between identical runs"


def test_decode_steps_beside_the_extract_stage_are_reproducible():
    """one engine's greedy decode steps alone and beside the Extract stage running on another stream: same tokens"""
    import threading
    from etude_amd.config import ExtractorConfig
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines
    from etude_amd.extractor import AMTAPC_Extractor
    cfg = EtudeDecoderConfig(**synth.decoder_dims())
    dec = EtudeDecoder(cfg, synth.decoder_state_dict(1, {}), "cuda", precision="f16", max_streams=54)
    jobs, v = _jobs(54, 4), _vocab()
    alone, _ = run_engines([dec], jobs, v, force_bar_tokens=24)()
    torch.cuda.synchronize()
    ex = AMTAPC_Extractor(ExtractorConfig(), synth.extractor_state_dict(7), "cuda", max_windows=4)
    xs = torch.from_numpy(synth.window_features(5, 4)).cuda()
    stop = [False]

    def extract():
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                ex.transcript_windows(xs)
                st.synchronize()

    th = threading.Thread(target=extract)
    th.start()
    try:
        beside = [run_engines([dec], jobs, v, force_bar_tokens=24)()[0] for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        stop[0] = True
        th.join()
    dec.close()
    assert beside[0] == alone and beside[1] == alone


def test_tokens_do_not_depend_on_the_number_of_engines():
    """the same 54 jobs on 1, 2 and 4 engines (different batch compositions, different overlap of prefills and steps): every job's
    greedy tokens are the same -- on the big-tile / fused-step paths (batches of 16 jobs and more) a row's arithmetic does not depend on
    which rows share its launch; smaller batches take other kernel paths with other bf16 roundings (tools/probe_batch_dependence.py)"""
    ref = _run("f16", 1, 54, 4, 1)[0]
    for n in (2, 4):
        out = _run("f16", n, 54, 4, 1)[0]
        differ = sum(1 for a, b in zip(ref, out) if a != b)
        assert differ == 0, f"{differ} of {len(ref)} jobs differ between 1 and {n} engines"


def test_ragged_eos_bars_are_reproducible_and_invariant_in_fp32():
    """The reference's stopping rule (Bar_EOS ends a bar) makes streams finish at different steps: the scheduler restarts them one by one (continuous batching), so WHICH rows share a
    prefill pass and a step launch depends on the tokens themselves.  With the context weights (ragged bars of 2 .. 128 tokens) and concurrent engines: (i) the 16-bit mode gives the same
    ids run after run (what extras.ragged_bars of bench.py times is a deterministic job); (ii) in the exact-parity mode the ids do not depend on the engine layout either (1 x 54 = 2 x 27)."""
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines
    cfg = EtudeDecoderConfig(**synth.decoder_dims())
    sd = synth.decoder_state_dict_ctx(1)
    v = _vocab()
    jobs = _jobs(54, 6)
    kw = dict(max_bar_token_limit=128, temperature=0.0)
    res = {}
    for prec, n_eng in (("f16", 2), ("fp32", 1), ("fp32", 2)):
        per = (len(jobs) + n_eng - 1) // n_eng
        decs = [EtudeDecoder(cfg, sd, "cuda", precision=prec, max_streams=per)]
        decs += [decs[0].clone() for _ in range(n_eng - 1)]
        outs = []
        for _ in range(2):
            out, _stats = run_engines(decs, jobs, v, **kw)()
            torch.cuda.synchronize()
            outs.append(out)
        for d in reversed(decs):
            d.close()
        assert outs[0] == outs[1], f"{prec} x {n_eng} engines: ragged ids differ between identical runs"
        res[(prec, n_eng)] = outs[0]
    lens = {len(b) for job in res[("fp32", 1)] for b in job}
    assert len(lens) > 10, "the bars of this job list should be ragged"
    assert res[("fp32", 1)] == res[("fp32", 2)], "exact-parity ids depend on the engine layout"

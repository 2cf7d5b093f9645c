"""Run-to-run reproducibility of the Decode stage ("OPEN ISSUE"; profiles/r02_reproducibility.txt).

What must hold and is asserted: one engine gives the same tokens every time (bf16 and fp32); several engines in the fp32 parity mode
give the same tokens every time even though their prefills and steps overlap.  What is known NOT to hold and is only measured and
printed: several bf16 engines whose prefills overlap other engines' steps -- a stream can flip a near-tie token from run to run."""
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


def _run(precision, n_engines, n_jobs, n_bars, reps, bar_tokens=24):
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig, run_engines
    assert torch.cuda.is_available()
    cfg = EtudeDecoderConfig(**synth.decoder_dims())
    per = (n_jobs + n_engines - 1) // n_engines
    decs = [EtudeDecoder(cfg, synth.decoder_state_dict(1, {}), "cuda", precision=precision, max_streams=per)]
    decs += [decs[0].clone() for _ in range(n_engines - 1)]
    jobs, v = _jobs(n_jobs, n_bars), _vocab()
    outs = []
    for _ in range(reps):
        out, _stats = run_engines(decs, jobs, v, force_bar_tokens=bar_tokens)()
        torch.cuda.synchronize()
        outs.append(out)
    for d in reversed(decs):
        d.close()
    return outs


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_one_engine_is_reproducible(precision):
    outs = _run(precision, 1, 54 if precision == "bf16" else 16, 4, 3)
    assert sum(len(b) for job in outs[0] for b in job) > 1000
    assert outs[1] == outs[0] and outs[2] == outs[0]


def test_concurrent_fp32_engines_are_reproducible():
    outs = _run("fp32", 4, 64, 3, 2)
    assert outs[1] == outs[0]


def test_concurrent_bf16_engines_report_their_run_to_run_difference():
    """Known open issue: prefills of one engine beside steps of another perturb the steps.  Measured on MI355X: 15-40 of 216 jobs of the
    headline workload differ between identical runs after 24 bars.  Here: 4 engines x 27 jobs x 6 bars, twice -- the number of jobs that
    differ is printed; the assertion is only a floor against something grossly worse (most jobs must still agree)."""
    outs = _run("bf16", 4, 108, 6, 2)
    differ = sum(1 for a, b in zip(outs[0], outs[1]) if a != b)
    print(f"concurrent bf16 engines: {differ} of {len(outs[0])} jobs differ between two identical runs")
    assert differ <= len(outs[0]) // 2

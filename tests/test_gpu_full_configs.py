"""Parity at the FULL size of BASELINE.json's configurations (the other GPU tests run reduced sizes).

configs[1]  one 3-min 44.1 kHz clip, default n_frame = 512 (22 windows, ragged tail), windows batched four at a time, then
            notes -> tokenizer -> greedy decode -> MIDI: against tests/golden/clip_full.npz, which make_golden.py produced by
            running the REFERENCE chain on the same clip (extractor.py:199-446, tokenizer.py, etude_decoder.py:209-354).
configs[3]  128 concurrent streams at a 3.5 k-token context (KV ring of 4096): fp32 ids against the oracle's greedy
            continuation, bf16 streams against each other and against the fp32 ids.
Also here because they need the full geometry: the KV-window bound of generate() (context_overlap_ratio 0.9) and the bf16
greedy divergence rate over the whole configs[0] song and all 27 attribute tuples.
"""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from etude_amd import synth

pytestmark = pytest.mark.gpu

from tests._util import EXT_P_MEAN as P_MEAN, EXT_P_TOL as P_TOL, close_to      # the 16-bit serving mode vs the fp32 reference (the bounds of tests/test_gpu_extractor.py)
HOP_S = 256 / 16000.0
TOL16 = 1e-2                         # the decoder's 16-bit serving mode: logits vs the fp32 oracle (tests/test_gpu_decoder_parity.py: TOL)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _vocab():
    from etude_amd.vocab import Vocab
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def match_notes(ref, got, tol_frames=1):
    """(recall, precision): a note matches when pitch is equal and the onset is within tol_frames hops; one-to-one, greedy in time"""
    tol = tol_frames * HOP_S + 1e-9
    import bisect
    by_pitch = {}
    for n in got:
        by_pitch.setdefault(n["pitch"], []).append(n["onset"])
    for v in by_pitch.values():
        v.sort()
    used = {p: [False] * len(v) for p, v in by_pitch.items()}
    hit = 0
    for n in sorted(ref, key=lambda n: n["onset"]):
        cand = by_pitch.get(n["pitch"], [])
        lo = bisect.bisect_left(cand, n["onset"] - tol)
        best, bd = -1, tol
        for i in range(lo, len(cand)):
            d = abs(cand[i] - n["onset"])
            if cand[i] > n["onset"] + tol:
                break
            if not used[n["pitch"]][i] and d <= bd:
                best, bd = i, d
        if best >= 0:
            used[n["pitch"]][best] = True
            hit += 1
    return hit / max(1, len(ref)), hit / max(1, len(got))


def test_config1_single_clip_full_chain(dev, golden_dir, tmp_path):
    from etude_amd.config import ExtractorConfig
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from etude_amd.extractor import AMTAPC_Extractor, write_wav_f32
    from etude_amd.tokenizer import TinyREMITokenizer
    from oracle import mel
    g = np.load(golden_dir / "clip_full.npz")
    wav = synth.clip_audio(seed=1234, seconds=180.0)
    # ---- the features the golden was made from, rebuilt here (CPU, oracle/mel.py) and checked against the stored rows
    feat_o = mel.wav2feature(torch.from_numpy(wav), 44100).numpy()
    T = int(g["n_frames"])
    assert feat_o.shape == (T, 256) and T == 11251
    assert np.abs(feat_o[::512] - g["feat_rows"]).max() < 1e-3
    cfg = ExtractorConfig()
    sd = synth.extractor_state_dict(0)
    ex4 = AMTAPC_Extractor(cfg, sd, "cuda", max_windows=4)
    ex1 = AMTAPC_Extractor(cfg, sd, "cuda", max_windows=1)
    feat_d = ex4.wav2feature_tensor(wav, 44100)
    assert tuple(feat_d.shape) == (T, 256) and np.abs(feat_d.cpu().numpy() - feat_o).max() < 2e-3
    # ---- 22 windows of 512 frames, ragged tail, four windows per batch vs one: not a bit may differ
    fo = torch.from_numpy(feat_o).to(dev)
    out4 = [t.cpu().numpy() for t in ex4.transcript(fo)]
    out1 = [t.cpu().numpy() for t in ex1.transcript(fo)]
    assert out4[0].shape == (11264, 88)
    for a, b in zip(out4, out1):
        assert np.array_equal(a, b)
    on, off, mpe, vel = out4
    # ---- frames against the reference (every 8th frame of the golden)
    worst = 0.0
    for name, got in (("onset_B", on), ("offset_B", off), ("mpe_B", mpe)):
        ref = g[name].astype(np.float32)
        e = np.abs(got[::8] - ref)
        worst = max(worst, float(e.max()))
        close_to(got[::8], ref, P_TOL, P_MEAN, "configs[1] 3-min clip " + name)
    vel_ref = g["velocity_B"]
    vel_agree = float((vel == vel_ref).mean())
    sat_ref = np.unpackbits(g["offset_B_sat"])[: off.size].reshape(off.shape).astype(bool)
    sat_agree = float(((off >= 1.0) == sat_ref).mean())
    # ---- notes: the reference's _mpe2note on the reference's frames (golden) vs ours on our frames, +-1 frame on the onset
    ref_notes = [dict(onset=float(a), offset=float(b), pitch=int(p), velocity=int(v))
                 for a, b, p, v in zip(g["note_onset"], g["note_offset"], g["note_pitch"], g["note_velocity"])]
    arr = ex4.mpe2note_device(*[torch.from_numpy(x).to(dev) for x in (on, off, mpe, vel)], 0.5, 1.0, 0.5)
    got_notes = ex4._notes_from_array(arr)
    assert got_notes == ex4._mpe2note(on, off, mpe, vel, 0.5, 1.0, 0.5)              # device == host note picking on 11 264 frames
    keep = lambda ns: [n for n in ns if not (n["offset"] - n["onset"] < 0.08)]       # noqa: E731  (what _note2json writes, extractor.py:435-437)
    recall_all, precision_all = match_notes(ref_notes, got_notes)
    recall, precision = match_notes(keep(ref_notes), keep(got_notes))
    print(f"configs[1] extract: max frame error {worst:.3e}, velocity argmax agreement {vel_agree:.4f}, offset saturation agreement {sat_agree:.6f}; "
          f"notes written to extract.json (>= 0.08 s): {len(keep(got_notes))} vs {len(keep(ref_notes))} reference, recall {recall:.4f} precision {precision:.4f} (+-1 frame); "
          f"all raw notes incl. sub-80-ms fragments: {len(got_notes)} vs {len(ref_notes)}, recall {recall_all:.4f} precision {precision_all:.4f}")
    # 16-bit compute on SYNTHETIC weights: seeded random weights give noise-like activations that hover around the 0.5 thresholds (49 k raw notes in 3
    # minutes), so borderline notes flip with any rounding.  IEEE-half operands (round 5): frames within 1.2e-2, written notes 0.982 / 0.979, raw 0.993
    # (bf16 operands, rounds 1-4: 6.6e-2, 0.84 / 0.86, 0.956).  The exact-parity mode below is held to >= 0.98 and measures 1.0000.
    assert recall >= 0.95 and precision >= 0.95 and recall_all >= 0.985
    assert vel_agree > 0.995 and sat_agree > 0.999
    # ---- fp32 parity mode on the same features: the reference's fp32 arithmetic on the device (csrc/ext_fp32.hip)
    ex32 = AMTAPC_Extractor(cfg, sd, "cuda", precision="fp32")
    on3, off3, mpe3, vel3 = [t.cpu().numpy() for t in ex32.transcript(fo)]
    ex32.close()
    worst32 = max(float(np.abs(got[::8] - g[name].astype(np.float32)).max()) for name, got in (("onset_B", on3), ("offset_B", off3), ("mpe_B", mpe3)))
    notes32 = ex4._mpe2note(on3, off3, mpe3, vel3, 0.5, 1.0, 0.5)
    r32, p32 = match_notes(keep(ref_notes), keep(notes32))
    r32a, p32a = match_notes(ref_notes, notes32)
    print(f"configs[1] extract, fp32 parity mode: max frame error {worst32:.3e} (golden stored as fp16: 5e-4 resolution), velocity agreement {float((vel3 == vel_ref).mean()):.5f}; "
          f"written notes recall {r32:.4f} precision {p32:.4f}; raw notes {len(notes32)} vs {len(ref_notes)}: recall {r32a:.4f} precision {p32a:.4f}")
    assert worst32 < 2e-3 and r32 >= 0.98 and p32 >= 0.98 and r32a >= 0.98 and p32a >= 0.98
    assert float((vel3 == vel_ref).mean()) > 0.999 and float(((off3 >= 1.0) == sat_ref).mean()) > 0.9999
    # ---- extract() itself: wav file -> JSON (device front end); same note population
    write_wav_f32(tmp_path / "origin.wav", wav, 44100)
    ex4.extract(str(tmp_path / "origin.wav"), str(tmp_path / "extract.json"), str(tmp_path / "extract.mid"))
    js = json.loads((tmp_path / "extract.json").read_text())
    print(f"configs[1] extract(): {len(js)} notes written from the wav file (device front end) vs {int(g['n_kept'])} the reference writes from its features")
    assert abs(len(js) - int(g["n_kept"])) <= max(3, int(g["n_kept"]) // 100)          # measured: 3 262 vs 3 248 (0.43 %: the device front end + the 16-bit model)
    assert (tmp_path / "extract.mid").read_bytes()[:4] == b"MThd"
    ex4.close(); ex1.close()
    # ---- tokenizer (native) on the REFERENCE notes -> exactly the reference's condition bars
    kept = [n for n in ref_notes if not (n["offset"] - n["onset"] < 0.08)]
    assert len(kept) == int(g["n_kept"])
    tempo = [{"start": 0.5, "bpm": 120, "time_sig": 4, "downbeats": [round(0.5 + 2.0 * i, 6) for i in range(90)]}]
    (tmp_path / "tempo.json").write_text(json.dumps(tempo))
    (tmp_path / "ref_extract.json").write_text(json.dumps(kept))
    v = _vocab()
    tk = TinyREMITokenizer(str(tmp_path / "tempo.json"))
    ids = v.encode_sequence(tk.encode(str(tmp_path / "ref_extract.json")))
    bars = tk.split_sequence_into_bars(ids, v.get_bar_bos_id(), v.get_bar_eos_id())
    lens = g["bar_lens"].tolist()
    assert [len(b) for b in bars] == lens and [t for b in bars for t in b] == g["bar_ids"].tolist()
    # ---- decode: greedy ids of the whole song, fp32 mode == the reference's generate(); bf16 agreement reported
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    dsd = synth.decoder_state_dict(1, {})
    attrs = [synth.attrs(1, 1, 1, 2)] * len(bars)
    d32 = EtudeDecoder(dcfg, dsd, "cuda", precision="fp32")
    ev = d32.generate(v, bars, attrs, temperature=0.0, top_p=0.9)
    gen = [v.encode(e) if e.type_ not in v.special_tokens else v.token_to_id[e.type_] for e in ev]
    assert gen == g["gen_ids"].tolist()                                              # bit-exact integer parity over the whole song
    ids32 = d32.generate_ids(v, bars, attrs, temperature=0.0)
    d32.close()
    d16 = EtudeDecoder(dcfg, dsd, "cuda", precision="f16")
    ids16 = d16.generate_ids(v, bars, attrs, temperature=0.0)
    d16.close()
    same_bars = sum(1 for a, b in zip(ids32, ids16) if a == b)
    print(f"configs[1] decode: {len(gen)} ids identical to the reference in fp32; bf16: {same_bars}/{len(ids32)} bars identical")
    # ---- and out to MIDI (tokenizer.py:446-524)
    notes_out = tk.decode_to_notes(ev)
    TinyREMITokenizer.note_to_midi(notes_out, tmp_path / "output.mid")
    assert (tmp_path / "output.mid").stat().st_size > 22


def _raw_generate(dec, prompts, tgt4, n_steps):
    """begin_bar per stream (raw prompts, no bar logic) + n_steps batched decode steps; returns [stream][1 + n_steps] ids"""
    from etude_amd import _lib
    lib = _lib.lib()
    st = dec._stream()
    tg = np.asarray(tgt4, np.int32)
    for s, (ids, cls, a4) in enumerate(prompts):
        ids = np.ascontiguousarray(ids, np.int32); cls = np.ascontiguousarray(cls, np.int32); a4 = np.ascontiguousarray(a4, np.int32)
        _lib.check(lib.etd_decoder_begin_bar(dec._h, s, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, len(ids), tg.ctypes.data, -1, n_steps + 1, st), "begin_bar")
    slots = np.arange(len(prompts), dtype=np.int32)
    _lib.check(lib.etd_decoder_step(dec._h, slots.ctypes.data, len(prompts), n_steps, st), "step")
    out = np.zeros((len(prompts), n_steps + 1), np.int32)
    cnt = np.zeros(len(prompts), np.int32)
    _lib.check(lib.etd_decoder_read_many(dec._h, len(prompts), slots.ctypes.data, out.ctypes.data, n_steps + 1, cnt.ctypes.data, st), "read_many")
    assert (cnt == n_steps + 1).all()
    return out


def test_config3_128_streams_at_4k_context(dev):
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from oracle import neox
    from tests._util import neox_dims, torch_sd
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    sd_np = synth.decoder_state_dict(1, {})
    sd = torch_sd(sd_np)
    nd = neox_dims({})
    ctx0, steps = 3500, 64
    rng = np.random.default_rng(17)
    tgt = [2, 1, 0, 1]                                                      # ABI order: overlap, polyphony, sustain, rhythm
    base = [(rng.integers(6, 154, ctx0), rng.integers(1, 3, ctx0), rng.integers(0, 3, (4, ctx0))) for _ in range(2)]
    # ---- oracle: greedy continuation of the two prompts (positions 3500 .. 3500 + n: far beyond max_position_embeddings = 1024,
    # where HF's rotary embedding is computed on the fly, modeling_gpt_neox.py:72-107)
    n_chk = 12
    want, want_lg, prompt_lg = [], [], []
    for ids, cls, a4 in base:
        at = {"pitch_overlap": torch.from_numpy(a4[0])[None], "polyphony": torch.from_numpy(a4[1])[None],
              "note_sustain": torch.from_numpy(a4[2])[None], "rhythm_intensity": torch.from_numpy(a4[3])[None]}
        lg, kv = neox.forward_logits(sd, nd, torch.from_numpy(ids)[None], torch.from_numpy(cls)[None], at)
        prompt_lg.append(lg[0].numpy().copy())
        seq = []
        for _ in range(n_chk + 1):
            nxt = int(torch.argmax(lg[:, -1, :], -1))
            seq.append(nxt)
            want_lg.append(lg[0, -1, :].numpy().copy())
            one = lambda x: torch.tensor([[x]])                             # noqa: E731
            lg, kv = neox.forward_logits(sd, nd, one(nxt), one(2), {"pitch_overlap": one(tgt[0]), "polyphony": one(tgt[1]),
                                                                    "note_sustain": one(tgt[2]), "rhythm_intensity": one(tgt[3])}, kv)
        want.append(seq)
    # ---- fp32 engine, the two prompts side by side: ids identical to the oracle
    d32 = EtudeDecoder(dcfg, sd_np, "cuda", precision="fp32", max_streams=2, max_ctx=4096)
    got32 = _raw_generate(d32, base, tgt, n_chk)
    d32.close()
    assert got32.tolist() == want
    # ---- bf16, 128 streams (64 copies of each prompt, interleaved), 64 steps through the captured decode step: streams
    # with equal prompts must produce equal tokens whatever slot they sit in, and the start agrees with the fp32 ids
    d16 = EtudeDecoder(dcfg, sd_np, "cuda", precision="f16", max_streams=128, max_ctx=4096)
    # the 16-bit contract is stated on logits: the batched prefill (MFMA attention from the cache rows, 55 key tiles per query) at T = 3 500 against the oracle, every 50th position
    for k, (ids, cls, a4) in enumerate(base):
        lg16 = d16.prefill_logits(ids, cls, a4)
        err = float(np.abs(lg16[::50] - prompt_lg[k][::50]).max())
        print(f"configs[3]: 16-bit prefill logits of prompt {k} at T = {ctx0}: max error {err:.3e} (tol {TOL16:.0e})")
        assert err < TOL16, (k, err)
    got16 = _raw_generate(d16, [base[s % 2] for s in range(128)], tgt, steps)
    d16.close()
    for s in range(2, 128):
        assert np.array_equal(got16[s], got16[s % 2]), s
    agree = [int(np.argmax(np.r_[got16[k][: n_chk + 1] != np.asarray(want[k]), True])) for k in range(2)]
    print(f"configs[3]: fp32 ids == oracle over {n_chk + 1} tokens at ctx {ctx0}; f16 first divergence from fp32 after {agree} tokens")
    # greedy paths may part -- but only at a near tie: where the f16 stream first leaves the fp32 ids, the token it chose sits within the 16-bit logit tolerance
    # (2 x TOL16 = 2e-2: both candidates move) of the reference's maximum at that step.  (Until round 4 prompts longer than 1 088 tokens took the fp32-query
    # attention and this test asked for >= 1 equal token; they now run on the MFMA prefill attention like every other prompt.)
    for k in range(2):
        if agree[k] <= n_chk:
            lgk = want_lg[k * (n_chk + 1) + agree[k]]
            gap = float(lgk[want[k][agree[k]]] - lgk[int(got16[k][agree[k]])])
            print(f"  prompt {k}: f16 token {int(got16[k][agree[k]])} vs {want[k][agree[k]]} at step {agree[k]}: reference logit gap {gap:.4f}")
            assert 0.0 <= gap < 2 * TOL16, (k, agree[k], gap)


def test_generate_kv_window_bound_with_large_overlap_ratio(dev):
    """context_overlap_ratio = 0.9: prompts keep int(1024 * 0.9) = 921 tokens (+ Bar_BOS), plus up to max_bar_token_limit
    generated ones -- beyond the 1088 positions round 1 allocated.  The KV window is now sized for it: ids == oracle; and a
    decoder created with a window that is too small refuses instead of wrapping silently."""
    from etude_amd import _lib
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from oracle import neox
    from tests._util import neox_dims, torch_sd
    v = _vocab()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    sd_np = synth.decoder_state_dict(1, {})
    bars = synth.song_bars(seed=8, n_bars=5, notes_per_bar=70)
    at = [synth.attrs(2, 2, 0, 2)] * len(bars)
    limit = 260
    want = neox.generate_ids(torch_sd(sd_np), neox_dims({}), 4, 5, bars, at, max_bar_token_limit=limit, context_overlap_ratio=0.9)
    dec = EtudeDecoder(dcfg, sd_np, "cuda", precision="fp32")
    assert dec.ctx_needed(limit, 0.9) == 922 + limit - 1 and dec.max_ctx >= 2048
    assert dec.generate_ids(v, bars, at, max_bar_token_limit=limit, context_overlap_ratio=0.9, temperature=0.0) == want
    dec.close()
    small = EtudeDecoder(dcfg, sd_np, "cuda", precision="fp32", max_ctx=1088)
    with pytest.raises(_lib.EtudeHipError, match="KV positions"):
        small.generate_ids(v, bars, at, max_bar_token_limit=limit, context_overlap_ratio=0.9, temperature=0.0)
    # the C ABI refuses too: prompt + limit beyond max_ctx
    ids = np.full(1000, 7, np.int32); cls = np.ones(1000, np.int32); a4 = np.ones((4, 1000), np.int32); tg = np.ones(4, np.int32)
    rc = _lib.lib().etd_decoder_begin_bar(small._h, 0, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, 1000, tg.ctypes.data, 5, 512, small._stream())
    assert rc == -22 and b"KV positions" in _lib.lib().etd_last_error()
    small.close()


def test_f16_greedy_divergence_rate_full_song_all_tuples(dev):
    """SURVEY.md section 7 hard part 2: the reference decodes in fp32; bf16 weights / KV flip near-tie argmaxes.  Whole configs[0]
    song (92 bars) x all 27 attribute tuples, fp32 engine vs bf16 engine, same scheduler: per-bar comparison.  A bar can only be
    compared while the two histories are still equal, so the rate is 'bars identical among bars whose context was identical'."""
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    v = _vocab()
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    sd_np = synth.decoder_state_dict(1, {})
    bars = synth.song_bars(seed=1234, n_bars=92)
    jobs = [(bars, [synth.attrs(p, r, s, 2)] * len(bars)) for p in range(3) for r in range(3) for s in range(3)]
    d32 = EtudeDecoder(dcfg, sd_np, "cuda", precision="fp32", max_streams=27)
    r32 = d32.generate_many(jobs, v)
    d32.close()
    d16 = EtudeDecoder(dcfg, sd_np, "cuda", precision="f16", max_streams=27)
    r16 = d16.generate_many(jobs, v)
    d16.close()
    comparable = same = tok = tok_same = 0
    first_div = []
    for a, b in zip(r32, r16):
        k = 0
        while k < min(len(a), len(b)) and a[k] == b[k]:
            k += 1
        first_div.append(k)
        n_cmp = min(k + 1, len(a), len(b))          # the bars before the first difference + the one that differs
        comparable += n_cmp
        same += k
        for i in range(n_cmp):
            m = min(len(a[i]), len(b[i]))
            tok += max(len(a[i]), len(b[i]))
            tok_same += sum(1 for x, y in zip(a[i][:m], b[i][:m]) if x == y)
    rate = 1.0 - same / max(1, comparable)
    print(f"bf16 vs fp32 greedy: {same}/{comparable} comparable bars identical (divergence rate {rate:.4f} per bar), token agreement "
          f"{tok_same / max(1, tok):.4f} on those bars; jobs identical end to end: {sum(1 for k, a in zip(first_div, r32) if k == len(a))}/27; "
          f"first differing bar per job: min {min(first_div)} median {int(np.median(first_div))}")
    assert same / max(1, comparable) >= 0.90


def test_config2_sixteen_window_batch(dev, golden_dir):
    """BASELINE configs[2]: `transcript_windows` on 16 x 512-frame windows (synth.window_features(5, 16): SURVEY 8(d) config 3), the extractor-roofline call of bench.py.
    (1) window 0 is the reference's golden window (`hft_full.npz` was recorded on window_features(5, 1) -- the same first window): inside the stated bf16 tolerance;
    (2) every window of the batch is bit-identical to the same window transcribed alone (a window's result does not depend on what shares its launches:
        extractor.py:227-248 runs them one at a time);
    (3) the exact-parity mode (fp32 extractor) on two sampled windows of the same batch sits within 2e-4 of the oracle's fp32 forward, velocity argmax on the oracle's
        maximum wherever the oracle's top-2 gap is not a rounding matter."""
    from etude_amd.config import ExtractorConfig
    from etude_amd.extractor import AMTAPC_Extractor
    from oracle import hft
    x16 = synth.window_features(5, 16)
    assert np.array_equal(x16[:1], synth.window_features(5, 1))
    g = np.load(golden_dir / "hft_full.npz")
    sdn = synth.extractor_state_dict(7, dict(n_frame=512))
    cfg = ExtractorConfig()
    ex = AMTAPC_Extractor(cfg, sdn, "cuda", max_windows=4)
    xd = torch.from_numpy(x16).to(dev)
    on, off, mpe, vel = [t.cpu().numpy() for t in ex.transcript_windows(xd)]
    assert on.shape == (16 * 512, 88) and vel.shape == (16 * 512, 88)
    for name, got in (("onset_B", on[:512]), ("offset_B", off[:512]), ("mpe_B", mpe[:512])):
        err = np.abs(got - g[name])
        close_to(got, g[name], P_TOL, P_MEAN, "configs[2] window 0 " + name)
    for w in range(16):
        a = [t.cpu().numpy() for t in ex.transcript_windows(xd[w:w + 1])]
        sl = slice(w * 512, (w + 1) * 512)
        for got, alone in zip((on[sl], off[sl], mpe[sl], vel[sl]), a):
            assert np.array_equal(got, alone), f"window {w} of the 16-window batch differs from the window transcribed alone"
    ex.close()
    ex32 = AMTAPC_Extractor(cfg, sdn, "cuda", max_windows=4, precision="fp32")
    on32, off32, mpe32, vel32 = [t.cpu().numpy() for t in ex32.transcript_windows(xd)]
    ex32.close()
    tsd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    worst = 0.0
    for w in (3, 11):
        r = hft.model_forward(tsd, torch.from_numpy(x16[w:w + 1]), hft.HftDims(n_frame=512))
        sl = slice(w * 512, (w + 1) * 512)
        for got, ref in ((on32[sl], r[5][0].numpy()), (off32[sl], r[6][0].numpy()), (mpe32[sl], r[7][0].numpy())):
            worst = max(worst, float(np.abs(got - ref).max()))
        lg = r[8][0].numpy()
        top2 = np.sort(lg, -1)[..., -2:]
        clear = (top2[..., 1] - top2[..., 0]) > 1e-3
        assert (vel32[sl].astype(np.int64) == lg.argmax(-1))[clear].all()
    print(f"configs[2], exact-parity mode: max |p - oracle| over two sampled windows = {worst:.2e} (tolerance 2e-4)")
    assert worst < 2e-4

"""Host-side logic of the drop-in layer (no GPU): prompt assembly, vocab, wav I/O, config, front-end tables."""
import json

import numpy as np
import pytest
import torch

from etude_amd import synth
from etude_amd import _lib
from etude_amd.decoder import ABI_ATTR_KEYS, EtudeDecoderConfig, expected_state_keys
from etude_amd.extractor import read_wav, write_wav_f32
from etude_amd.vocab import Event, Vocab
from oracle import neox


def _native_prompt(hist, x, y, n_ctx=4, max_pos=1024, limit=512, ratio=0.5):
    """The native scheduler's prompt for one bar (etd_debug_assemble_prompt): hist = [(xs, ys, attrs dict)]."""
    import ctypes as C
    lib = _lib.lib()
    sc = _lib.SchedCfg(bar_bos_id=4, bar_eos_id=5, n_ctx_pairs=n_ctx, max_position_embeddings=max_pos, max_output_tokens=25600,
                       max_bar_token_limit=limit, context_overlap_ratio=ratio, force_bar_tokens=0, max_streams=1, max_prefill_rows=4096,
                       steps_per_poll=8)
    n = len(hist)
    hx = [np.asarray(h[0], np.int32) for h in hist]
    hy = [np.asarray(h[1], np.int32) for h in hist]
    hxp = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in hx])
    hyp = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in hy])
    hxn = np.asarray([a.size for a in hx] or [0], np.int32)
    hyn = np.asarray([a.size for a in hy] or [0], np.int32)
    ha = np.ascontiguousarray(np.asarray([[h[2][k] for k in ABI_ATTR_KEYS] for h in hist] or [[0, 0, 0, 0]], np.int32))
    xa = np.asarray(x, np.int32)
    ya = np.asarray([y[k] for k in ABI_ATTR_KEYS], np.int32)
    cap = 4096
    ids, cls, at = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros((4, cap), np.int32)
    T = C.c_int()
    _lib.check(lib.etd_debug_assemble_prompt(C.byref(sc), n, hxp, hxn.ctypes.data, hyp, hyn.ctypes.data, ha.ctypes.data, xa.ctypes.data, xa.size,
                                             ya.ctypes.data, ids.ctypes.data, cls.ctypes.data, at.ctypes.data, cap, C.byref(T)), "assemble")
    t = T.value
    return ids[:t].tolist(), cls[:t].tolist(), {k: at[j, :t].tolist() for j, k in enumerate(ABI_ATTR_KEYS)}


def test_native_prompt_assembly_matches_oracle_incl_truncation():
    rng = np.random.default_rng(0)
    d = neox.NeoxDims()
    keys = sorted(ABI_ATTR_KEYS)
    for trial in range(40):
        hist = []
        for _ in range(int(rng.integers(0, 7))):
            xs = rng.integers(4, 150, int(rng.integers(2, 200))).tolist()
            ys = rng.integers(4, 150, int(rng.integers(2, 300))).tolist()
            hist.append((xs, ys, {k: int(rng.integers(0, 3)) for k in keys}))
        x = rng.integers(4, 150, int(rng.integers(2, 120))).tolist()
        y = {k: int(rng.integers(0, 3)) for k in keys}
        a = _native_prompt(hist, x, y)
        b = neox.build_bar_prompt(hist, x, y, keys, 4, 5, d, 512, 0.5)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], trial
        assert len(a[0]) <= 513
    # non-default knobs: 2 context pairs, tighter limits
    d2 = neox.NeoxDims(max_position_embeddings=256, context_num_past_xy_pairs=2)
    hist = [(list(range(6, 90)), list(range(10, 140)), {k: 2 for k in keys})] * 3
    x, y = list(range(20, 60)), {k: 0 for k in keys}
    assert _native_prompt(hist, x, y, n_ctx=2, max_pos=256, limit=100, ratio=0.25) == tuple(neox.build_bar_prompt(hist, x, y, keys, 4, 5, d2, 100, 0.25))
    # random knobs: history depth, position budget, bar limit and overlap ratio all varied (the oracle itself was checked against
    # the reference's generate() over random configurations of the same knobs; see oracle/README or DESIGN.md section 3)
    for trial in range(80):
        n_ctx = int(rng.integers(1, 6)); max_pos = int(rng.choice([64, 96, 128, 256, 512, 1024])); ratio = float(rng.choice([0.25, 0.3, 0.5, 0.7]))
        limit = int(rng.integers(3, max(4, max_pos // 2)))
        dd = neox.NeoxDims(max_position_embeddings=max_pos, context_num_past_xy_pairs=n_ctx)
        hist = []
        for _ in range(int(rng.integers(0, 8))):
            hist.append((rng.integers(4, 150, int(rng.integers(2, 60))).tolist(), rng.integers(4, 150, int(rng.integers(1, 90))).tolist(),
                         {k: int(rng.integers(0, 3)) for k in keys}))
        x = rng.integers(4, 150, int(rng.integers(2, 50))).tolist()
        y = {k: int(rng.integers(0, 3)) for k in keys}
        assert _native_prompt(hist, x, y, n_ctx=n_ctx, max_pos=max_pos, limit=limit, ratio=ratio) == tuple(neox.build_bar_prompt(hist, x, y, keys, 4, 5, dd, limit, ratio)), trial


def test_vocab_roundtrip_and_event_decoding(tmp_path):
    p = tmp_path / "vocab.json"
    synth.write_vocab(str(p))
    v = Vocab.load(p)
    assert len(v) == 154 and v.get_bar_bos_id() == 4 and v.get_bar_eos_id() == 5 and v.get_pad_id() == 0
    ev = v.decode_to_event(v.encode("Note_60"))
    assert ev == Event("Note", 60) and isinstance(ev.value, int)
    assert v.decode_to_event(4) == Event("Bar", "BOS")
    assert v.decode_to_event(2) == Event("<BOS>", "")
    assert v.encode("Nope_1") == v.token_to_id["<UNK>"]
    assert v.decode_sequence_to_events([4, 0, 5]) == [Event("Bar", "BOS"), Event("Bar", "EOS")]   # PAD dropped
    v.save(tmp_path / "v2.json")
    assert json.loads((tmp_path / "v2.json").read_text()) == json.loads(p.read_text())
    with pytest.raises(FileNotFoundError):
        Vocab.load(tmp_path / "missing.json")


def test_wav_roundtrip(tmp_path):
    x = synth.clip_audio(seed=1, seconds=0.25)
    write_wav_f32(tmp_path / "a.wav", x, 44100)
    y, sr = read_wav(tmp_path / "a.wav")
    assert sr == 44100 and y.shape == x.shape and np.array_equal(x, y)
    with pytest.raises(ValueError):
        (tmp_path / "b.wav").write_bytes(b"not a wav file at all")
        read_wav(tmp_path / "b.wav")


def test_decoder_config_json_contract(tmp_path):
    p = tmp_path / "etude_decoder_config.json"
    p.write_text(json.dumps(dict(synth.decoder_config_json(), architectures=["EtudeDecoder"], transformers_version="4.51.3",
                                 rotary_pct=0.25, rotary_emb_base=10000, use_parallel_residual=True, hidden_act="gelu")))
    c = EtudeDecoderConfig.from_json_file(p)
    assert (c.hidden_size, c.num_hidden_layers, c.vocab_size, c.context_num_past_xy_pairs) == (512, 8, 154, 4)
    assert c.rotary_pct == 0.25 and c.rope_theta == 10000.0 and c.layer_norm_eps == 1e-5
    keys = expected_state_keys(c)
    assert set(keys) == set(synth.decoder_state_dict(0).keys())       # == reference state_dict keys (strict-loaded in make_golden.py)


def test_frontend_tables_match_oracle_definitions():
    from etude_amd.frontend import _mel_csr, _resample_table
    from oracle import mel
    kt, width, orig, new = _resample_table(44100, 16000)
    ko, w2, o2, n2 = mel.sinc_resample_kernel(44100, 16000)
    assert (width, orig, new) == (w2, o2, n2) == (17, 441, 160) and kt.shape == (475, 160)
    assert np.array_equal(kt.T, ko.numpy())
    start, length, w = _mel_csr(1025, 8000.0, 256, 16000)
    fb = mel.melscale_fbanks(1025, 0.0, 8000.0, 256, 16000).numpy()
    dense = np.zeros_like(fb)
    p = 0
    for m in range(256):
        dense[start[m]:start[m] + length[m], m] = w[p:p + length[m]]
        p += length[m]
    assert np.array_equal(dense, fb)


def test_decoder_kv_window_requirement_formula():
    """EtudeDecoder.ctx_needed: positions a bar can touch under generate()'s truncation rule (etude_decoder.py:285-300); the
    default window (2 * max_pos + 64) covers ratio <= 1 with limits up to the 1024-token output ring."""
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    d = object.__new__(EtudeDecoder)
    d.config = EtudeDecoderConfig()
    assert d.ctx_needed(512, 0.5) == 513 + 511                  # the reference's defaults: prompt <= 512 (+ Bar_BOS), 512 generated
    assert d.ctx_needed(512, 0.9) == 922 + 511
    assert d.ctx_needed(100, 0.5) == 925 + 99                   # short limit: untruncated prompts up to max_pos - limit
    assert max(d.ctx_needed(l, r) for l in (1, 64, 512, 1024) for r in (0.0, 0.5, 0.9, 1.0)) <= 2 * 1024 + 64


def test_library_build_id_matches_the_tree():
    from etude_amd import _lib, build
    assert _lib.lib().etd_build_id().decode() == build.src_hash()


def test_traffic_table_is_keyed_by_the_names_the_bench_profiler_uses():
    """bench.py fills roofline.traffic from profiles/traffic.json[<dominant kernel>]: the committed table must be the aggregated one
    (tools/summarize_profile.py's traffic_bench.json: template instances merged), not the raw per-instantiation table"""
    import json
    from pathlib import Path
    t = json.loads((Path(__file__).resolve().parent.parent / "profiles" / "traffic.json").read_text())
    for name in ("k_dstep_attn_down", "k_dstep_qkv_up", "k_enc_layer"):
        assert isinstance(t.get(name), (int, float)) and t[name] > 0, name
    assert not any("<" in k for k in t)


def test_traffic_table_was_measured_on_the_launch_shape_the_bench_stamps():
    """roofline.traffic is static (profiles/traffic.json, PMC passes of tools/profile.sh): it must describe the launches the bench stamps -- the default engine
    layout of the 64-clip batch (rows per launch), the attention form that layout takes -- and say which library build it was measured on.  bench.py leaves
    `traffic` null when the shapes differ; this test makes a stale table fail the suite instead of going unnoticed."""
    import json
    from pathlib import Path
    import bench
    sj = json.loads((Path(__file__).resolve().parent.parent / "profiles" / "traffic.json").read_text())["k_dstep_attn_down_steady"]
    n_jobs = 64 * 27
    eng = bench.default_engines(n_jobs, 1)
    assert sj["engines"] == eng and sj["rows_per_launch"] == (n_jobs + eng - 1) // eng, (sj.get("engines"), sj.get("rows_per_launch"))
    assert sj["form"].startswith("k_dstep_attn_down<8, false, true>")          # >= 32 rows at contexts 192 .. 1024: the paired-rows form (api_dec.hip)
    assert isinstance(sj.get("build_id"), str) and len(sj["build_id"]) == 32
    assert 0.9 < sj["bytes_per_launch"] / (sj["rows_per_launch"] * 8 * 537 * 256.0 + 2.6e6) < 1.3      # ~ the algorithmic bytes of such a launch (K + V of ~537 positions per (row, head))


def _bench(args, env_extra):
    import os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ); env.update(env_extra)
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(root / "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=str(root))


def test_bench_refuses_more_gpus_than_the_node_has():
    """`bench.py --gpus N` on a node with fewer GPUs: exit 3, NO JSON line -- and the parent decides it without initialising HIP
    (GPUs are counted from the KFD topology in sysfs; here: none)"""
    import bench
    have = bench.count_gpus_without_hip()
    p = _bench(["--gpus", str(have + 2)], {})
    assert p.returncode == 3 and p.stdout.strip() == "" and "not reporting" in p.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = _bench(["--gpus", "2"], {"WORLD_SIZE": "1"})
    assert p.returncode == 2 and p.stdout.strip() == "" and "mislabelled" in p.stderr
    p = _bench(["--gpus", "1"], {"WORLD_SIZE": "4"})
    assert p.returncode == 2 and p.stdout.strip() == ""


def test_config_structs_carry_their_size():
    """since ABI version 2 every config struct leads with struct_bytes and the library refuses another layout (host-only entry point)"""
    import ctypes as C
    import numpy as np
    from etude_amd import _lib
    lib = _lib.lib()
    sc = _lib.SchedCfg(bar_bos_id=4, bar_eos_id=5, n_ctx_pairs=4, max_position_embeddings=1024, max_output_tokens=25600, max_bar_token_limit=512,
                       context_overlap_ratio=0.5, max_streams=1, max_prefill_rows=1024, steps_per_poll=8)
    assert sc.struct_bytes == C.sizeof(_lib.SchedCfg)
    x = np.asarray([4, 10, 5], np.int32); ya = np.asarray([1, 1, 1, 1], np.int32)
    ids = np.zeros(64, np.int32); cls = np.zeros(64, np.int32); a4 = np.zeros((4, 64), np.int32); T = C.c_int()
    args = (0, None, None, None, None, None, x.ctypes.data, 3, ya.ctypes.data, ids.ctypes.data, cls.ctypes.data, a4.ctypes.data, 64, C.byref(T))
    assert lib.etd_debug_assemble_prompt(C.byref(sc), *args) == 0 and T.value == 2 * 4 * 2 + 3 + 1
    sc.struct_bytes -= 8
    assert lib.etd_debug_assemble_prompt(C.byref(sc), *args) == -22 and b"etd_sched_cfg" in lib.etd_last_error()


def test_union_of_launch_intervals():
    """bench.py's roofline for concurrent engines = bytes / the union of their attention launches' (start, end) intervals (device ticks of 10 ns)"""
    import numpy as np
    import bench
    assert bench.union_seconds(np.zeros((0, 2), np.uint64)) == 0.0
    one = np.asarray([[100, 200], [200, 300], [350, 400]], np.uint64)            # one engine: back to back and a gap -> the sum
    assert abs(bench.union_seconds(one) - 250e-8) < 1e-15
    three = np.asarray([[100, 200], [150, 260], [120, 130], [400, 500], [450, 460], [499, 520]], np.uint64)   # overlapping, nested, unsorted input
    assert abs(bench.union_seconds(three[::-1]) - (160 + 120) * 1e-8) < 1e-15
    big = np.asarray([[2 ** 62, 2 ** 62 + 10]], np.uint64)                        # s_memrealtime values are large
    assert abs(bench.union_seconds(big) - 10e-8) < 1e-15


def test_default_engine_layout():
    import bench
    assert bench.default_engines(64 * 27, 1) == 3 and bench.default_engines(8 * 27, 8) == 4 and bench.default_engines(8 * 27, 1) == 4
    assert bench.default_engines(32 * 27, 2) == 4 and bench.default_engines(64 * 27, 8) == 3


def test_device_clip_synthesis_follows_config2_construction():
    """synth.clip_audio_device (bench.py's 64 clips, seeds 0..63) is synth.clip_audio's construction with device noise: same tonal part, peak 0.5, deterministic per seed"""
    import numpy as np
    from etude_amd import synth
    a = synth.clip_audio_device(5, seconds=1.0, device="cpu").numpy()
    b = synth.clip_audio_device(5, seconds=1.0, device="cpu").numpy()
    c = synth.clip_audio_device(6, seconds=1.0, device="cpu").numpy()
    ref = synth.clip_audio(5, seconds=1.0)
    assert a.shape == ref.shape == (2, 44100) and a.dtype == np.float32
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(float(np.abs(a).max()) - 0.5) < 1e-6
    assert np.corrcoef(a[0], ref[0])[0, 1] > 0.98 and np.corrcoef(a[1], ref[1])[0, 1] > 0.98       # same sinusoids; the -30 dB noise differs (device generator)


def test_plane_scale_bounds_do_bound():
    """csrc/gemm3.h: an fp32 operand x is carried as f16 planes of s x with |s x| < 2^15 by a PROVABLE bound of |x| -- checked here on adversarial LayerNorm inputs
    (one-hot and two-level rows reach the sqrt(n - 1) extreme of a normalised coordinate) and random ones, through the library's own bound functions."""
    import ctypes as C
    import numpy as np
    from etude_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    N, K = 96, 256
    W = (rng.standard_normal((N, K)) * 0.07).astype(np.float32); c = rng.standard_normal(N).astype(np.float32)
    g = rng.uniform(0.2, 3.0, K).astype(np.float32) * rng.choice([-1, 1], K).astype(np.float32); b = (rng.standard_normal(K) * 0.5).astype(np.float32)
    out = np.zeros(4, np.float32); lg = np.zeros(4, np.int32)
    _lib.check(lib.etd_debug_g3_bounds(W.ctypes.data, c.ctypes.data, N, K, g.ctypes.data, b.ctypes.data, C.c_float(7.5), out.ctypes.data, lg.ctypes.data), "g3_bounds")
    rows = [rng.standard_normal(K) * rng.uniform(1e-3, 1e3) for _ in range(200)]
    for k in range(0, K, 17):
        e = np.zeros(K); e[k] = 1.0; rows.append(e); rows.append(-e * 1e4)                  # one coordinate carries everything: |z_k| = sqrt(K - 1)
    rows.append(np.r_[np.ones(K // 2), -np.ones(K - K // 2)] * 3.0)
    x = np.stack(rows).astype(np.float64)
    z = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-5)
    y = z * g + b
    assert np.abs(y).max() <= out[0] * (1 + 1e-6), (np.abs(y).max(), out[0])
    assert np.abs(y).max() > 0.5 * out[0]                                                      # ... and the bound is tight, not vacuous
    u = y @ W.astype(np.float64).T + c
    assert np.abs(u).max() <= out[1] * (1 + 1e-6), (np.abs(u).max(), out[1])
    xe = rng.uniform(-7.5, 7.5, (300, K)); xe[0] = 7.5 * np.sign(W[0])
    assert np.abs(xe @ W.astype(np.float64).T + c).max() <= out[2] * (1 + 1e-6)
    for i in range(3):
        assert out[i] * 2.0 ** lg[i] < 2 ** 15 and out[i] * 2.0 ** lg[i] >= 2 ** 13.9           # scaled bound inside f16's range with less than a bit to spare
    assert 2 ** 14 <= out[3] < 2 ** 15 and abs(np.abs(W).max() * 2.0 ** lg[3] - out[3]) <= 8    # the weight planes' own maximum (f16 rounding of the top value)

"""The batch pipeline bench.py times (etude_amd/pipeline.py) against the STAGE-BY-STAGE calls of the reference's surface, and bench.py itself.

infer.py:82-104,165-207 per clip: extractor.extract(wav -> extract.json), analyze_volume(-> volume.json), TinyREMITokenizer(tempo.json)
.encode(extract.json) -> vocab.encode_sequence -> split_sequence_into_bars -> model.generate -> tokenizer.decode_to_notes(volume.json).
The chained array path must give exactly what those calls give, job by job."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from etude_amd import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _vocab():
    from etude_amd.vocab import Vocab
    v = Vocab()
    v.token_to_id = synth.vocab_json()["token_to_id"]
    v.id_to_token = [""] * len(v.token_to_id)
    for t, i in v.token_to_id.items():
        v.id_to_token[i] = t
    return v


def test_chained_pipeline_equals_stage_by_stage(tmp_path):
    from etude_amd.config import ExtractorConfig
    from etude_amd.decoder import EtudeDecoder, EtudeDecoderConfig
    from etude_amd.extractor import AMTAPC_Extractor, write_wav_f32
    from etude_amd.pipeline import ClipBatchPipeline, attr_grid, synthetic_tempo
    from etude_amd.preprocess import analyze_volume, save_volume_map
    from etude_amd.tokenizer import TinyREMITokenizer
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    cfg = ExtractorConfig()
    exs = [AMTAPC_Extractor(cfg, synth.extractor_state_dict(0), dev, max_windows=4) for _ in range(2)]
    dcfg = EtudeDecoderConfig(**synth.decoder_dims())
    # fp32 engines: bf16 tokens of a job may depend on the batch it is decoded in (skinny vs big-tile GEMM roundings); fp32 does not
    decs = [EtudeDecoder(dcfg, synth.decoder_state_dict_ctx(1), dev, precision="fp32", max_streams=5)]
    decs.append(decs[0].clone())
    v = _vocab()
    tempo = synthetic_tempo(n_downbeats=12)
    grid = attr_grid(27)[::9] + attr_grid(1)                      # four tuples
    pipe = ClipBatchPipeline(exs, decs, v, tempo, grid, 44100, max_bar_token_limit=40)
    base = synth.clip_audio(seed=77, seconds=26.0)
    waves = [base, np.ascontiguousarray(np.roll(base, 40000, axis=1) * np.float32(0.8)), np.ascontiguousarray(base[:, : 44100 * 19])]
    r = pipe.run([torch.from_numpy(w).to(dev) for w in waves])
    assert len(r["results"]) == len(waves) * len(grid) == len(r["notes"])
    (tmp_path / "tempo.json").write_text(json.dumps(tempo))
    single = EtudeDecoder(dcfg, synth.decoder_state_dict_ctx(1), dev, precision="fp32", max_streams=1)
    n_notes = 0
    for c, w in enumerate(waves):
        wav = tmp_path / f"origin{c}.wav"
        write_wav_f32(wav, w, 44100)
        exs[0].extract(str(wav), str(tmp_path / "extract.json"))                              # infer.py:90-96
        save_volume_map(analyze_volume(wav), tmp_path / "volume.json")                          # infer.py:99-104
        js = json.loads((tmp_path / "extract.json").read_text())
        cd = r["conditions"][c]
        assert len(js) == cd.notes.size
        assert [n["onset"] for n in js] == cd.notes["onset"].tolist() and [n["pitch"] for n in js] == cd.notes["pitch"].tolist()
        assert np.array_equal(np.asarray(json.loads((tmp_path / "volume.json").read_text()), np.float64), cd.volume.astype(np.float64))
        tk = TinyREMITokenizer(str(tmp_path / "tempo.json"))                                   # infer.py:180-186
        ids = v.encode_sequence(tk.encode(str(tmp_path / "extract.json")))
        bars = tk.split_sequence_into_bars(ids, v.get_bar_bos_id(), v.get_bar_eos_id())
        assert [cd.bars.bar(i) for i in range(len(cd.bars))] == bars and len(bars) > 8
        for k, attrs in enumerate(grid):
            ev = single.generate(v, bars, [attrs] * len(bars), max_bar_token_limit=40, temperature=0.0, top_p=0.9)   # infer.py:192-198
            want_ids = [v.encode(e) if e.type_ not in v.special_tokens else v.token_to_id[e.type_] for e in ev]
            flat, lens = r["results"][c * len(grid) + k]
            assert flat.tolist() == want_ids and int(lens.sum()) == len(want_ids)
            want_notes = tk.decode_to_notes(ev, volume_map_path=str(tmp_path / "volume.json"))   # infer.py:202-204
            got = r["notes"][c * len(grid) + k]
            assert [(n["pitch"], n["onset"], n["offset"], n["velocity"]) for n in want_notes] == \
                list(zip(got["pitch"].tolist(), got["onset"].tolist(), got["offset"].tolist(), got["velocity"].tolist()))
            n_notes += len(want_notes)
    assert n_notes > 0
    single.close()
    pipe.close()
    for d in reversed(decs):
        d.close()
    for e in exs:
        e.close()


def _run_bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                       # ONE JSON line on stdout
    return json.loads(lines[0])


def _no_errors(node, path="result"):
    if isinstance(node, dict):
        assert "error" not in node, f"{path}: {node['error']}"
        for k, v in node.items():
            _no_errors(v, f"{path}.{k}")


def test_bench_line_small_configuration():
    """bench.py end to end on a small batch: contract keys, roofline / cpu_baseline objects, no silent {"error": ...} in any extra,
    strong-scaling label, device stamps present and consistent with the exact byte counters"""
    d = _run_bench(["--batch-clips", "2", "--attr-grid", "3", "--seconds", "24", "--steps", "1", "--warmup", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "extras", "tokens_sha256_rank0"):
        assert k in d, k
    _no_errors(d)
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "strong" and d["dtype"] == "f16" and d["vs_baseline"] is None
    assert d["config"]["batch_clips"] == 2 and d["jobs_gathered"] == 6
    r = d["roofline"]
    assert r["kernel"] == "k_dstep_attn_down" and r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["launches"] > 0 and r["decode_stage"]["frac"] > 0
    assert set(d["extras"]) >= {"extractor_only", "single_clip", "decoder_streams", "decoder_streams_4k"}
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    # the checker leg: the CPU oracle's greedy bars of cpu_baseline() against the exact-parity engine's ids for the same clip / tuple / condition bars
    chk = d["extras"].get("oracle_check") or d["extras"]["parity_mode"]["oracle_check"]
    assert chk["bars"] == 5 and chk["oracle_bars_identical"] == "5/5", chk
    assert "_oracle_ids" not in d["cpu_baseline"]
    assert abs(d["value"] - 24.0 * 2 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3


def test_bench_spawned_single_rank_uses_the_launcher_and_rccl():
    """ETD_FORCE_SPAWN=1 python bench.py --gpus 1: the parent stays GPU-free (GPUs counted from sysfs), starts torch.distributed.run with
    one fresh rank, which initialises RCCL and runs the real gather; rank 0's line is relayed"""
    d = _run_bench(["--gpus", "1", "--batch-clips", "1", "--attr-grid", "2", "--seconds", "20", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                   {"ETD_FORCE_SPAWN": "1"})
    assert d["n_gpus"] == 1 and d["jobs_gathered"] == 2 and d["value"] > 0

import sys, types, importlib.util
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/reference")   # this container only
spec = importlib.util.spec_from_file_location("mg", "/root/repo/tests/golden/make_golden.py"); mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
from etude_amd import synth
from oracle import neox
import logging; logging.disable(logging.CRITICAL)
vocab = mg._Vocab().v
bos, eos = vocab.get_bar_bos_id(), vocab.get_bar_eos_id()
bad = 0; N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for case in range(N):
    r = np.random.default_rng(300 + case)
    dims = dict(mg.TINY_DEC); dims["max_position_embeddings"] = int(r.choice([64, 96, 128, 200])); dims["context_num_past_xy_pairs"] = int(r.integers(1, 5))
    try:
        model, d = mg.ref_decoder(dims, seed=int(r.integers(1, 50)), gain=float(r.choice([1.0, 2.0, 3.0])), p_eos=float(r.choice([0.05, 0.15, 0.4])))
    except TypeError as e:
        print("ref_decoder signature:", e); break
    n_bars = int(r.integers(2, 9))
    bars = synth.song_bars(seed=int(r.integers(0, 1000)), n_bars=n_bars, notes_per_bar=int(r.integers(2, 9)))
    attrs = [synth.attrs(int(r.integers(0, 3)), int(r.integers(0, 3)), int(r.integers(0, 3)), int(r.integers(0, 3))) for _ in range(n_bars)]
    lim = int(r.integers(3, 30)); mot = int(r.choice([25600, int(r.integers(5, 80))])); cor = float(r.choice([0.3, 0.5, 0.7]))
    if lim >= dims["max_position_embeddings"] - 8: lim = 10
    try:
        ev = model.generate(vocab, bars, attrs, temperature=0.0, top_p=0.9, max_bar_token_limit=lim, max_output_tokens=mot, context_overlap_ratio=cor)
        ref_ids = [vocab.encode(e) if e.type_ not in vocab.special_tokens else vocab.token_to_id[e.type_] for e in ev]
    except Exception as e:
        ref_ids = ("EXC", repr(e)[:100])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    nd = neox.NeoxDims(**{k: d[k] for k in neox.NeoxDims.__dataclass_fields__ if k in d}) if hasattr(neox.NeoxDims, "__dataclass_fields__") else neox.NeoxDims()
    try:
        got = neox.generate_ids(sd, nd, bos, eos, bars, attrs, max_output_tokens=mot, max_bar_token_limit=lim, context_overlap_ratio=cor)
        got_ids = [t for b in got for t in b]
    except Exception as e:
        got_ids = ("EXC", repr(e)[:100])
    if ref_ids != got_ids:
        bad += 1; print("MISMATCH case", case, dims["max_position_embeddings"], dims["context_num_past_xy_pairs"], lim, mot, cor, (ref_ids if isinstance(ref_ids, tuple) else len(ref_ids)), (got_ids if isinstance(got_ids, tuple) else len(got_ids)))
print("cases", N, "mismatches", bad)

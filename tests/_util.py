"""Shared helpers for the tests (dims of the golden configs, state-dict conversion)."""
import numpy as np
import torch

from etude_amd import synth

TINY_EXT = dict(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=32, pf_dim=64,
                n_heads=4, n_layers_enc=3, n_layers_dec=3, n_note=12, n_velocity=8)
TINY_DEC = dict(vocab_size=154, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                max_position_embeddings=128, attribute_emb_dim=16)
TINY_DEC_KW = dict(gain=2.0, p_eos=0.15)


def torch_sd(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def hft_dims(over):
    from oracle.hft import HftDims
    return HftDims(**synth.extractor_dims(**over))


def neox_dims(over):
    from oracle.neox import NeoxDims
    d = synth.decoder_dims(**over)
    return NeoxDims(vocab_size=d["vocab_size"], hidden_size=d["hidden_size"], num_hidden_layers=d["num_hidden_layers"],
                    num_attention_heads=d["num_attention_heads"], intermediate_size=d["intermediate_size"],
                    max_position_embeddings=d["max_position_embeddings"], attribute_emb_dim=d["attribute_emb_dim"],
                    context_num_past_xy_pairs=d["context_num_past_xy_pairs"])


def split_generated(ids, bos):
    """Flat generated id list -> per-bar lists (each starts with Bar_BOS)."""
    bars, cur = [], None
    for t in ids:
        if t == bos:
            if cur is not None:
                bars.append(cur)
            cur = [t]
        else:
            cur.append(t)
    if cur is not None:
        bars.append(cur)
    return bars

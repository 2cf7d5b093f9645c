"""ctypes binding of libetude_hip.so (the C ABI declared in include/etude_hip.h).

``import torch`` happens first on purpose: torch ships its own ``libamdhip64.so`` (same SONAME as
/opt/rocm's), and loading our library afterwards makes the dynamic linker bind it to that already
loaded HIP runtime -- one runtime per process, so ``tensor.data_ptr()`` addresses and
``torch.cuda`` streams are valid inside the library.

There is NO fallback: if the shared object is missing or a symbol is absent this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

LIB_PATH = Path(os.environ["ETD_LIB_PATH"]) if os.environ.get("ETD_LIB_PATH") else Path(__file__).resolve().parent / "libetude_hip.so"     # (ETD_LIB_PATH: measurement builds side by side, tools/runs)

c_int_p = C.POINTER(C.c_int)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_f32_p = C.POINTER(C.c_float)


class _SizedCfg(C.Structure):
    """Config structs (since ABI version 2) lead with `struct_bytes` = sizeof of the caller's layout; the library refuses any other."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_bytes = C.sizeof(type(self))


class ExtCfg(_SizedCfg):
    _fields_ = [("struct_bytes", C.c_int)] + [(n, C.c_int) for n in ("n_margin", "n_frame", "n_bin", "cnn_channel", "cnn_kernel", "hid_dim", "pf_dim",
                                       "n_heads", "n_layers_enc", "n_layers_dec", "n_note", "n_velocity")] + \
               [("min_value", C.c_float), ("max_windows", C.c_int), ("chunk_frames", C.c_int), ("precision", C.c_int)]


class DecCfg(_SizedCfg):
    _fields_ = [("struct_bytes", C.c_int)] + [(n, C.c_int) for n in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                       "intermediate_size", "max_position_embeddings", "num_classes",
                                       "num_attribute_bins", "attribute_emb_dim")] + \
               [("rotary_pct", C.c_float), ("rope_theta", C.c_float), ("layer_norm_eps", C.c_float),
                ("max_streams", C.c_int), ("max_ctx", C.c_int), ("precision", C.c_int), ("max_prefill_rows", C.c_int)]


class Job(C.Structure):
    _fields_ = [("x_ids", C.c_void_p), ("x_offsets", C.c_void_p), ("n_bars", C.c_int), ("attrs4", C.c_void_p), ("ready", C.c_void_p)]


class SchedCfg(_SizedCfg):
    _fields_ = [("struct_bytes", C.c_int)] + [(n, C.c_int) for n in ("bar_bos_id", "bar_eos_id", "n_ctx_pairs", "max_position_embeddings", "max_output_tokens",
                                       "max_bar_token_limit")] + [("context_overlap_ratio", C.c_float)] + \
               [(n, C.c_int) for n in ("force_bar_tokens", "max_streams", "max_prefill_rows", "steps_per_poll")] + \
               [("temperature", C.c_float), ("top_p", C.c_float), ("seed", C.c_ulonglong), ("job_key_offset", C.c_int), ("job_key_stride", C.c_int)]


class TempoRegion(C.Structure):
    _fields_ = [("bpm", C.c_double), ("time_sig", C.c_int), ("start", C.c_double), ("downbeats", C.c_void_p), ("n_downbeats", C.c_int)]


class TokEvent(C.Structure):
    _fields_ = [("type", C.c_int32), ("value", C.c_int32)]


class Note(C.Structure):
    _fields_ = [("onset", C.c_double), ("offset", C.c_double), ("pitch", C.c_int32), ("velocity", C.c_int32)]


ABI_VERSION = 3          # == ETD_ABI_VERSION of include/etude_hip.h; lib() refuses any other

# name -> (restype, argtypes); mirrors include/etude_hip.h (the boundary) and include/etude_hip_debug.h (test / measurement hooks) one to one
SIGNATURES = {
    "etd_version": (C.c_int, []),
    "etd_last_error": (C.c_char_p, []),
    "etd_build_id": (C.c_char_p, []),
    "etd_extractor_operand_type": (C.c_int, []),
    "etd_decoder_operand_type": (C.c_int, []),
    "etd_has_experiments": (C.c_int, []),
    "etd_prof_enable": (C.c_int, [C.c_int]),
    "etd_prof_reset": (C.c_int, []),
    "etd_prof_collect": (C.c_int, []),
    "etd_prof_count": (C.c_int, []),
    "etd_prof_entry": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "etd_debug_boundary_cost": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "etd_debug_linear": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]),
    "etd_debug_kernel_loop": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "etd_debug_g3_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "etd_debug_gemm3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "etd_debug_attn3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "etd_debug_empty_launch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "etd_frontend_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_void_p)]),
    "etd_frontend_destroy": (None, [C.c_void_p]),
    "etd_frontend_set_pad_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "etd_rms_frames": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]),
    "etd_frontend_resampled_len": (C.c_longlong, [C.c_void_p, C.c_longlong]),
    "etd_frontend_num_frames": (C.c_longlong, [C.c_void_p, C.c_longlong]),
    "etd_frontend_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong,
                                   C.POINTER(C.c_longlong), C.c_void_p]),
    "etd_extractor_create": (C.c_int, [C.POINTER(ExtCfg), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), c_i64_p, C.c_int,
                                       C.POINTER(C.c_void_p)]),
    "etd_extractor_destroy": (None, [C.c_void_p]),
    "etd_transcript": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_void_p] * 8 + [C.c_void_p]),
    "etd_transcript_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_void_p]),
    "etd_extractor_debug_vel_logits": (C.c_int, [C.c_void_p, C.c_void_p]),
    "etd_extractor_debug_tap": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "etd_extractor_window_flops": (C.c_double, [C.c_void_p]),
    "etd_mpe2note": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_float,
                               C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(Note), C.c_longlong, C.POINTER(C.c_longlong)]),
    "etd_mpe2note_modes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Note), C.c_longlong, C.POINTER(C.c_longlong)]),
    "etd_mpe2note_dev_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "etd_mpe2note_dev_destroy": (None, [C.c_void_p]),
    "etd_mpe2note_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float, C.c_float,
                                   C.c_int, C.c_int, C.c_int, C.POINTER(Note), C.c_longlong, C.POINTER(C.c_longlong), C.c_void_p]),
    "etd_mpe2note_dev_modes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float, C.c_float,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Note), C.c_longlong, C.POINTER(C.c_longlong), C.c_void_p]),
    "etd_tok_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "etd_tok_destroy": (None, [C.c_void_p]),
    "etd_tok_num_measures": (C.c_int, [C.c_void_p]),
    "etd_tok_measures": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "etd_tok_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong)]),
    "etd_tok_split_bars": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                     C.POINTER(C.c_longlong)]),
    "etd_tok_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                 C.POINTER(C.c_longlong)]),
    "etd_midi_write": (C.c_int, [C.c_void_p, C.c_longlong, C.c_char_p]),
    "etd_decoder_create": (C.c_int, [C.POINTER(DecCfg), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), c_i64_p, C.c_int,
                                     C.POINTER(C.c_void_p)]),
    "etd_decoder_destroy": (None, [C.c_void_p]),
    "etd_decoder_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "etd_decoder_set_sampling": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_ulonglong, C.c_void_p]),
    "etd_decoder_set_keys": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "etd_decoder_begin_bar": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_void_p]),
    "etd_decoder_begin_bars": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_void_p]),
    "etd_decoder_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "etd_decoder_poll": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "etd_decoder_read_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, c_int_p, C.c_void_p]),
    "etd_decoder_read_many": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "etd_decoder_run_jobs": (C.c_int, [C.c_void_p, C.POINTER(SchedCfg), C.POINTER(Job), C.c_int, C.c_void_p, C.c_longlong, C.c_void_p,
                                       C.POINTER(C.c_longlong), C.c_void_p]),
    "etd_debug_assemble_prompt": (C.c_int, [C.POINTER(SchedCfg), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_int_p]),
    "etd_decoder_generate_bar": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_int, C.c_int, C.c_void_p, c_int_p, C.c_void_p]),
    "etd_decoder_prefill_logits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p]),
    "etd_decoder_step_bytes": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "etd_decoder_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_void_p]),
    "etd_decoder_stats_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "etd_decoder_stamp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "etd_decoder_stamp_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong), C.c_void_p]),
    "etd_debug_decoder_force_pair": (C.c_int, [C.c_void_p, C.c_int]),
    "etd_debug_decoder_step_logits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "etd_debug_decoder_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int, C.c_void_p]),
    "etd_debug_decoder_kv_rowsums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "etd_debug_decoder_trace_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "etd_debug_decoder_trace_slabs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "etd_debug_decoder_trace_lanes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "etd_debug_decoder_trace_q": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "etd_debug_decoder_peek_kv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "etd_debug_decoder_trace_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
}

_lib = None


class EtudeHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the library; raises if it is missing -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise EtudeHipError(f"{LIB_PATH} not found: build it with `python -m etude_amd.build` "
                                "(hipcc --offload-arch=gfx950); etude_amd has no CPU fallback")
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("ETD_PARTIAL") and not hasattr(l, name):
                continue                   # bring-up only: a library built from a subset of the sources
            fn = getattr(l, name)          # AttributeError if the .so does not export it: fail loudly
            fn.restype = res
            fn.argtypes = args
        if l.etd_version() != ABI_VERSION:
            raise EtudeHipError(f"{LIB_PATH} speaks ABI version {l.etd_version()}, this binding {ABI_VERSION}; rebuild with `python -m etude_amd.build`")
        # provenance: the binary must have been built from the sources it sits next to (diagnostic builds opt out explicitly)
        if not os.environ.get("ETD_PARTIAL") and not os.environ.get("ETD_ALLOW_STALE_LIB"):
            from .build import src_hash
            have, want = (l.etd_build_id() or b"").decode(), src_hash()
            if have != want:
                raise EtudeHipError(f"{LIB_PATH} is stale: built from sources {have}, the tree is {want}; rebuild with `python -m etude_amd.build`")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().etd_last_error()
        raise EtudeHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def weights_arrays(state: dict):
    """name->float32 ndarray dict  ->  (names**, ptrs**, numels*, n, keepalive)."""
    import numpy as np
    names = list(state.keys())
    arrs = [np.ascontiguousarray(np.asarray(state[k], dtype=np.float32)) for k in names]
    c_names = (C.c_char_p * len(names))(*[k.encode() for k in names])
    c_ptrs = (C.c_void_p * len(names))(*[a.ctypes.data for a in arrs])
    c_num = (C.c_int64 * len(names))(*[a.size for a in arrs])
    return c_names, c_ptrs, c_num, len(names), arrs


def prof_enable(on: bool) -> None:
    lib().etd_prof_enable(1 if on else 0)


def prof_reset() -> None:
    lib().etd_prof_reset()


def prof_report() -> dict:
    """name -> dict(ms, launches, flops, bytes) accumulated since the last reset (synchronises)."""
    l = lib()
    l.etd_prof_collect()
    out = {}
    for i in range(l.etd_prof_count()):
        name = C.create_string_buffer(64)
        ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        check(l.etd_prof_entry(i, name, 64, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "etd_prof_entry")
        out[name.value.decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out

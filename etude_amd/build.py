"""Build libetude_hip.so (gfx950) in-tree with hipcc.  ``python -m etude_amd.build [--force]``.

The shared object lands next to this file so that it travels with a repository snapshot; there is
no JIT cache and no pip install.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "csrc" / "_obj"
LIB = Path(os.environ["ETD_LIB_OUT"]).resolve() if os.environ.get("ETD_LIB_OUT") else HERE / "libetude_hip.so"      # (ETD_LIB_OUT: measurement builds side by side; load them with ETD_LIB_PATH)
SOURCES = ["ext_kernels.hip", "ext_fused.hip", "ext_fp32.hip", "api_ext.hip", "frontend.hip", "dec_kernels.hip", "dec_fused.hip", "dec_prefill.hip", "gemm3.hip", "api_dec.hip", "mpe2note.cpp", "mpe2note_dev.hip", "prof.hip", "sched_dec.cpp", "tokenizer.cpp", "midi.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result",
         "-I", str(HERE.parent / "include")]
# gfx950 can hand the first kernel arguments to a wave in SGPRs at launch (kernarg preload): kernels whose hot arguments are
# leading scalars / pointers then start without the initial s_load round trip.  ETD_KERNARG_PRELOAD=0 turns it off.
if os.environ.get("ETD_KERNARG_PRELOAD", "1") != "0":
    FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
# MFMA results in VGPRs (gfx950's register file is unified): left to its default, hipcc puts every compiler-selected MFMA accumulator into AGPRs and copies it to
# VGPRs and back around each VALU use (online-softmax rescale, epilogues): 2 209 v_accvgpr moves across the library, 256 of them per key tile in the prefill
# attention.  With the VGPR form there are none, the kernels need 15-40 % fewer registers (k_attn 200 -> 134, k_embed 424 -> 250) and the arithmetic is unchanged.
# (k_dmlp_fused places its operands by asm constraints and is not affected.)  ETD_MFMA_VGPR_FORM=0 turns it off.
SIDECAR = LIB.with_suffix(".flags.json")      # the optional flags the .so next to it was built with (written by build(), read by src_hash(): loading never probes)
_LOADED = Path(os.environ["ETD_LIB_PATH"]).resolve().with_suffix(".flags.json") if os.environ.get("ETD_LIB_PATH") else SIDECAR      # ... of the .so a process LOADS
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]


def _probe_llvm_flag(flag: str) -> bool:
    """Does this hipcc's LLVM know the (hidden) -mllvm option?  One trivial gfx950 compile.  Called from build() ONLY (never at import, never when the library is
    loaded: a serving / test / profiled process that has initialised the GPU must not fork a compiler): a toolchain without the option would otherwise abort the
    whole build with 'Unknown command line argument'."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = Path(td) / "p.hip"
        src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(float* p) { p[0] = 1.f; }\n")
        r = subprocess.run([_hipcc(), "-O1", "--offload-arch=gfx950", "-mllvm", flag, "-c", str(src), "-o", str(Path(td) / "p.o")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode == 0


def _optional_flags(probe: bool = False) -> list:
    """The flags that depend on the toolchain.  probe=True (build()): ask the compiler and record the answer in SIDECAR.  probe=False (importing this module,
    `src_hash()` for the loader's build-id check): the answer recorded beside the .so, or -- no sidecar, e.g. a tree that was never built -- the default set."""
    import json
    if os.environ.get("ETD_MFMA_VGPR_FORM", "1") == "0":
        return []
    if probe:
        ok = _probe_llvm_flag("-amdgpu-mfma-vgpr-form")
        if not ok:
            print("etude_amd.build: this hipcc does not know -mllvm -amdgpu-mfma-vgpr-form; building without it (MFMA accumulators in AGPRs: correct, 15-40 % more registers)",
                  file=sys.stderr)
        try:
            SIDECAR.write_text(json.dumps({"mfma_vgpr_form": ok}))
        except OSError:
            pass
        return list(_VGPR_FORM) if ok else []
    try:
        return list(_VGPR_FORM) if json.loads(_LOADED.read_text()).get("mfma_vgpr_form", True) else []
    except Exception:      # noqa: BLE001
        return list(_VGPR_FORM)


BASE_FLAGS = list(FLAGS)
EXTRA_FLAGS = os.environ.get("ETD_EXTRA_FLAGS", "").split()      # diagnostic builds (e.g. -DETD_HEAD_STAMP, -DETD_LIN_STAMP)
FLAGS = BASE_FLAGS + _optional_flags() + EXTRA_FLAGS      # (part of every file's flag set, hence of the build id: builds with and without an option never mix)
# -ffp-contract=off applies to HOST code only in effect: device kernels use explicit fmaf where wanted.


def file_flags(src_name: str) -> list:
    """Effective compile flags of one source: the common FLAGS (incl. ETD_EXTRA_FLAGS) + its ETD_FLAGS_<FILE> experiment flags."""
    return FLAGS + os.environ.get("ETD_FLAGS_" + src_name.split(".")[0].upper(), "").split()


def _flags_tag(src_name: str) -> str:
    import hashlib
    return hashlib.sha256("\0".join(file_flags(src_name)).encode()).hexdigest()[:10]


def src_hash() -> str:
    """sha256 over every source and header the library is built from AND the effective compile flags of every file.  build() bakes
    it into the .so (`etd_build_id`) and `_lib.lib()` compares it with the tree (and environment) it is loaded from: a stale binary
    next to newer sources -- or a diagnostic build made with ETD_EXTRA_FLAGS / ETD_FLAGS_* (some of which compute wrong results on
    purpose) loaded by a process that does not ask for those flags -- fails loudly instead of standing in for the shipped kernels."""
    import hashlib
    h = hashlib.sha256()
    inc = HERE.parent / "include"
    files = sorted([CSRC / s for s in SOURCES] + list(CSRC.glob("*.h")) + [inc / "etude_hip.h", inc / "etude_hip_debug.h"], key=lambda p: p.name)
    for f in files:
        if f.exists():
            h.update(f.name.encode()); h.update(b"\0"); h.update(f.read_bytes()); h.update(b"\0")
    for s in SOURCES:
        h.update(s.encode()); h.update(b"\0"); h.update("\0".join(x for x in file_flags(s) if x != str(inc)).encode()); h.update(b"\0")
    return h.hexdigest()[:32]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (Path(c).exists() or c == "hipcc"):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> Path:
    global FLAGS
    OBJ.mkdir(exist_ok=True)
    hipcc = _hipcc()
    FLAGS = BASE_FLAGS + _optional_flags(probe=True) + EXTRA_FLAGS    # the ONE place that asks the compiler; the answer is recorded beside the .so
    headers = list(CSRC.glob("*.h")) + [HERE.parent / "include" / "etude_hip.h", HERE.parent / "include" / "etude_hip_debug.h"]
    newest_h = max(h.stat().st_mtime for h in headers)
    objs = []
    rebuilt = False
    procs = []
    for s in SOURCES:
        src = CSRC / s
        if not src.exists():
            continue
        o = OBJ / f"{s}.{_flags_tag(s)}.o"          # one object per (source, flag set): builds with different flags never reuse each other's objects
        objs.append(o)
        if force or not o.exists() or o.stat().st_mtime < max(src.stat().st_mtime, newest_h):
            cmd = [hipcc, *file_flags(s), "-c", str(src), "-o", str(o)]   # incl. per-file experiment flags, e.g. ETD_FLAGS_EXT_FUSED
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            rebuilt = True
    # the build id: a generated translation unit holding the hash of the sources this binary was made from
    bid_src, bid_obj = OBJ / "build_id.cpp", OBJ / "build_id.cpp.o"
    want = 'extern "C" const char* etd_build_id(void) { return "%s"; }\n' % src_hash()
    if force or not bid_obj.exists() or not bid_src.exists() or bid_src.read_text() != want:
        bid_src.write_text(want)
        procs.append(("build_id.cpp", subprocess.Popen([hipcc, "-O1", "-fPIC", "-c", "-x", "c++", str(bid_src), "-o", str(bid_obj)],
                                                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        rebuilt = True
    objs.append(bid_obj)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    if rebuilt or not LIB.exists():
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    if force and not os.environ.get("ETD_LIB_OUT"):
        # a forced build of the shipped library is the tree's clean point: objects of other flag sets / older sources (measurement builds) go
        for o in OBJ.glob("*.o"):
            if o not in objs:
                o.unlink()
        (OBJ / "flag_probe.json").unlink(missing_ok=True)
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)

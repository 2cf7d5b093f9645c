"""The built library holds no packed-FP32 instruction whose LOW result reads an operand's HIGH register (v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32 with a 1 in op_sel).  Measured rule (tools/ubench/pk_cross_repro.hip, profiles/r02_pk_cross_repro.txt): with op_sel[1] = 1 --
SRC1's high register feeding the low result -- that operand reads as 0 in lanes 48-63 while another wave's MFMA runs on the SIMD; the
guard checks the superset "any 1 in op_sel".  In the attention core this made concurrent decoding irreproducible (LABNOTES.md, "packed FP32 with
crossed op_sel"; csrc/dec_kernels.hip merge_sum / fma_scalar).  hipcc's SLP vectoriser emits it on its own, so the check is on the
machine code of every gfx950 code object in libetude_hip.so (tools/isa_scan.py)."""
import importlib.util
import os
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _scan_module():
    spec = importlib.util.spec_from_file_location("isa_scan", ROOT / "tools" / "isa_scan.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_no_packed_fp32_instruction_reads_a_high_register_into_its_low_result():
    m = _scan_module()
    lib = ROOT / "etude_amd" / "libetude_hip.so"
    if not lib.exists():
        pytest.skip("libetude_hip.so is not built")
    if not os.path.exists(m.OBJDUMP):
        pytest.skip("llvm-objdump is not installed")
    bad = m.scan(lib)
    assert not bad, "crossed packed-FP32 instructions in: " + ", ".join(sorted({k for k, _ in bad}))


def test_the_scanner_recognises_the_form():
    m = _scan_module()
    hit = "v_pk_fma_f32 v[66:67], v[64:65], v[6:7], v[52:53] op_sel:[0,0,1] op_sel_hi:[1,1,0]"
    miss = "v_pk_fma_f32 v[50:51], v[50:51], v[14:15], v[16:17] op_sel_hi:[1,0,1]"
    assert m.PK.search(hit) and m.LOW_READS_HIGH.search(hit)
    assert m.PK.search(miss) and not m.LOW_READS_HIGH.search(miss)
    assert m.LOW_READS_HIGH.search("v_pk_add_f32 v[22:23], v[22:23], v[24:25] op_sel:[0,1] op_sel_hi:[1,0]")

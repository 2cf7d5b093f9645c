#!/usr/bin/env python3
"""Disassemble every gfx950 code object of the built library and list the packed-FP32 instructions (v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32) whose op_sel makes the LOW result read an operand's HIGH register (the form that misbehaved beside another queue's
MFMA waves, LABNOTES.md).  usage: isa_scan.py [library.so]   exit code 1 when any is found."""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\bv_pk_(?:mul|add|fma)_f32\b")
LOW_READS_HIGH = re.compile(r"op_sel:\[[01,]*1[01,]*\]")


def scan(lib: Path):
    """-> list of (kernel, instruction) for every offending instruction"""
    found = []
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td) / lib.name
        shutil.copy(lib, tmp)                                  # llvm-objdump --offloading writes the bundles next to its input
        subprocess.run([OBJDUMP, "--offloading", str(tmp)], check=True, capture_output=True)
        cos = sorted(Path(td).glob(lib.name + ".*gfx950"))
        if not cos:
            raise RuntimeError("no gfx950 code objects in %s" % lib)
        for co in cos:
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    kernel = m.group(1)
                    continue
                if PK.search(line) and LOW_READS_HIGH.search(line):
                    found.append((kernel, re.sub(r"\s+", " ", line.split("//")[0]).strip()))
    return found


if __name__ == "__main__":
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "etude_amd" / "libetude_hip.so"
    bad = scan(lib)
    for k, ins in bad:
        print("%s: %s" % (k, ins))
    print("%d packed-FP32 instruction(s) whose low result reads a high register" % len(bad))
    sys.exit(1 if bad else 0)

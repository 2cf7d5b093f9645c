#!/usr/bin/env python3
"""Register / LDS / scratch footprint of every kernel in the built library (from the code objects' metadata notes).
usage: kernel_regs.py [pattern] [library.so]"""
import re, shutil, subprocess, sys, tempfile
from pathlib import Path
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

def kernels(lib: Path):
    out = []
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td) / lib.name
        shutil.copy(lib, tmp)
        subprocess.run([OBJDUMP, "--offloading", str(tmp)], check=True, capture_output=True)
        for co in sorted(Path(td).glob(lib.name + ".*gfx950")):
            txt = subprocess.run([READELF, "--notes", str(co)], check=True, capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]      # noqa: E731
                ag = blk.strip().split()[0]
                out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=ag, sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"), wg=g("max_flat_workgroup_size")))
    return out

if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = Path(sys.argv[2]) if len(sys.argv) > 2 else Path(__file__).resolve().parent.parent / "etude_amd" / "libetude_hip.so"
    for k in kernels(lib):
        nm = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        if pat in nm:
            print(f"{nm[:110]:110s} vgpr {k['vgpr']:>4} agpr {k['agpr']:>4} sgpr {k['sgpr']:>4} spill {k['spill']:>4} lds {k['lds']:>7} scratch {k['scratch']:>5} wg {k['wg']}")

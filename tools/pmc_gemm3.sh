#!/bin/bash
# PMC counters of k_gemm3 on the exact-parity mode's GEMM shapes (tools/bench_gemm3.py big): separate rocprofv3 --pmc passes with --kernel-trace only; per launch shape
# (launches are told apart by their grid): MFMA-pipe busy share, wave-cycle split (parked / issue-stalled / issuing), LDS bank conflicts, VALU instructions per wave,
# HBM bytes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md section HBM).
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/pmc_gemm3; rm -rf $O; mkdir -p $O
cd /tmp
B="python3 $R/tools/bench_gemm3.py big"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $B > /dev/null 2>$O/p1.err || { tail -5 $O/p1.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- $B > /dev/null 2>$O/p2.err || { tail -5 $O/p2.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $B > /dev/null 2>$O/p3.err || { tail -5 $O/p3.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > /dev/null 2>$O/p4.err || { tail -5 $O/p4.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
from collections import defaultdict
cnt = defaultdict(lambda: defaultdict(float)); nl = defaultdict(lambda: defaultdict(int)); dur = defaultdict(float); ndur = defaultdict(int)
def key(r):
    n = r["Kernel_Name"]
    if "k_gemm3" not in n: return None
    return "grid %s x %s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "?"))
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"gpurun_out/pmc_gemm3/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k: cnt[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k][r["Counter_Name"]] += 1
    if p == "p2":
        for f in glob.glob(f"gpurun_out/pmc_gemm3/{p}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = key(r)
                if k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; ndur[k] += 1
print("== k_gemm3 by launch grid (threads): averages per launch; counter mode slows the kernels: ratios matter")
print("launch                 launches  us(pmc run)  MFMA busy/SQ busy  parked  issue-stall  issuing  LDS conflict/active  VALU insts/wave  LDS insts/wave  fetch MB  write MB")
for k in sorted(cnt):
    c = cnt[k]; n = max(1, nl[k]["SQ_WAVE_CYCLES"]); wc = c["SQ_WAVE_CYCLES"] or 1.0
    us = dur[k] / max(1, ndur[k])
    print(f"{k:22s} {n:8d} {us:11.1f}  {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['SQ_BUSY_CYCLES'] or 1):17.3f}  {c['SQ_WAIT_ANY'] / wc:6.3f}  {c['SQ_WAIT_INST_ANY'] / wc:11.3f}  {c['SQ_ACTIVE_INST_ANY'] / wc:7.3f}"
          f"  {c['SQ_LDS_BANK_CONFLICT'] / (c['SQ_LDS_IDX_ACTIVE'] or 1):19.4f}  {c['SQ_INSTS_VALU'] / (c['SQ_WAVES'] or 1):15.0f}  {c['SQ_INSTS_LDS'] / (c['SQ_WAVES'] or 1):14.0f}"
          f"  {2.0 * c['FETCH_SIZE'] * 1024 / max(1, nl[k]['FETCH_SIZE']) / 1e6:8.1f}  {c['WRITE_SIZE'] * 1024 / max(1, nl[k]['WRITE_SIZE']) / 1e6:8.1f}")
PY

#!/bin/bash
# PMC counters of the batched-prefill kernels (k_pqkv, k_pattn, k_dmlp_fused) on ONE pass of the headline's shape (tools/bench_prefill.py: 500 prompts x 513 tokens),
# separate rocprofv3 --pmc passes with --kernel-trace only; per kernel: MFMA-pipe busy share, wave-cycle split (parked / issue-stalled / issuing), LDS bank conflicts,
# HBM bytes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md section HBM) and the effective clock (GRBM_GUI_ACTIVE / 8 / duration).
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/pmc_prefill; rm -rf $O; mkdir -p $O
cd /tmp
B="python3 $R/tools/bench_prefill.py --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $B > /dev/null 2>$O/p1.err || { tail -5 $O/p1.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- $B > /dev/null 2>$O/p2.err || { tail -5 $O/p2.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $B > /dev/null 2>$O/p3.err || { tail -5 $O/p3.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > /dev/null 2>$O/p4.err || { tail -5 $O/p4.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)", n); return m.group(1) if m else n[:30]
cnt = defaultdict(lambda: defaultdict(float)); nl = defaultdict(lambda: defaultdict(int)); dur = defaultdict(float); ndur = defaultdict(int)
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"gpurun_out/pmc_prefill/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); cnt[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k][r["Counter_Name"]] += 1
    if p == "p2":
        for f in glob.glob(f"gpurun_out/pmc_prefill/{p}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; ndur[k] += 1
print("== batched prefill, one 256 500-row pass x (warm-up passes + 1): per kernel, averages per launch (counter mode serialises and slows the kernels: ratios matter)")
print("kernel              launches  us(pmc run)  MFMA busy/wave-cyc  parked  issue-stall  issuing  LDS conflict/active  VALU insts/wave  fetch MB  write MB  clock GHz")
for k in ("k_pqkv", "k_pattn", "k_dmlp_fused", "k_linear", "k_ln_rows", "k_dembed"):
    c = cnt.get(k)
    if not c: continue
    n = max(1, nl[k]["SQ_WAVE_CYCLES"]); wc = c["SQ_WAVE_CYCLES"] or 1.0
    # SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD... report the plain ratios and the busy share of SQ_BUSY_CYCLES
    us = dur[k] / max(1, ndur[k])
    clk = c["GRBM_GUI_ACTIVE"] / max(1, nl[k]["GRBM_GUI_ACTIVE"]) / 8.0 / (us * 1e-6) / 1e9 if us > 0 else 0.0
    print(f"{k:18s} {n:8d} {us:11.1f}  {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['SQ_BUSY_CYCLES'] or 1):18.3f}  {c['SQ_WAIT_ANY'] / wc:6.3f}  {c['SQ_WAIT_INST_ANY'] / wc:11.3f}  {c['SQ_ACTIVE_INST_ANY'] / wc:7.3f}"
          f"  {c['SQ_LDS_BANK_CONFLICT'] / (c['SQ_LDS_IDX_ACTIVE'] or 1):19.4f}  {c['SQ_INSTS_VALU'] / (c['SQ_WAVES'] or 1):15.0f}  {2.0 * c['FETCH_SIZE'] * 1024 / max(1, nl[k]['FETCH_SIZE']) / 1e6:8.1f}  {c['WRITE_SIZE'] * 1024 / max(1, nl[k]['WRITE_SIZE']) / 1e6:8.1f}  {clk:9.2f}")
print("raw sums:")
for k in ("k_pqkv", "k_pattn", "k_dmlp_fused"):
    if k in cnt: print(k, {a: round(b) for a, b in cnt[k].items()})
PY
find $O -name "*.csv" -size +5M -delete

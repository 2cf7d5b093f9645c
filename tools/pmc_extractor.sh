#!/bin/bash
# PMC counters of the extractor kernels (separate passes, kernel-trace only), summarised per kernel.
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/pmc_ext; rm -rf $O; mkdir -p $O
cd /tmp
python3 $R/tools/bench_extractor.py 4 3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/bench_extractor.py 1 1 > /dev/null 2>$O/p1.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/bench_extractor.py 1 1 > /dev/null 2>$O/p2.err
# HBM bytes of the whole extractor (4 windows in one batch): FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md section HBM) and WRITE_SIZE, separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- python3 $R/tools/bench_extractor.py 4 0 4 > /dev/null 2>$O/p3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- python3 $R/tools/bench_extractor.py 4 0 4 > /dev/null 2>$O/p4.err
cd $R
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)", n); return m.group(1) if m else n[:30]
tot = defaultdict(lambda: [0.0, 0.0, 0])
for p, col, mul in (("p3", 0, 2.0), ("p4", 1, 1.0)):
    for f in glob.glob(f"gpurun_out/pmc_ext/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"): continue
            k = short(r["Kernel_Name"]); tot[k][col] += mul * float(r["Counter_Value"]) * 1024
            if col == 0: tot[k][2] += 1
# bench_extractor with reps = 0 runs the model twice (warm-up + the profiler pass): 8 windows in all
nwin = 8.0
print("== HBM traffic of the extractor, GB per 512-frame window (FETCH_SIZE x2 + WRITE_SIZE over 8 windows in batches of 4): kernel, launches, fetch, write, total")
gt = 0.0
for k, (f, w, n) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    if not k.startswith("k_"): continue
    print(f"{k:18s} {n:5d} {f/nwin/1e9:8.3f} {w/nwin/1e9:8.3f} {(f+w)/nwin/1e9:8.3f}")
    gt += (f + w) / nwin / 1e9
print(f"total {gt:.3f} GB per window (round 1: 9.2)")
PY
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)(?:<([^>]*)>)?", n); return (m.group(1) + (f"<{m.group(2)}>" if m.group(2) else "")) if m else n[:30]
for p in ("p1", "p2"):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in glob.glob(f"gpurun_out/pmc_ext/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for k in acc for c in acc[k]})
    print("==", p, names)
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("SQ_LDS_IDX_ACTIVE", 0)))[:8]:
        print(k.ljust(16), " ".join(f"{acc[k].get(c,0):.3e}" for c in names))
PY
tail -3 $O/p1.err
find $O -name "*.csv" -size +5M -delete
